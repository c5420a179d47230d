// binocular3dgs_amd._C -- the compiled module the reference's `diff_gaussian_rasterization._C` is (README.md:37,
// gaussian_renderer/__init__.py:14): `rasterize_gaussians`, `rasterize_gaussians_backward`, `mark_visible` with the upstream
// positional signatures, plus the autograd nodes and launch assembly of this build's python surface (rasterizer.py,
// loss_utils.py, graphics_utils.py, optim.py, gaussian_model.py), which keeps POLICY only.
#include "common.h"

namespace py = pybind11;

namespace b3 {

static PyObject* g_error_class = nullptr;   // binocular3dgs_amd._lib.B3gsError

[[noreturn]] void raise(const std::string& msg) {
  py::gil_scoped_acquire gil;
  PyErr_SetString(g_error_class ? g_error_class : PyExc_RuntimeError, msg.c_str());
  python_error err;
  err.persist();
  throw err;
}

void check(int rc, const char* what) {
  if (rc == B3GS_OK) return;
  const char* name = rc == B3GS_ERR_ARG ? "B3GS_ERR_ARG" : rc == B3GS_ERR_ALLOC ? "B3GS_ERR_ALLOC" : rc == B3GS_ERR_HIP ? "B3GS_ERR_HIP"
                     : rc == B3GS_ERR_CAPACITY ? "B3GS_ERR_CAPACITY" : rc == B3GS_ERR_NO_DEVICE ? "B3GS_ERR_NO_DEVICE" : "error";
  const char* last = b3gs_last_error();
  raise(std::string(what) + " failed: " + name + ": " + (last ? last : ""));
}

}  // namespace b3

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "binocular3dgs_amd._C: host-only launch layer over libb3gs_raster.so (include/b3gs_raster.h)";
  {
    py::object cls = py::module_::import("binocular3dgs_amd._lib").attr("B3gsError");
    b3::g_error_class = cls.inc_ref().ptr();
    m.attr("B3gsError") = cls;
  }
  py::register_exception_translator([](std::exception_ptr p) {
    try {
      if (p) std::rethrow_exception(p);
    } catch (python_error& e) {
      e.restore();
    }
  });
  m.def("abi_version", []() { return b3gs_abi_version(); });
  m.attr("ABI_VERSION") = B3GS_ABI_VERSION;
  b3::bind_loss(m);
  b3::bind_optim(m);
  b3::bind_raster(m);
}
