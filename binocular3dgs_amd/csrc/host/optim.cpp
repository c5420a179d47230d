// The per-iteration model / optimiser statements of the training loop, each ONE launch through the C ABI:
//   gaussians.optimizer.step()              train.py:196-198   -> b3gs_adam_step_at   (optim.Adam keeps torch's state layout)
//   gaussians.opacity_decay(factor)         scene/gaussian_model.py:307-309 -> b3gs_opacity_decay
//   gaussians.add_densification_stats(...)  scene/gaussian_model.py:409-411 -> b3gs_add_densification_stats
#include "common.h"

namespace py = pybind11;
using at::Tensor;

namespace b3 {

static const char* NO_CPU_OPT = "binocular3dgs_amd.optim.Adam: parameters must live on the HIP device (no CPU path)";

// One Adam step (step number `step`, 1-based) for parameter tensors that share betas / eps; learning rate per tensor.
// Up to 8 tensors per launch.  The kernels write through raw pointers: the version counters are bumped here.
static void adam_step_at(const std::vector<Tensor>& params, const std::vector<Tensor>& grads, const std::vector<Tensor>& exp_avg,
                         const std::vector<Tensor>& exp_avg_sq, const std::vector<double>& lrs, int64_t step, double beta1,
                         double beta2, double eps) {
  size_t n = params.size();
  if (grads.size() != n || exp_avg.size() != n || exp_avg_sq.size() != n || lrs.size() != n)
    throw py::value_error("adam_step_at: params / grads / exp_avg / exp_avg_sq / lrs must have the same length");
  if (n == 0) return;
  at::Device dev = params[0].device();
  std::vector<Tensor> g32(n);
  for (size_t i = 0; i < n; ++i) {
    const Tensor &p = params[i], &m = exp_avg[i], &v = exp_avg_sq[i];
    if (!p.is_cuda()) raise(NO_CPU_OPT);
    if (p.device() != dev) throw py::value_error("adam_step_at: one call, one device");
    if (p.scalar_type() != at::kFloat || !p.is_contiguous())
      raise("binocular3dgs_amd.optim.Adam: parameters must be contiguous float32 tensors");
    if (!(m.is_contiguous() && v.is_contiguous() && m.scalar_type() == at::kFloat && v.scalar_type() == at::kFloat &&
          m.device() == dev && v.device() == dev && m.numel() == p.numel() && v.numel() == p.numel()))
      raise("binocular3dgs_amd.optim.Adam: exp_avg / exp_avg_sq must be contiguous float32 tensors of the parameter's size "
            "on its device");
    const Tensor& g = grads[i];
    if (g.is_sparse()) throw std::runtime_error("Adam does not support sparse gradients");
    g32[i] = (g.scalar_type() == at::kFloat && g.is_contiguous()) ? g : g.to(at::kFloat).contiguous();
    if (g32[i].numel() != p.numel() || g32[i].device() != dev)
      throw py::value_error("adam_step_at: a gradient does not match its parameter");
  }
  DeviceGuard guard(dev);
  b3gs_stream_t s = cur_stream(dev);
  for (size_t c0 = 0; c0 < n; c0 += 8) {
    B3gsAdamSegment segs[8];
    int k = 0;
    for (size_t i = c0; i < n && k < 8; ++i, ++k) {
      Tensor p = params[i], m = exp_avg[i], v = exp_avg_sq[i];
      B3gsAdamSegment& sg = segs[k];
      int64_t cnt = p.numel();
      sg.param = cnt ? p.data_ptr<float>() : nullptr;
      sg.grad = cnt ? g32[i].data_ptr<float>() : nullptr;
      sg.exp_avg = cnt ? m.data_ptr<float>() : nullptr;
      sg.exp_avg_sq = cnt ? v.data_ptr<float>() : nullptr;
      sg.count = cnt;
      sg.lr = (float)lrs[i];
      sg.row_len = 0;
      sg.first_row = 0;
      sg.lr_dev = nullptr;
    }
    check(b3gs_adam_step_at(k, segs, (int32_t)step, (float)beta1, (float)beta2, (float)eps, s), "b3gs_adam_step_at");
  }
  // (saved-tensor checks; the depth-order hint of rasterizer._RasterizeRaw keys on the position tensor's version)
  for (const Tensor& p : params) p.unsafeGetTensorImpl()->bump_version();
}

static void opacity_decay(const Tensor& opacity, double factor) {
  if (!opacity.is_cuda()) raise("opacity_decay: _opacity is on the CPU (GaussianModel keeps the PyTorch statement for that)");
  if (!(opacity.is_contiguous() && opacity.scalar_type() == at::kFloat))
    raise("opacity_decay: _opacity must be a contiguous float32 tensor");
  {
    DeviceGuard guard(opacity.device());
    check(b3gs_opacity_decay(opacity.data_ptr<float>(), opacity.numel(), (float)factor, cur_stream(opacity.device())),
          "b3gs_opacity_decay");
  }
  opacity.unsafeGetTensorImpl()->bump_version();
}

// rows selected by the boolean `update_filter`: accum += ||grad[row, :2]||, denom += 1.  (the mask must have one
// element per row and live on the gradient's device -- a short or host-resident mask would be read out of bounds.)
static void add_densification_stats(const Tensor& grad, const Tensor& update_filter, const Tensor& accum, const Tensor& denom) {
  int64_t P = grad.size(0);
  if (update_filter.scalar_type() != at::kBool || update_filter.dim() != 1 || update_filter.size(0) != P)
    throw py::index_error("add_densification_stats: the boolean mask has shape " + std::to_string(update_filter.numel()) +
                          " for " + std::to_string(P) + " Gaussians");
  if (update_filter.device() != grad.device() || accum.device() != grad.device() || denom.device() != grad.device())
    raise("add_densification_stats: mask, statistics and gradient must live on the same HIP device");
  if (!(grad.is_cuda() && grad.scalar_type() == at::kFloat && grad.dim() == 2 && grad.stride(1) == 1 && accum.is_contiguous() &&
        denom.is_contiguous() && accum.scalar_type() == at::kFloat && denom.scalar_type() == at::kFloat &&
        accum.numel() == P && denom.numel() == P))
    raise("add_densification_stats: expects a float32 [P, >=2] gradient with unit inner stride and contiguous float32 [P, 1] "
          "statistics");
  Tensor f = update_filter.is_contiguous() ? update_filter : update_filter.contiguous();
  DeviceGuard guard(grad.device());
  check(b3gs_add_densification_stats(P, grad.data_ptr<float>(), grad.stride(0), (const uint8_t*)f.data_ptr<bool>(),
                                     accum.data_ptr<float>(), denom.data_ptr<float>(), cur_stream(grad.device())),
        "b3gs_add_densification_stats");
  accum.unsafeGetTensorImpl()->bump_version();
  denom.unsafeGetTensorImpl()->bump_version();
}

void bind_optim(py::module_& m) {
  m.def("adam_step_at", &adam_step_at, py::arg("params"), py::arg("grads"), py::arg("exp_avg"), py::arg("exp_avg_sq"),
        py::arg("lrs"), py::arg("step"), py::arg("beta1"), py::arg("beta2"), py::arg("eps"));
  m.def("opacity_decay", &opacity_decay, py::arg("opacity"), py::arg("factor"));
  m.def("add_densification_stats", &add_densification_stats, py::arg("grad"), py::arg("update_filter"),
        py::arg("xyz_gradient_accum"), py::arg("denom"));
}

}  // namespace b3
