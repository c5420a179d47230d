// Host-only C++ side of binocular3dgs_amd._C (g++ against the torch headers; no device code in this directory).
// Everything here calls libb3gs_raster.so through the C ABI of include/b3gs_raster.h -- the same boundary the reference's
// own binding would be given -- and keeps torch types OUT of that boundary.
#pragma once
#include <torch/extension.h>
#include <torch/csrc/Exceptions.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>

#include <string>

#include "../../../include/b3gs_raster.h"

namespace b3 {

// Raises binocular3dgs_amd._lib.B3gsError (a RuntimeError) -- from a pybind call or from inside an autograd node running on
// the engine's thread (python_error travels through the engine with its Python type intact).
[[noreturn]] void raise(const std::string& msg);
void check(int rc, const char* what);

inline b3gs_stream_t cur_stream(const at::Device& d) {
  return (b3gs_stream_t)c10::hip::getCurrentHIPStream(d.index()).stream();
}

// Switches the current device only when it is not `d` already (a launch through the C ABI uses the current device).
struct DeviceGuard {
  c10::DeviceIndex prev = -1;
  explicit DeviceGuard(const at::Device& d) {
    c10::DeviceIndex cur = c10::hip::current_device();
    if (d.has_index() && d.index() != cur) {
      prev = cur;
      c10::hip::set_device(d.index());
    }
  }
  ~DeviceGuard() {
    if (prev >= 0) c10::hip::set_device(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// A float32, contiguous tensor on the HIP device, or B3gsError (there is no CPU path).
inline at::Tensor dev_f32(const at::Tensor& t, const char* name, const char* what) {
  if (!t.is_cuda()) raise(std::string(name) + " is on " + t.device().str() + ": " + what);
  at::Tensor r = t.scalar_type() == at::kFloat ? t : t.to(at::kFloat);
  return r.is_contiguous() ? r : r.contiguous();
}
inline const float* fptr(const at::Tensor& t) { return (t.defined() && t.numel()) ? t.data_ptr<float>() : nullptr; }
inline float* fptr_mut(at::Tensor& t) { return (t.defined() && t.numel()) ? t.data_ptr<float>() : nullptr; }

void bind_loss(pybind11::module_& m);
void bind_optim(pybind11::module_& m);
void bind_raster(pybind11::module_& m);

}  // namespace b3
