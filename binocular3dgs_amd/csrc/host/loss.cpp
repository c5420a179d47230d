// The reference's loss functions as C++ autograd nodes (one launch forward, one backward, each through the C ABI):
//   l1_loss(network_output, gt, mask=None)                  utils/loss_utils.py:18-21
//   ssim(img1, img2, window_size=11, size_average=True)     utils/loss_utils.py:36-66
//   SmoothLoss.forward(disparity, image)                    utils/loss_utils.py:68-91
//   inverse_warp_images(image, disparity, rows, cols)       utils/graphics_utils.py:80-125
// Argument checks and broadcasting rules live here too; binocular3dgs_amd/loss_utils.py / graphics_utils.py keep the
// reference's names and docstrings in front of them.  float64 inputs are computed and returned in float32 (the kernels'
// arithmetic type; the reference would keep float64).
#include "common.h"

#include <map>
#include <mutex>

namespace py = pybind11;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;
using at::Tensor;

namespace b3 {

static const char* NO_CPU = "the loss functions run on the HIP device only (binocular3dgs_amd.loss holds the PyTorch statement)";

// ---- the scalar reductions' workspace: one zeroed, self-cleaning buffer per (device, stream) ---------------------------
// (guarded by a lock; an entry whose launch failed is dropped, so the next call starts from fresh zeros.)
static std::mutex g_ws_lock;
static std::map<std::pair<int, void*>, Tensor> g_ws;

static float* workspace(const at::Device& dev, b3gs_stream_t stream, int64_t planes, int H, int W) {
  size_t need = b3gs_lossfn_workspace_floats(planes, H, W);
  std::lock_guard<std::mutex> lock(g_ws_lock);
  auto key = std::make_pair((int)dev.index(), (void*)stream);
  auto it = g_ws.find(key);
  if (it == g_ws.end() || (size_t)it->second.numel() < need) {
    Tensor w = at::zeros({(int64_t)need}, at::TensorOptions().dtype(at::kFloat).device(dev));
    g_ws[key] = w;
    return w.data_ptr<float>();
  }
  return it->second.data_ptr<float>();
}
static void drop_workspace(const at::Device& dev, b3gs_stream_t stream) {
  std::lock_guard<std::mutex> lock(g_ws_lock);
  g_ws.erase(std::make_pair((int)dev.index(), (void*)stream));
}
static void check_ws(int rc, const char* what, const at::Device& dev, b3gs_stream_t stream) {
  if (rc != B3GS_OK) drop_workspace(dev, stream);     // its arrival counters may be left non-zero
  check(rc, what);
}

static Tensor grad_scalar(const Tensor& g) {
  Tensor r = g.scalar_type() == at::kFloat ? g : g.to(at::kFloat);
  return r.is_contiguous() ? r : r.contiguous();
}

// ---- l1 ------------------------------------------------------------------------------------------------------------------
struct L1Fn : public torch::autograd::Function<L1Fn> {
  // mean |x*m - y*m| over [batch, channels, hw]; mask undefined or [batch, hw]
  // (an absent mask travels as nullopt: the engine records device / layout of every TENSOR argument and refuses an
  // undefined one)
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& y, const c10::optional<Tensor>& mask_opt,
                        int64_t batch, int64_t channels, int64_t hw, bool nx, bool ny, bool nm) {
    Tensor mask = mask_opt.has_value() ? *mask_opt : Tensor();
    at::Device dev = x.device();
    Tensor out = at::empty({}, x.options());
    b3gs_stream_t s = cur_stream(dev);
    {
      DeviceGuard g(dev);
      check_ws(b3gs_l1_loss_forward(fptr(x), fptr(y), fptr(mask), batch, (int32_t)channels, hw, out.data_ptr<float>(),
                                    workspace(dev, s, 1, 32, 32), s), "b3gs_l1_loss_forward", dev, s);
    }
    ctx->save_for_backward({x, y, mask});
    ctx->saved_data["dims"] = std::vector<int64_t>{batch, channels, hw, nx, ny, nm};
    return out;
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &y = saved[1], &mask = saved[2];
    auto d = ctx->saved_data["dims"].toIntVector();
    Tensor gx = d[3] ? at::empty_like(x) : Tensor();
    Tensor gy = d[4] ? at::empty_like(y) : Tensor();
    Tensor gm = (d[5] && mask.defined()) ? at::empty_like(mask) : Tensor();
    Tensor g = grad_scalar(grads[0]);
    {
      DeviceGuard guard(x.device());
      check(b3gs_l1_loss_backward(fptr(x), fptr(y), fptr(mask), d[0], (int32_t)d[1], d[2], g.data_ptr<float>(), fptr_mut(gx),
                                  fptr_mut(gy), fptr_mut(gm), cur_stream(x.device())), "b3gs_l1_loss_backward");
    }
    return {gx, gy, gm, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

static bool wants_grad(const Tensor& t) { return t.defined() && t.requires_grad() && at::GradMode::is_enabled(); }

static Tensor l1_loss(const Tensor& network_output, const Tensor& gt, const c10::optional<Tensor>& mask_opt) {
  Tensor x = network_output, y = gt;
  Tensor mask = mask_opt.has_value() ? *mask_opt : Tensor();
  if (x.sizes() != y.sizes()) {
    auto b = at::broadcast_tensors({x, y});
    x = b[0], y = b[1];
  }
  bool per_channel = false;
  if (mask.defined() && mask.sizes() != x.sizes()) {
    // the reference's use (train.py:135): [1,3,H,W] against a [1,1,H,W] mask -- broadcast over the channel axis in-kernel
    int64_t n = x.dim();
    per_channel = n >= 3 && mask.dim() == n && mask.size(n - 3) == 1 && mask.size(n - 2) == x.size(n - 2) &&
                  mask.size(n - 1) == x.size(n - 1);
    for (int64_t i = 0; per_channel && i < n - 3; ++i) per_channel = mask.size(i) == x.size(i);
    if (!per_channel) {
      auto shape = at::infer_size(x.sizes(), mask.sizes());
      x = x.expand(shape), y = y.expand(shape), mask = mask.expand(shape);
    }
  }
  bool nx = wants_grad(x), ny = wants_grad(y), nm = wants_grad(mask);
  x = dev_f32(x, "network_output", NO_CPU), y = dev_f32(y, "gt", NO_CPU);
  int64_t n = x.numel();
  if (n == 0) return at::abs(x - y).mean();     // (nan, like the reference)
  if (!mask.defined()) return L1Fn::apply(x, y, c10::optional<Tensor>(), (int64_t)1, (int64_t)1, n, nx, ny, false);
  mask = dev_f32(mask, "mask", NO_CPU);
  if (mask.sizes() == x.sizes()) return L1Fn::apply(x, y, c10::optional<Tensor>(mask), (int64_t)1, (int64_t)1, n, nx, ny, nm);
  int64_t hw = x.size(-1) * x.size(-2), c = x.size(-3);
  return L1Fn::apply(x, y, c10::optional<Tensor>(mask), n / (c * hw), c, hw, nx, ny, nm);
}

// ---- ssim ----------------------------------------------------------------------------------------------------------------
struct SsimFn : public torch::autograd::Function<SsimFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& img1, const Tensor& img2, int64_t batch, int64_t channels,
                        bool size_average, bool need1, bool need2) {
    at::Device dev = img1.device();
    int64_t H = img1.size(-2), W = img1.size(-1), planes = batch * channels;
    Tensor maps = (need1 || need2) ? at::empty({need2 ? 5 : 3, planes, H, W}, img1.options()) : Tensor();
    Tensor out = size_average ? at::empty({}, img1.options()) : at::empty({batch}, img1.options());
    b3gs_stream_t s = cur_stream(dev);
    {
      DeviceGuard g(dev);
      check_ws(b3gs_ssim_forward(fptr(img1), fptr(img2), (int32_t)batch, (int32_t)channels, (int32_t)H, (int32_t)W,
                                 size_average ? 1 : 0, fptr_mut(maps), need2 ? 1 : 0, out.data_ptr<float>(),
                                 workspace(dev, s, planes, (int)H, (int)W), s), "b3gs_ssim_forward", dev, s);
    }
    ctx->save_for_backward({img1, img2, maps});
    ctx->saved_data["dims"] = std::vector<int64_t>{batch, channels, H, W, size_average, need1, need2};
    return out;
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &img1 = saved[0], &img2 = saved[1], &maps = saved[2];
    auto d = ctx->saved_data["dims"].toIntVector();
    Tensor g1 = d[5] ? at::empty_like(img1) : Tensor();
    Tensor g2 = d[6] ? at::empty_like(img2) : Tensor();
    Tensor g = grad_scalar(grads[0]);
    if (maps.defined()) {
      DeviceGuard guard(img1.device());
      check(b3gs_ssim_backward(fptr(img1), fptr(img2), fptr(maps), (int32_t)d[0], (int32_t)d[1], (int32_t)d[2], (int32_t)d[3],
                               (int32_t)d[4], g.data_ptr<float>(), fptr_mut(g1), fptr_mut(g2), cur_stream(img1.device())),
            "b3gs_ssim_backward");
    }
    return {g1, g2, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

static Tensor ssim(const Tensor& a, const Tensor& b, int64_t window_size, bool size_average) {
  if (window_size != 11) raise("ssim: the HIP kernels are built for the reference's window_size=11");
  Tensor img1 = a, img2 = b;
  if (img1.sizes() != img2.sizes()) {
    auto bc = at::broadcast_tensors({img1, img2});
    img1 = bc[0], img2 = bc[1];
  }
  if (img1.dim() != 3 && img1.dim() != 4) throw py::value_error("ssim expects [C,H,W] or [B,C,H,W] images");
  if (!size_average && img1.dim() != 4)
    throw py::index_error("Dimension out of range (size_average=False needs a [B,C,H,W] input, as in the reference)");
  bool n1 = wants_grad(img1), n2 = wants_grad(img2);
  img1 = dev_f32(img1, "img1", NO_CPU), img2 = dev_f32(img2, "img2", NO_CPU);
  int64_t batch = img1.dim() == 4 ? img1.size(0) : 1;
  return SsimFn::apply(img1, img2, batch, img1.size(-3), size_average, n1, n2);
}

// ---- edge-aware smoothness ---------------------------------------------------------------------------------------------
struct SmoothFn : public torch::autograd::Function<SmoothFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& disparity, const Tensor& image, bool nd, bool ni) {
    at::Device dev = disparity.device();
    Tensor out = at::empty({}, disparity.options());
    b3gs_stream_t s = cur_stream(dev);
    {
      DeviceGuard g(dev);
      check_ws(b3gs_smooth_loss_forward(fptr(disparity), fptr(image), (int32_t)image.size(0), (int32_t)image.size(1),
                                        (int32_t)image.size(2), (int32_t)image.size(3), out.data_ptr<float>(),
                                        workspace(dev, s, 1, 32, 32), s), "b3gs_smooth_loss_forward", dev, s);
    }
    ctx->save_for_backward({disparity, image});
    ctx->saved_data["need"] = std::vector<int64_t>{nd, ni};
    return out;
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &disparity = saved[0], &image = saved[1];
    auto need = ctx->saved_data["need"].toIntVector();
    Tensor gd = need[0] ? at::empty_like(disparity) : Tensor();
    Tensor gi = need[1] ? at::empty_like(image) : Tensor();
    Tensor g = grad_scalar(grads[0]);
    {
      DeviceGuard guard(image.device());
      check(b3gs_smooth_loss_backward(fptr(disparity), fptr(image), (int32_t)image.size(0), (int32_t)image.size(1),
                                      (int32_t)image.size(2), (int32_t)image.size(3), g.data_ptr<float>(), fptr_mut(gd),
                                      fptr_mut(gi), cur_stream(image.device())), "b3gs_smooth_loss_backward");
    }
    return {gd, gi, Tensor(), Tensor()};
  }
};

static std::string shape_str(const Tensor& t) {
  std::string s = "(";
  for (int64_t i = 0; i < t.dim(); ++i) s += (i ? ", " : "") + std::to_string(t.size(i));
  return s + (t.dim() == 1 ? ",)" : ")");
}

static Tensor smooth_loss(const Tensor& disparity, const Tensor& image) {
  if (disparity.dim() != 4 || image.dim() != 4 || disparity.size(1) != 1 || disparity.size(0) != image.size(0) ||
      disparity.size(2) != image.size(2) || disparity.size(3) != image.size(3))
    throw py::value_error("SmoothLoss expects disparity [B,1,H,W] and image [B,C,H,W], got " + shape_str(disparity) + " and " +
                          shape_str(image));
  if (image.size(-1) < 3 || image.size(-2) < 3)
    throw std::runtime_error("SmoothLoss: the 3x3 stencil needs at least 3x3 pixels (the reference's convolution raises too)");
  bool nd = wants_grad(disparity), ni = wants_grad(image);
  return SmoothFn::apply(dev_f32(disparity, "disparity", NO_CPU), dev_f32(image, "image", NO_CPU), nd, ni);
}

// ---- inverse warp ------------------------------------------------------------------------------------------------------
struct WarpFn : public torch::autograd::Function<WarpFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& image, const Tensor& disparity, bool ni, bool nd) {
    at::Device dev = image.device();
    Tensor out = at::empty_like(image);
    // the gradient of the image is a scatter (atomics): its buffer is zeroed by THIS launch on its way through the pixels,
    // so the backward needs no fill in front of it
    Tensor gbuf = ni ? at::empty_like(image) : Tensor();
    {
      DeviceGuard g(dev);
      check(b3gs_inverse_warp_forward(fptr(image), fptr(disparity), (int32_t)image.size(0), (int32_t)image.size(1),
                                      (int32_t)image.size(2), (int32_t)image.size(3), out.data_ptr<float>(), fptr_mut(gbuf),
                                      cur_stream(dev)), "b3gs_inverse_warp_forward");
    }
    ctx->save_for_backward({image, disparity});
    ctx->saved_data["need"] = std::vector<int64_t>{ni, nd};
    if (gbuf.defined()) ctx->saved_data["gbuf"] = gbuf;
    return out;
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &image = saved[0], &disparity = saved[1];
    auto need = ctx->saved_data["need"].toIntVector();
    Tensor g = dev_f32(grads[0], "grad", NO_CPU);
    Tensor gi;
    if (need[0]) {
      auto it = ctx->saved_data.find("gbuf");
      if (it != ctx->saved_data.end()) {
        gi = it->second.toTensor();             // (handed to autograd: nothing here keeps a reference to it)
        ctx->saved_data.erase(it);
      } else {
        gi = at::zeros_like(image);             // a second backward through a retained graph
      }
    }
    Tensor gd = need[1] ? at::empty_like(disparity) : Tensor();
    {
      DeviceGuard guard(image.device());
      check(b3gs_inverse_warp_backward(fptr(image), fptr(disparity), fptr(g), (int32_t)image.size(0), (int32_t)image.size(1),
                                       (int32_t)image.size(2), (int32_t)image.size(3), fptr_mut(gi), fptr_mut(gd),
                                       cur_stream(image.device())), "b3gs_inverse_warp_backward");
    }
    return {gi, gd, Tensor(), Tensor()};
  }
};

static Tensor inverse_warp_images(const Tensor& image, const Tensor& disparity, const py::object&, const py::object&) {
  // row_indices / column_indices (the reference's meshgrid of pixel coordinates, train.py:56-57) are accepted for signature
  // compatibility; the kernel knows where its pixels are
  if (image.dim() != 4 || disparity.dim() != 4 || disparity.size(1) != 1 || disparity.size(0) != image.size(0) ||
      disparity.size(2) != image.size(2) || disparity.size(3) != image.size(3))
    throw py::value_error("inverse_warp_images expects image [B,C,H,W] and disparity [B,1,H,W], got " + shape_str(image) +
                          " and " + shape_str(disparity));
  bool ni = wants_grad(image), nd = wants_grad(disparity);
  return WarpFn::apply(dev_f32(image, "image", NO_CPU), dev_f32(disparity, "disparity", NO_CPU), ni, nd);
}

void bind_loss(py::module_& m) {
  m.def("l1_loss", &l1_loss, py::arg("network_output"), py::arg("gt"), py::arg("mask") = py::none());
  m.def("ssim", &ssim, py::arg("img1"), py::arg("img2"), py::arg("window_size") = 11, py::arg("size_average") = true);
  m.def("smooth_loss", &smooth_loss, py::arg("disparity"), py::arg("image"));
  m.def("inverse_warp_images", &inverse_warp_images, py::arg("image"), py::arg("disparity"),
        py::arg("row_indices") = py::none(), py::arg("column_indices") = py::none());
  m.def("_lossfn_workspaces", []() {
    std::lock_guard<std::mutex> lock(g_ws_lock);
    return (int64_t)g_ws.size();
  });
}

}  // namespace b3
