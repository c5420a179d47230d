// The rasterizer behind the reference's binding surface (gaussian_renderer/__init__.py:14,36-49,85-93):
//   _C.rasterize_gaussians(...) / _C.rasterize_gaussians_backward(...) / _C.mark_visible(...)   upstream positional order
//   _RasterizeGaussians as a C++ autograd node (rasterize_gaussians_autograd)
// and the launch assembly of this build's raw-parameter node (rasterizer._RasterizeRaw: policy in python, structs here):
//   raw_prepare / raw_forward_launch / raw_backward_launch  ->  b3gs_forward_raw_batch, b3gs_blend_backward_batch,
//   b3gs_backward_raw_accumulate.
// No CPU path: host tensors raise B3gsError.
#include "common.h"

#include <limits>

namespace py = pybind11;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;
using at::Tensor;

namespace b3 {

static const char* NO_CPU_R = "the rasterizer runs on an MI355X (HIP) device only; there is no CPU path";

static at::TensorOptions u8(const at::Device& d) { return at::TensorOptions().dtype(at::kByte).device(d); }
static at::TensorOptions f32(const at::Device& d) { return at::TensorOptions().dtype(at::kFloat).device(d); }
static at::TensorOptions i32(const at::Device& d) { return at::TensorOptions().dtype(at::kInt).device(d); }

static bool present(const Tensor& t) { return t.defined() && t.numel() != 0; }

// ---- the scene struct of one view of explicit (activated) tensors, validated like the upstream binding ------------------
struct Scene {
  B3gsScene sc{};
  std::vector<Tensor> keep;
  at::Device dev{at::kCPU};
  int64_t P = 0, M = 0;
};

static Tensor opt_rows(Scene& s, const Tensor& t, const char* name, int64_t cols) {
  if (!present(t)) return Tensor();
  Tensor r = dev_f32(t, name, NO_CPU_R);
  if (r.dim() != 2 || r.size(0) != s.P || r.size(1) != cols)
    throw py::value_error(std::string(name) + " must have dimensions (num_points, " + std::to_string(cols) + ")");
  s.keep.push_back(r);
  return r;
}

static Scene make_scene(const Tensor& background, const Tensor& means3D_in, const Tensor& colors, const Tensor& opacity,
                        const Tensor& scales, const Tensor& rotations, double scale_modifier, const Tensor& cov3D_precomp,
                        const Tensor& viewmatrix, const Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t H,
                        int64_t W, const Tensor& sh, int64_t degree, const Tensor& campos, bool prefiltered, bool debug) {
  Scene s;
  Tensor means3D = dev_f32(means3D_in, "means3D", NO_CPU_R);
  if (means3D.dim() != 2 || means3D.size(1) != 3) throw py::value_error("means3D must have dimensions (num_points, 3)");
  s.P = means3D.size(0);
  s.dev = means3D.device();
  s.keep.push_back(means3D);
  Tensor sh_t;
  if (present(sh)) {
    sh_t = dev_f32(sh, "sh", NO_CPU_R);
    if (sh_t.dim() != 3 || sh_t.size(0) != s.P || sh_t.size(2) != 3)
      throw py::value_error("sh must have dimensions (num_points, K, 3)");
    s.M = sh_t.size(1);
    s.keep.push_back(sh_t);
  }
  Tensor colors_t = opt_rows(s, colors, "colors_precomp", 3);
  Tensor opacity_t = dev_f32(opacity, "opacities", NO_CPU_R).reshape({-1});
  if (opacity_t.numel() != s.P) throw py::value_error("opacities must have num_points elements");
  Tensor scales_t = opt_rows(s, scales, "scales", 3);
  Tensor rot_t = opt_rows(s, rotations, "rotations", 4);
  Tensor cov_t = opt_rows(s, cov3D_precomp, "cov3D_precomp", 6);
  Tensor bg = dev_f32(background, "bg", NO_CPU_R).reshape({-1});
  Tensor vm = dev_f32(viewmatrix, "viewmatrix", NO_CPU_R).reshape({-1});
  Tensor pm = dev_f32(projmatrix, "projmatrix", NO_CPU_R).reshape({-1});
  Tensor cp = dev_f32(campos, "campos", NO_CPU_R).reshape({-1});
  if (bg.numel() != 3 || vm.numel() != 16 || pm.numel() != 16 || cp.numel() != 3)
    throw py::value_error("bg/campos must have 3 and viewmatrix/projmatrix 16 elements");
  for (const Tensor& t : {opacity_t, bg, vm, pm, cp}) s.keep.push_back(t);
  // one launch reads every tensor through raw pointers on the device of means3D: a tensor of another GPU would be a wild
  // pointer there (the upstream binding leaves that to the CUDA runtime; here it is an argument error)
  for (const Tensor& t : s.keep)
    if (t.device() != s.dev)
      throw py::value_error("all rasterizer inputs must live on the device of means3D (" + s.dev.str() + "), got a tensor on " +
                            t.device().str());
  B3gsScene& c = s.sc;
  c.P = (int32_t)s.P, c.D = (int32_t)degree, c.M = (int32_t)s.M, c.W = (int32_t)W, c.H = (int32_t)H;
  c.tan_fovx = (float)tan_fovx, c.tan_fovy = (float)tan_fovy, c.scale_modifier = (float)scale_modifier;
  c.prefiltered = prefiltered ? 1 : 0, c.debug = debug ? 1 : 0;
  c.background = fptr(bg), c.means3D = fptr(means3D), c.shs = fptr(sh_t), c.colors_precomp = fptr(colors_t);
  c.opacities = fptr(opacity_t), c.scales = fptr(scales_t), c.rotations = fptr(rot_t), c.cov3D_precomp = fptr(cov_t);
  c.viewmatrix = fptr(vm), c.projmatrix = fptr(pm), c.campos = fptr(cp);
  return s;
}

// ---- forward: exact (one read-back of N, buffers sized by the library through callbacks) or into a given capacity ------
struct AllocSlot {
  Tensor t;
  at::Device dev{at::kCPU};
};
static char* alloc_cb(void* user, size_t bytes) {
  auto* slot = static_cast<AllocSlot*>(user);
  slot->t = at::empty({(int64_t)std::max<size_t>(bytes, 1)}, u8(slot->dev));
  return (char*)slot->t.data_ptr();
}

struct FwdOut {
  int64_t n = 0;          // exact: N; capacity mode: the capacity
  Tensor color, depth, alpha, radii, geom, binning, img, n_dev;
};

static FwdOut forward_impl(const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity,
                           const Tensor& scales, const Tensor& rotations, double scale_modifier, const Tensor& cov3D_precomp,
                           const Tensor& viewmatrix, const Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t H,
                           int64_t W, const Tensor& sh, int64_t degree, const Tensor& campos, bool prefiltered, bool debug,
                           int64_t capacity) {
  FwdOut o;
  if (means3D.dim() == 2 && means3D.size(0) == 0) {
    // no Gaussians: the upstream binding skips the rasterizer and returns its zero-initialised images (NOT the
    // background), num_rendered 0, empty state
    at::Device dev = dev_f32(means3D, "means3D", NO_CPU_R).device();
    o.color = at::zeros({3, H, W}, f32(dev)), o.depth = at::zeros({1, H, W}, f32(dev)), o.alpha = at::zeros({1, H, W}, f32(dev));
    o.radii = at::empty({0}, i32(dev));
    o.geom = at::empty({0}, u8(dev)), o.binning = at::empty({0}, u8(dev)), o.img = at::empty({0}, u8(dev));
    return o;
  }
  Scene s = make_scene(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                       projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug);
  at::Device dev = s.dev;
  o.color = at::empty({3, H, W}, f32(dev)), o.depth = at::empty({1, H, W}, f32(dev)), o.alpha = at::empty({1, H, W}, f32(dev));
  o.radii = at::empty({s.P}, i32(dev));
  DeviceGuard guard(dev);
  b3gs_stream_t stream = cur_stream(dev);
  if (capacity > 0) {
    o.geom = at::empty({(int64_t)b3gs_geometry_bytes((int32_t)s.P)}, u8(dev));
    o.binning = at::empty({(int64_t)b3gs_binning_bytes((int32_t)s.P, capacity)}, u8(dev));
    o.img = at::empty({(int64_t)b3gs_image_bytes((int32_t)W, (int32_t)H)}, u8(dev));
    o.n_dev = at::empty({1}, i32(dev));
    check(b3gs_forward_capacity(&s.sc, (char*)o.geom.data_ptr(), (char*)o.binning.data_ptr(), capacity, (char*)o.img.data_ptr(),
                                o.color.data_ptr<float>(), o.depth.data_ptr<float>(), o.alpha.data_ptr<float>(),
                                s.P ? o.radii.data_ptr<int32_t>() : nullptr, o.n_dev.data_ptr<int32_t>(), stream),
          "b3gs_forward_capacity");
    o.n = capacity;
    return o;
  }
  AllocSlot g{Tensor(), dev}, b{Tensor(), dev}, i{Tensor(), dev};
  int32_t n = 0;
  check(b3gs_forward(&s.sc, alloc_cb, &g, alloc_cb, &b, alloc_cb, &i, o.color.data_ptr<float>(), o.depth.data_ptr<float>(),
                     o.alpha.data_ptr<float>(), s.P ? o.radii.data_ptr<int32_t>() : nullptr, &n, stream), "b3gs_forward");
  o.n = n;
  o.geom = g.t, o.img = i.t;
  o.binning = b.t.defined() ? b.t : at::empty({0}, u8(dev));
  return o;
}

using OptT = c10::optional<Tensor>;     // (python None for an absent tensor, like an empty one)
static Tensor T(const OptT& o) { return o.has_value() ? *o : Tensor(); }

#define RASTER_ARGS                                                                                                      \
  const Tensor &background, const Tensor &means3D, const OptT &colors, const Tensor &opacity, const OptT &scales,          \
      const OptT &rotations, double scale_modifier, const OptT &cov3D_precomp, const Tensor &viewmatrix,                  \
      const Tensor &projmatrix, double tan_fovx, double tan_fovy, int64_t image_height, int64_t image_width,             \
      const OptT &sh, int64_t degree, const Tensor &campos, bool prefiltered, bool debug
#define RASTER_PASS                                                                                                      \
  background, means3D, T(colors), opacity, T(scales), T(rotations), scale_modifier, T(cov3D_precomp), viewmatrix,         \
      projmatrix, tan_fovx, tan_fovy, image_height, image_width, T(sh), degree, campos, prefiltered, debug

// upstream: (num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer)
static py::tuple rasterize_gaussians(RASTER_ARGS) {
  FwdOut o = forward_impl(RASTER_PASS, 0);
  return py::make_tuple(o.n, o.color, o.depth, o.alpha, o.radii, o.geom, o.binning, o.img);
}
// sync-free variant (not upstream): buffers for `capacity` instances, N stays on the device (the 9th element)
static py::tuple rasterize_gaussians_capacity(RASTER_ARGS, int64_t capacity) {
  if (capacity <= 0) throw py::value_error("capacity must be positive");
  FwdOut o = forward_impl(RASTER_PASS, capacity);
  return py::make_tuple(o.n, o.color, o.depth, o.alpha, o.radii, o.geom, o.binning, o.img, o.n_dev);
}

struct BwdOut {
  Tensor means2D, colors, opacity, means3D, cov3D, sh, scales, rot;
};

static BwdOut backward_impl(const Tensor& background, const Tensor& means3D_in, const Tensor& radii, const Tensor& colors,
                            const Tensor& scales, const Tensor& rotations, double scale_modifier, const Tensor& cov3D_precomp,
                            const Tensor& viewmatrix, const Tensor& projmatrix, double tan_fovx, double tan_fovy,
                            const Tensor& dL_dout_color, const Tensor& dL_dout_depth, const Tensor& dL_dout_alpha,
                            const Tensor& sh, int64_t degree, const Tensor& campos, const Tensor& geom, int64_t R,
                            const Tensor& binning, const Tensor& img, bool debug, const Tensor& opacities_in) {
  BwdOut o;
  int64_t P = means3D_in.size(0);
  at::Device dev = means3D_in.device();
  if (P == 0) {   // (see forward_impl: nothing was rendered)
    int64_t M0 = (sh.defined() && sh.dim() == 3) ? sh.size(1) : 0;
    o.means2D = at::zeros({0, 3}, f32(dev)), o.colors = at::zeros({0, 3}, f32(dev)), o.opacity = at::zeros({0, 1}, f32(dev));
    o.means3D = at::zeros({0, 3}, f32(dev)), o.cov3D = at::zeros({0, 6}, f32(dev)), o.sh = at::zeros({0, M0, 3}, f32(dev));
    o.scales = at::zeros({0, 3}, f32(dev)), o.rot = at::zeros({0, 4}, f32(dev));
    return o;
  }
  // (the kernels read opacity from the saved geometry state; a placeholder lets the scene struct be validated the same way)
  Tensor opac = opacities_in.defined() ? opacities_in : at::empty({P, 1}, f32(dev));
  int64_t H = dL_dout_color.size(-2), W = dL_dout_color.size(-1);
  Scene s = make_scene(background, means3D_in, colors, opac, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                       projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, false, debug);
  dev = s.dev;
  Tensor dC = dev_f32(dL_dout_color, "dL_dout_color", NO_CPU_R);
  Tensor dD = present(dL_dout_depth) ? dev_f32(dL_dout_depth, "dL_dout_depth", NO_CPU_R) : Tensor();
  Tensor dA = present(dL_dout_alpha) ? dev_f32(dL_dout_alpha, "dL_dout_alpha", NO_CPU_R) : Tensor();
  bool has_sr = s.sc.scales != nullptr;
  o.means2D = at::empty({P, 3}, f32(dev)), o.colors = at::empty({P, 3}, f32(dev)), o.opacity = at::empty({P, 1}, f32(dev));
  o.means3D = at::empty({P, 3}, f32(dev)), o.cov3D = at::empty({P, 6}, f32(dev)), o.sh = at::empty({P, s.M, 3}, f32(dev));
  o.scales = at::empty({has_sr ? P : 0, 3}, f32(dev)), o.rot = at::empty({has_sr ? P : 0, 4}, f32(dev));
  Tensor radii_i = radii.scalar_type() == at::kInt ? radii.contiguous() : radii.to(at::kInt).contiguous();
  DeviceGuard guard(dev);
  check(b3gs_backward(&s.sc, (int32_t)R, radii_i.numel() ? radii_i.data_ptr<int32_t>() : nullptr,
                      geom.numel() ? (const char*)geom.data_ptr() : nullptr, binning.numel() ? (const char*)binning.data_ptr() : nullptr,
                      img.numel() ? (const char*)img.data_ptr() : nullptr, fptr(dC), fptr(dD), fptr(dA), fptr_mut(o.means2D),
                      fptr_mut(o.colors), fptr_mut(o.opacity), fptr_mut(o.means3D), fptr_mut(o.cov3D), fptr_mut(o.sh),
                      fptr_mut(o.scales), fptr_mut(o.rot), cur_stream(dev)), "b3gs_backward");
  return o;
}

// upstream: -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
static py::tuple rasterize_gaussians_backward(const Tensor& background, const Tensor& means3D, const Tensor& radii,
                                              const OptT& colors, const OptT& scales, const OptT& rotations,
                                              double scale_modifier, const OptT& cov3D_precomp, const Tensor& viewmatrix,
                                              const Tensor& projmatrix, double tan_fovx, double tan_fovy,
                                              const Tensor& dL_dout_color, const OptT& dL_dout_depth, const OptT& dL_dout_alpha,
                                              const OptT& sh, int64_t degree, const Tensor& campos, const Tensor& geomBuffer,
                                              int64_t R, const Tensor& binningBuffer, const Tensor& imageBuffer,
                                              const OptT& alpha, bool debug, const OptT& opacities) {
  (void)alpha;
  BwdOut o = backward_impl(background, means3D, radii, T(colors), T(scales), T(rotations), scale_modifier, T(cov3D_precomp),
                           viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, T(dL_dout_depth), T(dL_dout_alpha), T(sh),
                           degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, T(opacities));
  return py::make_tuple(o.means2D, o.colors, o.opacity, o.means3D, o.cov3D, o.sh, o.scales, o.rot);
}

static Tensor mark_visible(const Tensor& means3D, const Tensor& viewmatrix, const Tensor& projmatrix) {
  Tensor m = dev_f32(means3D, "means3D", NO_CPU_R), vm = dev_f32(viewmatrix, "viewmatrix", NO_CPU_R),
         pm = dev_f32(projmatrix, "projmatrix", NO_CPU_R);
  Tensor present_ = at::zeros({m.size(0)}, at::TensorOptions().dtype(at::kBool).device(m.device()));
  DeviceGuard guard(m.device());
  check(b3gs_mark_visible((int32_t)m.size(0), fptr(m), fptr(vm), fptr(pm), m.size(0) ? (uint8_t*)present_.data_ptr<bool>() : nullptr,
                          cur_stream(m.device())), "b3gs_mark_visible");
  return present_;
}

// ---- _RasterizeGaussians: the autograd node of the module surface -----------------------------------------------------
struct RasterizeFn : public torch::autograd::Function<RasterizeFn> {
  // -> color, radii, depth, alpha, n_info (exact forward: N as a host int32 scalar; capacity > 0: N on the device)
  static tensor_list forward(AutogradContext* ctx, const Tensor& means3D, const Tensor& means2D, const Tensor& sh,
                             const Tensor& colors_precomp, const Tensor& opacities, const Tensor& scales, const Tensor& rotations,
                             const Tensor& cov3Ds_precomp, const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix,
                             const Tensor& campos, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                             int64_t sh_degree, bool prefiltered, bool debug, int64_t capacity) {
    (void)means2D;
    FwdOut o = forward_impl(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3Ds_precomp, viewmatrix,
                            projmatrix, tanfovx, tanfovy, H, W, sh, sh_degree, campos, prefiltered, debug, capacity);
    Tensor n_info = o.n_dev.defined() ? o.n_dev : at::scalar_tensor((int64_t)o.n, at::TensorOptions().dtype(at::kInt));
    ctx->save_for_backward({colors_precomp, means3D, scales, rotations, cov3Ds_precomp, o.radii, sh, o.geom, o.binning, o.img,
                            o.alpha, bg, viewmatrix, projmatrix, campos});
    ctx->saved_data["i"] = std::vector<int64_t>{H, W, sh_degree, debug, o.n};
    ctx->saved_data["f"] = std::vector<double>{tanfovx, tanfovy, scale_modifier};
    ctx->mark_non_differentiable({o.radii, n_info});
    // an output nobody differentiates (depth / alpha of most losses) arrives undefined, not as an image of zeros the blend
    // backward would have to read
    ctx->set_materialize_grads(false);
    return {o.color, o.radii, o.depth, o.alpha, n_info};
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list g) {
    auto sv = ctx->get_saved_variables();
    const Tensor &colors_precomp = sv[0], &means3D = sv[1], &scales = sv[2], &rotations = sv[3], &cov3Ds = sv[4], &radii = sv[5],
                 &sh = sv[6], &geom = sv[7], &binning = sv[8], &img = sv[9];
    auto iv = ctx->saved_data["i"].toIntVector();
    auto fv = ctx->saved_data["f"].toDoubleVector();
    Tensor gc = g[0].defined() ? g[0] : at::zeros({3, iv[0], iv[1]}, f32(means3D.device()));
    BwdOut o = backward_impl(sv[11], means3D, radii, colors_precomp, scales, rotations, fv[2], cov3Ds, sv[12], sv[13], fv[0],
                             fv[1], gc, g[2], g[3], sh, iv[2], sv[14], geom, iv[4], binning, img, iv[3] != 0, Tensor());
    auto pick = [&](const Tensor& t, bool had_input, size_t i) { return (had_input && ctx->needs_input_grad(i)) ? t : Tensor(); };
    tensor_list out(21);
    out[0] = pick(o.means3D, true, 0), out[1] = pick(o.means2D, true, 1), out[2] = pick(o.sh, present(sh), 2);
    out[3] = pick(o.colors, present(colors_precomp), 3), out[4] = pick(o.opacity, true, 4);
    out[5] = pick(o.scales, present(scales), 5), out[6] = pick(o.rot, present(rotations), 6);
    out[7] = pick(o.cov3D, present(cov3Ds), 7);
    return out;
  }
};

static tensor_list rasterize_gaussians_autograd(const Tensor& means3D, const Tensor& means2D, const Tensor& sh,
                                                const Tensor& colors_precomp, const Tensor& opacities, const Tensor& scales,
                                                const Tensor& rotations, const Tensor& cov3Ds_precomp, const Tensor& bg,
                                                const Tensor& viewmatrix, const Tensor& projmatrix, const Tensor& campos,
                                                int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                                                int64_t sh_degree, bool prefiltered, bool debug, int64_t capacity) {
  return RasterizeFn::apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, bg, viewmatrix,
                            projmatrix, campos, H, W, tanfovx, tanfovy, scale_modifier, sh_degree, prefiltered, debug, capacity);
}

// ---- the raw-parameter node's launch assembly --------------------------------------------------------------------------
// One view of raw parameters: the camera half of B3gsScene + the six parameter pointers.  Keeps the four small camera
// tensors alive; the parameters are kept by the autograd node that owns this object (its saved tensors).
struct RawView {
  B3gsScene sc{};
  B3gsRawParams rp{};
  Tensor bg, vm, pm, campos;
  at::Device dev{at::kCPU};
  int64_t P = 0, K = 0, W = 0, H = 0;
};

// -> (view, geometry, image state, out [5,H,W], its views colour [3,H,W] | depth [1,H,W] | alpha [1,H,W], radii,
//     visible or None, binning)
static py::tuple raw_prepare(const Tensor& xyz, const Tensor& f_dc, const Tensor& f_rest, const Tensor& scaling,
                             const Tensor& rotation, const Tensor& opacity, const Tensor& bg, const Tensor& viewmatrix,
                             const Tensor& projmatrix, const Tensor& campos, int64_t W, int64_t H, double tanfovx, double tanfovy,
                             double scale_modifier, int64_t sh_degree, bool debug, int64_t capacity, bool want_visible,
                             bool nan_fill) {
  auto v = std::make_shared<RawView>();
  v->dev = xyz.device();
  v->P = xyz.size(0), v->K = f_dc.size(1) + f_rest.size(1), v->W = W, v->H = H;
  v->bg = bg, v->vm = viewmatrix, v->pm = projmatrix, v->campos = campos;
  B3gsScene& c = v->sc;
  c.P = (int32_t)v->P, c.D = (int32_t)sh_degree, c.M = (int32_t)v->K, c.W = (int32_t)W, c.H = (int32_t)H;
  c.tan_fovx = (float)tanfovx, c.tan_fovy = (float)tanfovy, c.scale_modifier = (float)scale_modifier;
  c.prefiltered = 0, c.debug = debug ? 1 : 0;
  c.background = bg.data_ptr<float>(), c.viewmatrix = viewmatrix.data_ptr<float>(), c.projmatrix = projmatrix.data_ptr<float>();
  c.campos = campos.data_ptr<float>();
  v->rp.xyz = fptr(xyz), v->rp.features_dc = fptr(f_dc), v->rp.features_rest = fptr(f_rest);
  v->rp.scaling = fptr(scaling), v->rp.rotation = fptr(rotation), v->rp.opacity = fptr(opacity);
  at::Device dev = v->dev;
  Tensor geom = at::empty({(int64_t)b3gs_geometry_bytes((int32_t)v->P)}, u8(dev));
  // recycled allocator memory: B3gsForwardView::fresh_image tells the library to read nothing from it
  Tensor img = at::empty({(int64_t)b3gs_image_bytes((int32_t)W, (int32_t)H)}, u8(dev));
  Tensor out = at::empty({5, H, W}, f32(dev));
  if (nan_fill) out.fill_(std::numeric_limits<float>::quiet_NaN());   // see rasterizer._LazyOut: nobody may read these unnoticed
  Tensor radii = at::empty({v->P}, i32(dev));
  py::object vis = py::none();
  if (want_visible) vis = py::cast(at::empty({v->P}, at::TensorOptions().dtype(at::kBool).device(dev)));
  Tensor binning = at::empty({(int64_t)b3gs_binning_bytes((int32_t)v->P, capacity)}, u8(dev));
  return py::make_tuple(v, geom, img, out, out.narrow(0, 0, 3), out.narrow(0, 3, 1), out.narrow(0, 4, 1), radii, vis, binning);
}

static Tensor raw_binning(const std::shared_ptr<RawView>& v, int64_t capacity) {
  return at::empty({(int64_t)b3gs_binning_bytes((int32_t)v->P, capacity)}, u8(v->dev));
}

// views: [(view, geometry, binning, capacity, image, out, radii, words int32[4] = [N, overflow word, key mismatch, spare],
//          visible | None, depth_key_bits, depth_order_from, hint geometry | None, hint_trusted, fresh_image, seg1_fraction)]
static void raw_forward_launch(const py::list& views, int64_t stream_id) {
  size_t n = views.size();
  if (n == 0 || n > 8) throw py::value_error("raw_forward_launch: 1..8 views");
  B3gsForwardView fv[8];
  std::shared_ptr<RawView> v0;
  std::vector<Tensor> keep;
  keep.reserve(n * 8);
  for (size_t k = 0; k < n; ++k) {
    py::tuple t = views[k].cast<py::tuple>();
    auto v = t[0].cast<std::shared_ptr<RawView>>();
    if (k == 0) v0 = v;
    Tensor geom = t[1].cast<Tensor>(), binning = t[2].cast<Tensor>(), img = t[4].cast<Tensor>(), out = t[5].cast<Tensor>(),
           radii = t[6].cast<Tensor>(), words = t[7].cast<Tensor>();
    for (const Tensor& x : {geom, binning, img, out, radii, words}) keep.push_back(x);
    B3gsForwardView& f = fv[k];
    f = B3gsForwardView{};
    f.view = &v->sc;
    f.geometry = (char*)geom.data_ptr(), f.binning = (char*)binning.data_ptr(), f.image = (char*)img.data_ptr();
    f.binning_capacity = t[3].cast<int64_t>();
    float* o = out.data_ptr<float>();
    int64_t hw = v->H * v->W;
    f.out_color = o, f.out_depth = o + 3 * hw, f.out_alpha = o + 4 * hw;
    f.radii = v->P ? radii.data_ptr<int32_t>() : nullptr;
    int32_t* w = words.data_ptr<int32_t>();
    f.device_num_rendered = w, f.overflow_flag = w + 1, f.high_water = nullptr;
    if (!t[8].is_none()) {
      Tensor vis = t[8].cast<Tensor>();
      keep.push_back(vis);
      f.visible = v->P ? (uint8_t*)vis.data_ptr<bool>() : nullptr;     // (torch.bool is one byte, 0 / 1)
    }
    f.depth_key_bits = t[9].cast<int32_t>();
    f.depth_order_from = t[10].cast<int32_t>();
    if (!t[11].is_none()) {
      Tensor hint = t[11].cast<Tensor>();
      keep.push_back(hint);
      f.depth_order_hint = (const char*)hint.data_ptr(), f.hint_mismatch = w + 2;
    }
    f.hint_trusted = t[12].cast<int32_t>();
    f.fresh_image = t[13].cast<int32_t>();
    f.seg1_fraction = t[14].cast<float>();
  }
  DeviceGuard guard(v0->dev);
  check(b3gs_forward_raw_batch((int32_t)n, fv, &v0->rp, 3, (b3gs_stream_t)(intptr_t)stream_id), "b3gs_forward_raw_batch");
}

// jobs: [(view, radii, geometry, binning, image, dL_dcolor, dL_ddepth | None, dL_dalpha | None, wants dL_dmeans2D, capacity)]
// scratch: >= min(len(jobs), 8) zeroed float buffers of b3gs_backward_scratch_floats(P) (left zero); grads: the six
// parameter-shaped gradient tensors.  overwrite: the first chunk stores, later chunks add.  -> [dL_dmeans2D | None per job]
static py::list raw_backward_launch(const py::list& jobs, const std::vector<Tensor>& scratch, const std::vector<Tensor>& grads,
                                    bool overwrite, int64_t stream_id) {
  size_t n = jobs.size();
  if (n == 0) return py::list();
  if (grads.size() != 6) throw py::value_error("raw_backward_launch: six gradient tensors");
  if (scratch.size() < std::min<size_t>(n, 8)) throw py::value_error("raw_backward_launch: not enough scratch buffers");
  B3gsRawGrads gr{};
  std::vector<Tensor> gm(grads);
  gr.xyz = fptr_mut(gm[0]), gr.features_dc = fptr_mut(gm[1]), gr.features_rest = fptr_mut(gm[2]);
  gr.scaling = fptr_mut(gm[3]), gr.rotation = fptr_mut(gm[4]), gr.opacity = fptr_mut(gm[5]);
  gr.touched_rows = nullptr;
  py::list m2d_out;
  std::shared_ptr<RawView> v0 = jobs[0].cast<py::tuple>()[0].cast<std::shared_ptr<RawView>>();
  DeviceGuard guard(v0->dev);
  b3gs_stream_t s = (b3gs_stream_t)(intptr_t)stream_id;
  for (size_t c0 = 0; c0 < n; c0 += 8) {
    size_t m = std::min<size_t>(8, n - c0);
    B3gsBlendView bv[8];
    B3gsFusedView av[8];
    std::vector<Tensor> keep;
    for (size_t k = 0; k < m; ++k) {
      py::tuple t = jobs[c0 + k].cast<py::tuple>();
      auto v = t[0].cast<std::shared_ptr<RawView>>();
      Tensor radii = t[1].cast<Tensor>(), geom = t[2].cast<Tensor>(), binning = t[3].cast<Tensor>(), img = t[4].cast<Tensor>(),
             gc = t[5].cast<Tensor>();
      Tensor gd = t[6].is_none() ? Tensor() : t[6].cast<Tensor>(), ga = t[7].is_none() ? Tensor() : t[7].cast<Tensor>();
      Tensor g_m2d = t[8].cast<bool>() ? at::empty({v->P, 3}, f32(v->dev)) : Tensor();
      for (const Tensor& x : {radii, geom, binning, img, gc, gd, ga}) keep.push_back(x);
      if (g_m2d.defined()) m2d_out.append(g_m2d); else m2d_out.append(py::none());
      Tensor sc = scratch[k];
      B3gsBlendView& b = bv[k];
      b = B3gsBlendView{};
      b.view = &v->sc;
      b.geometry = (const char*)geom.data_ptr(), b.binning = (const char*)binning.data_ptr(), b.image = (const char*)img.data_ptr();
      b.dL_dcolor = fptr(gc), b.dL_ddepth = fptr(gd), b.dL_dalpha = fptr(ga);
      b.scratch = sc.data_ptr<float>(), b.binning_capacity = t[9].cast<int64_t>();
      B3gsFusedView& a = av[k];
      a = B3gsFusedView{};
      a.view = &v->sc;
      a.radii = v->P ? radii.data_ptr<int32_t>() : nullptr, a.geometry = (const char*)geom.data_ptr(), a.scratch = sc.data_ptr<float>();
      a.dL_dmeans2D = fptr_mut(g_m2d), a.densify_stats = 0;
    }
    check(b3gs_blend_backward_batch((int32_t)m, bv, s), "b3gs_blend_backward_batch");
    // overwrite mode: every row of every gradient tensor is stored (zeros for Gaussians without a contribution);
    // accumulate mode: only the rows that received something are touched
    check(b3gs_backward_raw_accumulate((int32_t)m, av, &v0->rp, &gr, (overwrite && c0 == 0) ? 1 : 0, nullptr, s),
          "b3gs_backward_raw_accumulate");
  }
  return m2d_out;
}

void bind_raster(py::module_& m) {
  const char* fwd_doc =
      "(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, "
      "tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug) -> (num_rendered, color[3,H,W], depth[1,H,W], "
      "alpha[1,H,W], radii[P] int32, geomBuffer, binningBuffer, imgBuffer)";
  m.def("rasterize_gaussians", &rasterize_gaussians, fwd_doc);
  m.def("rasterize_gaussians_capacity", &rasterize_gaussians_capacity);
  m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward, py::arg("background"), py::arg("means3D"), py::arg("radii"),
        py::arg("colors"), py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"),
        py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("dL_dout_color"),
        py::arg("dL_dout_depth"), py::arg("dL_dout_alpha"), py::arg("sh"), py::arg("degree"), py::arg("campos"),
        py::arg("geomBuffer"), py::arg("R"), py::arg("binningBuffer"), py::arg("imageBuffer"), py::arg("alpha"), py::arg("debug"),
        py::arg("opacities") = py::none());
  m.def("mark_visible", &mark_visible);
  m.def("rasterize_gaussians_autograd", &rasterize_gaussians_autograd);
  py::class_<RawView, std::shared_ptr<RawView>>(m, "RawView")
      .def_readonly("P", &RawView::P)
      .def_readonly("K", &RawView::K)
      .def_readonly("W", &RawView::W)
      .def_readonly("H", &RawView::H);
  m.def("raw_prepare", &raw_prepare);
  m.def("raw_binning", &raw_binning);
  m.def("raw_forward_launch", &raw_forward_launch);
  m.def("raw_backward_launch", &raw_backward_launch);
  m.def("geometry_bytes", [](int64_t P) { return (int64_t)b3gs_geometry_bytes((int32_t)P); });
  m.def("image_bytes", [](int64_t W, int64_t H) { return (int64_t)b3gs_image_bytes((int32_t)W, (int32_t)H); });
  m.def("binning_bytes", [](int64_t P, int64_t n) { return (int64_t)b3gs_binning_bytes((int32_t)P, n); });
  m.def("backward_scratch_floats", [](int64_t P) { return (int64_t)b3gs_backward_scratch_floats((int32_t)P); });
}

}  // namespace b3
