// Densify / clone / split / prune with optimiser-state surgery: the net effect of the reference's
// densify_and_prune (scene/gaussian_model.py:393-407 -> densify_and_clone :379-391, densify_and_split :355-377,
// prune_points :292-305, cat/prune of the Adam state :271-345) as classify -> (prefix sums) -> ONE scatter pass,
// instead of ~100 boolean-mask gathers and torch.cat calls over 6 parameters x (value, exp_avg, exp_avg_sq).
//
// Per original Gaussian i (g = xyz_gradient_accum/denom with NaN -> 0, s = exp(_scaling), smax = max(s)):
//   clone  iff |g| >= thr and smax <= percent_dense*extent      -> an identical copy is appended (Adam state 0)
//   split  iff  g  >= thr and smax >  percent_dense*extent      -> the original is removed, two children appended:
//             xyz' = R(q/|q|) (s * noise_k) + xyz, _scaling' = log(s / 1.6), everything else copied (Adam state 0)
//   prune  (applied to originals, clones and children alike) iff sigmoid(_opacity) < min_opacity, or -- only when
//          a screen-size threshold is given -- max(exp(_scaling)) > 0.1*extent.  The reference also forms
//          max_radii2D > max_screen_size, but it zeroes max_radii2D in densification_postfix (:345) before the
//          prune mask is built (:401), so that criterion never fires; reproduced by not evaluating it.
// Output order = the reference's: kept originals | kept clones | kept children k=0 | kept children k=1, each in
// index order.
#include "b3gs_internal.h"

namespace {

struct DIO {
  B3gsDensifyIO io;
};

__device__ __forceinline__ bool pruned(float opacity_raw, float smax, float min_opacity, float extent, int world_rule) {
  const float op = 1.0f / (1.0f + expf(-opacity_raw));
  return (op < min_opacity) || (world_rule && smax > 0.1f * extent);
}

__global__ void __launch_bounds__(256) densify_classify_kernel(B3gsDensifyIO io, int32_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= io.P) return;
  float g = io.xyz_gradient_accum[i] / io.denom[i];
  if (g != g) g = 0.0f;
  const float* ls = io.param[3] + 3 * (size_t)i;
  const float s0 = expf(ls[0]), s1 = expf(ls[1]), s2 = expf(ls[2]);
  const float smax = fmaxf(s0, fmaxf(s1, s2));
  const float limit = io.percent_dense * io.extent;
  const bool clone = (fabsf(g) >= io.grad_threshold) && (smax <= limit);
  const bool split = (g >= io.grad_threshold) && (smax > limit);
  const int world_rule = io.max_screen_size > 0.0f;
  const float op_raw = io.param[5][i];
  const bool drop = pruned(op_raw, smax, io.min_opacity, io.extent, world_rule);
  // children: scaling' = log(s / 1.6), evaluated through exp like the reference's get_scaling
  const float c0 = expf(logf(s0 / 1.6f)), c1 = expf(logf(s1 / 1.6f)), c2 = expf(logf(s2 / 1.6f));
  const bool drop_child = pruned(op_raw, fmaxf(c0, fmaxf(c1, c2)), io.min_opacity, io.extent, world_rule);
  flags[i] = (int)(!split && !drop) | ((int)(clone && !drop) << 1) | ((int)(split && !drop_child) << 2);
}

struct Outs {
  float* param[6];
  float* exp_avg[6];
  float* exp_avg_sq[6];
};

__device__ __forceinline__ void copy_row(const B3gsDensifyIO& io, const Outs& o, int t, int width, size_t src, size_t dst,
                                         bool keep_state) {
  const float* p = io.param[t] + src * width;
  float* q = o.param[t] + dst * width;
  for (int k = 0; k < width; k++) q[k] = p[k];
  if (o.exp_avg[t]) {
    float* m = o.exp_avg[t] + dst * width;
    float* v = o.exp_avg_sq[t] + dst * width;
    if (keep_state && io.exp_avg[t]) {
      const float* m0 = io.exp_avg[t] + src * width;
      const float* v0 = io.exp_avg_sq[t] + src * width;
      for (int k = 0; k < width; k++) { m[k] = m0[k]; v[k] = v0[k]; }
    } else {
      for (int k = 0; k < width; k++) { m[k] = 0.0f; v[k] = 0.0f; }
    }
  }
}

__global__ void __launch_bounds__(256)
    densify_scatter_kernel(B3gsDensifyIO io, const int32_t* __restrict__ flags, const int32_t* __restrict__ off_keep,
                           const int32_t* __restrict__ off_clone, const int32_t* __restrict__ off_split, int32_t n_keep,
                           int32_t n_clone, int32_t n_split, const float* __restrict__ noise, Outs o) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= io.P) return;
  const int f = flags[i];
  if (f == 0) return;
  const int widths[6] = {3, 3, 3 * (io.M - 1), 3, 4, 1};
  if (f & 1)
    for (int t = 0; t < 6; t++) copy_row(io, o, t, widths[t], (size_t)i, (size_t)off_keep[i], true);
  if (f & 2)
    for (int t = 0; t < 6; t++) copy_row(io, o, t, widths[t], (size_t)i, (size_t)n_keep + off_clone[i], false);
  if (f & 4) {
    const float* ls = io.param[3] + 3 * (size_t)i;
    const float s[3] = {expf(ls[0]), expf(ls[1]), expf(ls[2])};
    const float4 qr = reinterpret_cast<const float4*>(io.param[4])[i];
    const float inv = 1.0f / sqrtf(((qr.x * qr.x + qr.y * qr.y) + qr.z * qr.z) + qr.w * qr.w);
    const float r = qr.x * inv, x = qr.y * inv, y = qr.z * inv, z = qr.w * inv;
    const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
    const float* c = io.param[0] + 3 * (size_t)i;
    for (int k = 0; k < 2; k++) {
      const size_t dst = (size_t)n_keep + n_clone + (size_t)k * n_split + off_split[i];
      for (int t = 0; t < 6; t++) copy_row(io, o, t, widths[t], (size_t)i, dst, false);
      const float* nz = noise + ((size_t)k * io.P + i) * 3;
      const float v0 = s[0] * nz[0], v1 = s[1] * nz[1], v2 = s[2] * nz[2];
      float* xo = o.param[0] + 3 * dst;
      xo[0] = ((R[0] * v0 + R[1] * v1) + R[2] * v2) + c[0];
      xo[1] = ((R[3] * v0 + R[4] * v1) + R[5] * v2) + c[1];
      xo[2] = ((R[6] * v0 + R[7] * v1) + R[8] * v2) + c[2];
      float* so = o.param[3] + 3 * dst;
      so[0] = logf(s[0] / 1.6f); so[1] = logf(s[1] / 1.6f); so[2] = logf(s[2] / 1.6f);
    }
  }
}

int check_io(const B3gsDensifyIO* io) {
  if (!io || io->P < 0 || io->M < 1) return b3gs_fail(B3GS_ERR_ARG, "densify", "NULL io, negative P or M < 1");
  for (int t = 0; t < 6; t++)
    if (!io->param[t] && !(t == 2 && io->M == 1) && io->P > 0) return b3gs_fail(B3GS_ERR_ARG, "densify", "NULL parameter tensor");
  if (io->P > 0 && (!io->xyz_gradient_accum || !io->denom)) return b3gs_fail(B3GS_ERR_ARG, "densify", "NULL statistics array");
  return B3GS_OK;
}

}  // namespace

extern "C" int b3gs_densify_classify(const B3gsDensifyIO* io, int32_t* flags, b3gs_stream_t stream) {
  int rc = check_io(io);
  if (rc) return rc;
  if (io->P == 0) return B3GS_OK;
  if (!flags) return b3gs_fail(B3GS_ERR_ARG, "b3gs_densify_classify", "flags is NULL");
  hipLaunchKernelGGL(densify_classify_kernel, dim3((io->P + 255) / 256), dim3(256), 0, (hipStream_t)stream, *io, flags);
  return b3gs_launch_status("b3gs_densify_classify");
}

extern "C" int b3gs_densify_scatter(const B3gsDensifyIO* io, const int32_t* flags, const int32_t* off_keep,
                                    const int32_t* off_clone, const int32_t* off_split, int32_t n_keep, int32_t n_clone,
                                    int32_t n_split, const float* noise, float* const* out_param,
                                    float* const* out_exp_avg, float* const* out_exp_avg_sq, b3gs_stream_t stream) {
  int rc = check_io(io);
  if (rc) return rc;
  if (io->P == 0) return B3GS_OK;
  if (!flags || !off_keep || !off_clone || !off_split || !out_param || n_keep < 0 || n_clone < 0 || n_split < 0 ||
      (n_split > 0 && !noise))
    return b3gs_fail(B3GS_ERR_ARG, "b3gs_densify_scatter", "NULL flags / offsets / outputs, negative count, or split without noise");
  Outs o;
  for (int t = 0; t < 6; t++) {
    o.param[t] = out_param[t];
    o.exp_avg[t] = out_exp_avg ? out_exp_avg[t] : nullptr;
    o.exp_avg_sq[t] = out_exp_avg_sq ? out_exp_avg_sq[t] : nullptr;
    if (!o.param[t] && !(t == 2 && io->M == 1) && (n_keep + n_clone + n_split) > 0)
      return b3gs_fail(B3GS_ERR_ARG, "b3gs_densify_scatter", "NULL output parameter tensor");
    if ((o.exp_avg[t] == nullptr) != (o.exp_avg_sq[t] == nullptr))
      return b3gs_fail(B3GS_ERR_ARG, "b3gs_densify_scatter", "exp_avg / exp_avg_sq outputs must both be given or both be NULL");
  }
  hipLaunchKernelGGL(densify_scatter_kernel, dim3((io->P + 255) / 256), dim3(256), 0, (hipStream_t)stream, *io, flags,
                     off_keep, off_clone, off_split, n_keep, n_clone, n_split, noise, o);
  return b3gs_launch_status("b3gs_densify_scatter");
}
