// Fused Adam over the (up to 8) parameter tensors of the Gaussian model in ONE launch: the update the
// reference performs with torch.optim.Adam(eps=1e-15), six parameter groups with their own learning
// rates (scene/gaussian_model.py:154-167, train.py:196-198), optionally followed by the per-iteration
// opacity decay  o <- logit(sigmoid(o) * factor)  (scene/gaussian_model.py:307-309, train.py:171-173).
// torch's fused Adam issues 2 kernels per parameter group (0.35 ms per iteration at 1M Gaussians);
// this pass is HBM-bound at 28 B per parameter float (p, g, m, v read; p, m, v written).
// The step counter lives on the device so the launch can be replayed from a HIP graph.
#include "b3gs_internal.h"

namespace {

struct AdamSegs {
  int n;
  B3gsAdamSegment s[8];
  uint32_t start[9];  // cumulative element offsets
};

__global__ void __launch_bounds__(256)
    adam_kernel(AdamSegs segs, const int32_t* __restrict__ step_ptr, float beta1, float beta2, float eps,
                float opacity_decay, int opacity_seg, int decay_first) {
  const float t = (float)(*step_ptr + 1);
  const float bc1 = 1.0f - powf(beta1, t);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, t));
  const uint32_t total = segs.start[segs.n];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < 8; j++)
      if (j < segs.n && i >= segs.start[j]) k = j;
    float *p, *m, *v;
    const float* g;
    float lr;
    uint32_t base;
    // per-element select of the segment fields (k differs between lanes at segment borders)
    p = segs.s[0].param; g = segs.s[0].grad; m = segs.s[0].exp_avg; v = segs.s[0].exp_avg_sq; lr = segs.s[0].lr; base = 0;
#pragma unroll
    for (int j = 1; j < 8; j++)
      if (j == k) { p = segs.s[j].param; g = segs.s[j].grad; m = segs.s[j].exp_avg; v = segs.s[j].exp_avg_sq; lr = segs.s[j].lr; base = segs.start[j]; }
    const uint32_t e = i - base;
    const float grad = g[e];
    const float mi = beta1 * m[e] + (1.0f - beta1) * grad;
    const float vi = beta2 * v[e] + (1.0f - beta2) * grad * grad;
    m[e] = mi;
    v[e] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float delta = (lr / bc1) * (mi / denom);
    const bool decay = opacity_decay > 0.0f && k == opacity_seg;
    float pi = p[e];
    if (!(decay && decay_first)) pi -= delta;
    if (decay) {
      const float op = opacity_decay / (1.0f + expf(-pi));   // sigmoid(o) * factor
      pi = logf(op / (1.0f - op));                           // inverse sigmoid
      // reference order (train.py:171-173 before optimizer.step() at :196-198): the update computed from the
      // gradient at the un-decayed value is subtracted from the DECAYED logit
      if (decay_first) pi -= delta;
    }
    p[e] = pi;
  }
}

__global__ void bump_step(int32_t* step_ptr) { *step_ptr += 1; }

}  // namespace

extern "C" int b3gs_adam_step(int32_t nseg, const B3gsAdamSegment* segs, int32_t* device_step, float beta1, float beta2,
                              float eps, float opacity_decay, int32_t opacity_segment, int32_t opacity_decay_first,
                              int32_t bump_step_after, b3gs_stream_t stream) {
  if (nseg < 0 || nseg > 8 || (nseg > 0 && !segs) || !device_step)
    return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step", "0..8 segments and a device step counter are required");
  AdamSegs a;
  a.n = nseg;
  uint64_t tot = 0;
  for (int k = 0; k < nseg; k++) {
    // an empty segment (e.g. features_rest at SH degree 0: [P,0,3]) may carry NULL pointers
    if (segs[k].count < 0 ||
        (segs[k].count > 0 && (!segs[k].param || !segs[k].grad || !segs[k].exp_avg || !segs[k].exp_avg_sq)))
      return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step", "negative count or NULL pointer in a non-empty segment");
    a.s[k] = segs[k];
    a.start[k] = (uint32_t)tot;
    tot += (uint64_t)segs[k].count;
  }
  if (tot > 0xFFFFFFFFull) return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step", "more than 2^32 parameter floats in one call");
  a.start[nseg] = (uint32_t)tot;
  for (int k = nseg + 1; k < 9; k++) a.start[k] = (uint32_t)tot;
  if (tot == 0) {   // nothing to update (a rank whose shard is all padding); the step still counts
    if (bump_step_after) hipLaunchKernelGGL(bump_step, dim3(1), dim3(1), 0, (hipStream_t)stream, device_step);
    return b3gs_launch_status("b3gs_adam_step");
  }
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((tot + 255) / 256 < 256u * 32u ? (tot + 255) / 256 : 256u * 32u);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, a, device_step, beta1, beta2, eps, opacity_decay,
                     opacity_segment, opacity_decay_first);
  if (bump_step_after) hipLaunchKernelGGL(bump_step, dim3(1), dim3(1), 0, s, device_step);
  return b3gs_launch_status("b3gs_adam_step");
}
