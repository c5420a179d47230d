// Fused Adam over the (up to 8) parameter tensors of the Gaussian model in ONE launch: the update the
// reference performs with torch.optim.Adam(eps=1e-15), six parameter groups with their own learning
// rates (scene/gaussian_model.py:154-167, train.py:196-198), optionally followed by the per-iteration
// opacity decay  o <- logit(sigmoid(o) * factor)  (scene/gaussian_model.py:307-309, train.py:171-173).
// torch's fused Adam issues 2 kernels per parameter group (0.35 ms per iteration at 1M Gaussians);
// this pass is HBM-bound at 28 B per parameter float (p, g, m, v read; p, m, v written) -- 24 B for the rows a
// sparse-row gradient slab marks as untouched (row_mask).
// The step counter lives on the device so the launch can be replayed from a HIP graph.
#include "b3gs_internal.h"

namespace {

constexpr uint32_t B3GS_ADAM_SLOTS = 64u, B3GS_ADAM_SLOT_STRIDE = 32u;   // == B3GS_ADAM_STEP_WORDS of the header
static_assert(2u + B3GS_ADAM_SLOTS * B3GS_ADAM_SLOT_STRIDE == B3GS_ADAM_STEP_WORDS, "step words");

struct AdamSegs {
  int n;
  B3gsAdamSegment s[8];
  uint32_t start[9];  // cumulative element offsets
};

// One element's update; returns the new parameter value.
__device__ __forceinline__ float adam_one(float pi, float grad, float& mi, float& vi, float beta1, float beta2, float eps,
                                          float bc1, float bc2_sqrt, float lr, bool decay, float opacity_decay,
                                          int decay_first) {
  mi = beta1 * mi + (1.0f - beta1) * grad;
  vi = beta2 * vi + (1.0f - beta2) * grad * grad;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  const float delta = (lr / bc1) * (mi / denom);
  if (!(decay && decay_first)) pi -= delta;
  if (decay) {
    const float op = opacity_decay / (1.0f + expf(-pi));   // sigmoid(o) * factor
    pi = logf(op / (1.0f - op));                           // inverse sigmoid
    // reference order (train.py:171-173 before optimizer.step() at :196-198): the update computed from the
    // gradient at the un-decayed value is subtracted from the DECAYED logit
    if (decay_first) pi -= delta;
  }
  return pi;
}

// VEC = 4: every segment pointer is 16-byte aligned and every count a multiple of 4 (one thread = one float4 of one
// segment); VEC = 1 otherwise.  With a row mask the gradient of an element whose Gaussian received nothing this
// iteration is 0 without being read (its slab row is stale, B3gsRawGrads::touched_rows).
template <int VEC>
__global__ void __launch_bounds__(256)
    adam_kernel(AdamSegs segs, int32_t* __restrict__ step_ptr, int bump, float beta1, float beta2, float eps,
                float opacity_decay, int opacity_seg, int decay_first, const unsigned long long* __restrict__ row_mask,
                const int32_t* __restrict__ skip_if_nonzero, int host_step) {
  // a step rendered from truncated tile lists (B3gsForwardView::overflow_flag) is dropped here, on the device
  if (skip_if_nonzero && *skip_if_nonzero != 0) return;
  // host_step > 0 (b3gs_adam_step_at): the caller counts the steps itself, like torch.optim.Adam's state["step"]
  const float t = host_step > 0 ? (float)host_step : (float)(*step_ptr + 1);
  const float bc1 = 1.0f - powf(beta1, t);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, t));
  const uint32_t total = segs.start[segs.n] / VEC;
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
    const uint32_t i = it * VEC;
    int k = 0;
#pragma unroll
    for (int j = 1; j < 8; j++)
      if (j < segs.n && i >= segs.start[j]) k = j;
    float *p, *m, *v;
    const float* g;
    float lr;
    uint32_t base, rl, row0;
    // per-element select of the segment fields (k differs between lanes at segment borders)
    const float* lr_dev = segs.s[0].lr_dev;
    p = segs.s[0].param; g = segs.s[0].grad; m = segs.s[0].exp_avg; v = segs.s[0].exp_avg_sq; lr = segs.s[0].lr; base = 0;
    rl = (uint32_t)segs.s[0].row_len; row0 = (uint32_t)segs.s[0].first_row;
#pragma unroll
    for (int j = 1; j < 8; j++)
      if (j == k) {
        p = segs.s[j].param; g = segs.s[j].grad; m = segs.s[j].exp_avg; v = segs.s[j].exp_avg_sq; lr = segs.s[j].lr;
        base = segs.start[j]; rl = (uint32_t)segs.s[j].row_len; row0 = (uint32_t)segs.s[j].first_row;
        lr_dev = segs.s[j].lr_dev;
      }
    if (lr_dev) lr = *lr_dev;   // the learning rate of a graph-replayed step (B3gsAdamSegment::lr_dev)
    const uint32_t e = i - base;
    const bool decay = opacity_decay > 0.0f && k == opacity_seg;
    bool live[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) live[j] = true;
    bool any = true;
    if (row_mask && rl) {
      uint32_t q = e / rl, r = e - q * rl;   // one division per VEC elements, the others step through the rows
      q += row0;
      any = false;
#pragma unroll
      for (int j = 0; j < VEC; j++) {
        live[j] = (row_mask[q >> 6] >> (q & 63u)) & 1ull;
        any |= live[j];
        if (++r == rl) { r = 0; q++; }
      }
    }
    if (VEC == 4) {
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (any) gv = *reinterpret_cast<const float4*>(g + e);
      float4 mv = *reinterpret_cast<const float4*>(m + e), vv = *reinterpret_cast<const float4*>(v + e);
      float4 pv = *reinterpret_cast<const float4*>(p + e);
      pv.x = adam_one(pv.x, live[0] ? gv.x : 0.f, mv.x, vv.x, beta1, beta2, eps, bc1, bc2_sqrt, lr, decay, opacity_decay, decay_first);
      pv.y = adam_one(pv.y, live[1 % VEC] ? gv.y : 0.f, mv.y, vv.y, beta1, beta2, eps, bc1, bc2_sqrt, lr, decay, opacity_decay, decay_first);
      pv.z = adam_one(pv.z, live[2 % VEC] ? gv.z : 0.f, mv.z, vv.z, beta1, beta2, eps, bc1, bc2_sqrt, lr, decay, opacity_decay, decay_first);
      pv.w = adam_one(pv.w, live[3 % VEC] ? gv.w : 0.f, mv.w, vv.w, beta1, beta2, eps, bc1, bc2_sqrt, lr, decay, opacity_decay, decay_first);
      *reinterpret_cast<float4*>(m + e) = mv;
      *reinterpret_cast<float4*>(v + e) = vv;
      *reinterpret_cast<float4*>(p + e) = pv;
    } else {
      const float grad = any ? g[e] : 0.f;
      float mi = m[e], vi = v[e];
      const float pi = adam_one(p[e], grad, mi, vi, beta1, beta2, eps, bc1, bc2_sqrt, lr, decay, opacity_decay, decay_first);
      m[e] = mi;
      v[e] = vi;
      p[e] = pi;
    }
  }
  // the workgroup that finishes last advances the step counter (every workgroup has read it by then): no second launch.
  // The completion counter is the optimiser's own word next to its step (step_ptr[1], zero between launches).
  // Two levels (ABI 7): up to 8192 workgroups arriving at ONE address is ~10-18 ns each once they pile up -- 85 us of a
  // 105-us launch at 500k Gaussians, where the workgroups all finish together (tools/adam_time.py) -- so a workgroup counts
  // itself in one of B3GS_ADAM_SLOTS counters a cache line apart and only the last of every slot touches the shared one.
  if (bump) {
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t* w = reinterpret_cast<uint32_t*>(step_ptr);
      const uint32_t slot = blockIdx.x & (B3GS_ADAM_SLOTS - 1u);
      const uint32_t mine = (gridDim.x - slot + B3GS_ADAM_SLOTS - 1u) / B3GS_ADAM_SLOTS;   // workgroups that share this slot
      uint32_t* cnt = w + 2 + slot * B3GS_ADAM_SLOT_STRIDE;
      if (atomicAdd(cnt, 1u) == mine - 1u) {
        *cnt = 0u;
        const uint32_t nslots = gridDim.x < B3GS_ADAM_SLOTS ? gridDim.x : B3GS_ADAM_SLOTS;
        if (atomicAdd(w + 1, 1u) == nslots - 1u) {
          w[1] = 0u;
          *step_ptr += 1;
        }
      }
    }
  }
}

__global__ void bump_step(int32_t* step_ptr, const int32_t* skip_if_nonzero) {
  if (skip_if_nonzero && *skip_if_nonzero != 0) return;
  *step_ptr += 1;
}

}  // namespace

extern "C" int b3gs_adam_step(int32_t nseg, const B3gsAdamSegment* segs, int32_t* device_step, float beta1, float beta2,
                              float eps, float opacity_decay, int32_t opacity_segment, int32_t opacity_decay_first,
                              int32_t bump_step_after, const uint64_t* row_mask, const int32_t* skip_if_nonzero,
                              b3gs_stream_t stream) {
  if (nseg < 0 || nseg > 8 || (nseg > 0 && !segs) || !device_step)
    return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step", "0..8 segments and a device step counter are required");
  AdamSegs a;
  a.n = nseg;
  uint64_t tot = 0;
  bool vec4 = true;
  for (int k = 0; k < nseg; k++) {
    // an empty segment (e.g. features_rest at SH degree 0: [P,0,3]) may carry NULL pointers
    if (segs[k].count < 0 ||
        (segs[k].count > 0 && (!segs[k].param || !segs[k].grad || !segs[k].exp_avg || !segs[k].exp_avg_sq)))
      return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step", "negative count or NULL pointer in a non-empty segment");
    if (row_mask && (segs[k].row_len < 0 || segs[k].first_row < 0))
      return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step", "negative row_len / first_row in a masked step");
    a.s[k] = segs[k];
    a.start[k] = (uint32_t)tot;
    tot += (uint64_t)segs[k].count;
    const uintptr_t bits = (uintptr_t)segs[k].param | (uintptr_t)segs[k].grad | (uintptr_t)segs[k].exp_avg |
                           (uintptr_t)segs[k].exp_avg_sq;
    if (segs[k].count > 0 && (bits & 15u)) vec4 = false;
    if (segs[k].count > 0 && (segs[k].count & 3)) vec4 = false;
  }
  if (tot > 0xFFFFFFFFull) return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step", "more than 2^32 parameter floats in one call");
  a.start[nseg] = (uint32_t)tot;
  for (int k = nseg + 1; k < 9; k++) a.start[k] = (uint32_t)tot;
  if (tot == 0) {   // nothing to update (a rank whose shard is all padding); the step still counts
    if (bump_step_after) hipLaunchKernelGGL(bump_step, dim3(1), dim3(1), 0, (hipStream_t)stream, device_step, skip_if_nonzero);
    return b3gs_launch_status("b3gs_adam_step");
  }
  hipStream_t s = (hipStream_t)stream;
  const uint64_t items = vec4 ? tot / 4 : tot;
  const unsigned blocks = (unsigned)((items + 255) / 256 < 256u * 32u ? (items + 255) / 256 : 256u * 32u);
  const unsigned long long* mask = reinterpret_cast<const unsigned long long*>(row_mask);
  if (vec4)
    hipLaunchKernelGGL(adam_kernel<4>, dim3(blocks), dim3(256), 0, s, a, device_step, bump_step_after ? 1 : 0, beta1, beta2,
                       eps, opacity_decay, opacity_segment, opacity_decay_first, mask, skip_if_nonzero, 0);
  else
    hipLaunchKernelGGL(adam_kernel<1>, dim3(blocks), dim3(256), 0, s, a, device_step, bump_step_after ? 1 : 0, beta1, beta2,
                       eps, opacity_decay, opacity_segment, opacity_decay_first, mask, skip_if_nonzero, 0);
  return b3gs_launch_status("b3gs_adam_step");
}

// The same update with the step number given by the HOST (1-based: the value torch.optim.Adam's state["step"] holds AFTER
// its increment): the optimiser behind `gaussians.optimizer.step()` of an unchanged train.py:196-198 keeps torch's
// per-parameter state layout (step / exp_avg / exp_avg_sq), so the count lives where torch keeps it.
extern "C" int b3gs_adam_step_at(int32_t nseg, const B3gsAdamSegment* segs, int32_t step, float beta1, float beta2,
                                 float eps, b3gs_stream_t stream) {
  if (nseg < 0 || nseg > 8 || (nseg > 0 && !segs) || step < 1)
    return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step_at", "0..8 segments and a step number >= 1 are required");
  AdamSegs a;
  a.n = nseg;
  uint64_t tot = 0;
  bool vec4 = true;
  for (int k = 0; k < nseg; k++) {
    if (segs[k].count < 0 ||
        (segs[k].count > 0 && (!segs[k].param || !segs[k].grad || !segs[k].exp_avg || !segs[k].exp_avg_sq)))
      return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step_at", "negative count or NULL pointer in a non-empty segment");
    a.s[k] = segs[k];
    a.start[k] = (uint32_t)tot;
    tot += (uint64_t)segs[k].count;
    const uintptr_t bits = (uintptr_t)segs[k].param | (uintptr_t)segs[k].grad | (uintptr_t)segs[k].exp_avg |
                           (uintptr_t)segs[k].exp_avg_sq;
    if (segs[k].count > 0 && ((bits & 15u) || (segs[k].count & 3))) vec4 = false;
  }
  if (tot > 0xFFFFFFFFull) return b3gs_fail(B3GS_ERR_ARG, "b3gs_adam_step_at", "more than 2^32 parameter floats in one call");
  a.start[nseg] = (uint32_t)tot;
  for (int k = nseg + 1; k < 9; k++) a.start[k] = (uint32_t)tot;
  if (tot == 0) return B3GS_OK;
  hipStream_t s = (hipStream_t)stream;
  const uint64_t items = vec4 ? tot / 4 : tot;
  const unsigned blocks = (unsigned)((items + 255) / 256 < 256u * 32u ? (items + 255) / 256 : 256u * 32u);
  if (vec4)
    hipLaunchKernelGGL(adam_kernel<4>, dim3(blocks), dim3(256), 0, s, a, (int32_t*)nullptr, 0, beta1, beta2, eps, 0.0f, -1, 0,
                       (const unsigned long long*)nullptr, (const int32_t*)nullptr, step);
  else
    hipLaunchKernelGGL(adam_kernel<1>, dim3(blocks), dim3(256), 0, s, a, (int32_t*)nullptr, 0, beta1, beta2, eps, 0.0f, -1, 0,
                       (const unsigned long long*)nullptr, (const int32_t*)nullptr, step);
  return b3gs_launch_status("b3gs_adam_step_at");
}
