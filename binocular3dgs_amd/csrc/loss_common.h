// Pieces shared by the fused loss block (loss.hip) and the one-function-at-a-time loss entry points (lossfn.hip):
// the separable 11-tap SSIM window passes through LDS, the workgroup sum, the window itself.
#pragma once
#include "b3gs_internal.h"
#include <cmath>

namespace b3gs_loss {

struct Win { float w[11]; };

// window exactly as utils/loss_utils.py:23-26: double exp -> float tensor -> normalised in float
static inline Win make_window() {
  Win win;
  float g[11], tot = 0.f;
  for (int k = 0; k < 11; k++) { g[k] = (float)std::exp(-(double)((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5)); tot += g[k]; }
  for (int k = 0; k < 11; k++) win.w[k] = g[k] / tot;
  return win;
}

constexpr int LT = 16;          // output tile
constexpr int LR = 5;           // SSIM window radius
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ float block_sum_256(float v, float* tmp) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  const unsigned tid = threadIdx.y * blockDim.x + threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) tmp[tid >> 6] = v;
  __syncthreads();
  return tmp[0] + tmp[1] + tmp[2] + tmp[3];
}

// ---- SSIM: separable 11-tap window through LDS, 32x32 outputs per workgroup, 4 outputs per thread in both passes
// (a sliding window of 14 LDS values feeds 4 outputs: 3.4x fewer LDS reads than one output per thread)
constexpr int ST = 32;            // outputs per workgroup side
constexpr int SW = ST + 2 * LR;   // 42 inputs per side

// horizontal pass for NQ quantities: item = (row, group of 4 columns); in[q][row][col] -> out[q][row][col]
template <int NQ>
__device__ __forceinline__ void hpass4(const float (*in)[SW][SW + 1], float (*out)[SW][ST + 1], const Win& win, int tid) {
  for (int it = tid; it < SW * (ST / 4); it += 256) {
    const int r = it / (ST / 4), c0 = (it % (ST / 4)) * 4;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      float v[14];
#pragma unroll
      for (int j = 0; j < 14; j++) v[j] = in[q][r][c0 + j];
#pragma unroll
      for (int o = 0; o < 4; o++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) acc = fmaf(win.w[k], v[o + k], acc);
        out[q][r][c0 + o] = acc;
      }
    }
  }
}
// vertical pass: thread = (column tx, group of 4 rows): res[q][o] for rows 4*tg + o
template <int NQ>
__device__ __forceinline__ void vpass4(const float (*hq)[SW][ST + 1], const Win& win, int tx, int tg, float (&res)[NQ][4]) {
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    float v[14];
#pragma unroll
    for (int j = 0; j < 14; j++) v[j] = hq[q][4 * tg + j][tx];
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) acc = fmaf(win.w[k], v[o + k], acc);
      res[q][o] = acc;
    }
  }
}


}  // namespace b3gs_loss
