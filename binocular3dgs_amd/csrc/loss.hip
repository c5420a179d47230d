// Fused loss block of the binocular training step: value AND pixel gradients of
//   total = (1-l)*L1(image, gt) + l*(1 - SSIM(image, gt))                         train.py:145-147
//         + L1(warp(shifted, disparity)*mask, gt*mask) + 0.05*smooth(disparity*mask, gt)   train.py:131-136
//         + mean(|alpha| * alpha_weight)                                           train.py:139-143
// with  disparity = focal_x * (-trans_dist) / (depth + 1e-5)                       train.py:131,
// warp / mask = utils/graphics_utils.py:80-125 (linear interpolation between column c+floor(d) and the
// next one, zero where either tap leaves the image), smooth = utils/loss_utils.py:68-91 (central
// differences on the interior, weighted by exp(-0.33 |sum_c d gt|)), SSIM = utils/loss_utils.py:36-66
// (11x11 Gaussian window, sigma 1.5, zero padding, per channel).
// The reference runs ~40 PyTorch kernels per pair forward + their autograd (10.1 ms per iteration of 3 pairs
// at 800x600 on MI355X, four times the whole rasterizer); here: 4 launches per pair, gradients produced
// in the same pass as the value (the loss is the root of the graph: its upstream gradient is a scalar).
// Every image is H*W*{1,3} floats: L2-resident; the kernels are launch/latency bound, not HBM bound.
#include "loss_common.h"

namespace {
using namespace b3gs_loss;

// partial sums: thousands of workgroups adding to ONE address serialise in L2 (measured: 153 us for the SSIM
// statistics kernel, almost all of it the two atomics per workgroup); each of the 8 sums is spread over 64
// slots picked by workgroup index and folded by the finalize kernel
constexpr int SLOTS = 64;
__device__ __forceinline__ void add_sum(float* sums, int q, float v) {
  const unsigned b = blockIdx.y * gridDim.x + blockIdx.x + blockIdx.z * 7u;
  atomicAdd(&sums[q * SLOTS + (b & (SLOTS - 1))], v);
}

// one (input view, shifted view) pair; every kernel takes up to B3GS_MAX_FUSED_VIEWS pairs per launch (blockIdx.z)
struct PairArgs {
  int W, H;
  const float* image;
  const float* gt;
  const float* depth;
  const float* alpha;
  const float* shifted;       // null: no binocular term
  const float* alpha_weight;  // null: no alpha term
  float k_disp;               // focal_x * (-trans_dist)
  const float* t_dev;         // non-null: trans_dist lives on the device (B3gsLossIO::trans_dist_dev); k_disp is formed here
  float focal_x;
  float cS, cL1;              // -lambda_dssim*scale/(3HW), (1-lambda_dssim)*scale/(3HW)
  float c_l1m, c_smooth, c_alpha;   // scale/(3HW), lambda_smooth*scale/((H-2)(W-2)), scale/HW
  float lambda_dssim, lambda_smooth;
  float* sums;                // [8 * SLOTS]
  float* maps;                // [9 * HW]
  float* dL_dimage;
  float* dL_ddepth;
  float* dL_dalpha;
  float* dL_dshifted;         // zero on entry (atomics)
  float* parts;
};
struct LossBatch {
  int n;
  PairArgs p[B3GS_MAX_FUSED_VIEWS];
};

// sums[0] += sum |x-y|, sums[1] += sum ssim_map; maps: dS/dmu1, dS/dE[x^2], dS/dE[xy] per channel
// LDS: the x / y tile with its halo (14.4 KB) + the horizontal results of TWO quantities at a time (11 KB) = 25.5 KB, i.e.
// six workgroups per CU with the register cap of the launch bounds (76 VGPRs) -- the whole 800x600 grid (1425 workgroups)
// resident at once.  (All five quantities side by side were 42 KB and 114 VGPRs = three workgroups per CU = two rounds of a
// latency-bound workgroup: 28.5 us against 23.8 now; the gradient pass below 22.3 -> 17.4 us.)  The five window sums are
// formed group by group -- (mu1, mu2), (E[x^2], E[y^2]), E[xy] -- with the same taps in the same order: same bits.
__global__ void __launch_bounds__(256, 6) ssim_stats_kernel(LossBatch lb, Win win) {
  __shared__ float sin[2][SW][SW + 1];   // x, y with halo (the three products are formed in registers)
  __shared__ float hq[2][SW][ST + 1];
  __shared__ float red[4];
  const PairArgs& a = lb.p[blockIdx.z / 3];
  const int ch = blockIdx.z % 3;
  const int W = a.W, H = a.H;
  if ((int)blockIdx.x * ST >= W || (int)blockIdx.y * ST >= H) return;   // grid sized for the largest pair
  float* __restrict__ sums = a.sums;
  float* __restrict__ maps = a.maps;
  const size_t hw = (size_t)H * W;
  const float* __restrict__ x = a.image + ch * hw;
  const float* __restrict__ y = a.gt + ch * hw;
  const int tid = threadIdx.y * LT + threadIdx.x;
  const int r0 = blockIdx.y * ST - LR, c0 = blockIdx.x * ST - LR;
  for (int i = tid; i < SW * SW; i += 256) {
    const int r = i / SW, c = i - r * SW, gr = r0 + r, gc = c0 + c;
    const bool in = gr >= 0 && gr < H && gc >= 0 && gc < W;
    const float xv = in ? x[(size_t)gr * W + gc] : 0.f, yv = in ? y[(size_t)gr * W + gc] : 0.f;
    sin[0][r][c] = xv; sin[1][r][c] = yv;
  }
  __syncthreads();
  const int tx = tid & 31, tg = tid >> 5;
  float res[5][4];
#pragma unroll
  for (int grp = 0; grp < 3; grp++) {   // (mu1, mu2) | (E[x^2], E[y^2]) | (E[xy])
    for (int it = tid; it < SW * (ST / 4); it += 256) {   // horizontal pass: (row, group of 4 columns)
      const int r = it / (ST / 4), cc = (it % (ST / 4)) * 4;
      float xv[14], yv[14];
#pragma unroll
      for (int j = 0; j < 14; j++) { xv[j] = sin[0][r][cc + j]; yv[j] = sin[1][r][cc + j]; }
#pragma unroll
      for (int o = 0; o < 4; o++) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          const float wk = win.w[k], xx = xv[o + k], yy = yv[o + k];
          if (grp == 0) { a0 = fmaf(wk, xx, a0); a1 = fmaf(wk, yy, a1); }
          else if (grp == 1) { a0 = fmaf(wk, xx * xx, a0); a1 = fmaf(wk, yy * yy, a1); }
          else { a0 = fmaf(wk, xx * yy, a0); }
        }
        hq[0][r][cc + o] = a0;
        if (grp < 2) hq[1][r][cc + o] = a1;
      }
    }
    __syncthreads();
    if (grp < 2) {
      float two[2][4];
      vpass4<2>(hq, win, tx, tg, two);
#pragma unroll
      for (int o = 0; o < 4; o++) { res[2 * grp][o] = two[0][o]; res[2 * grp + 1][o] = two[1][o]; }
    } else {
      float one[1][4];
      vpass4<1>(hq, win, tx, tg, one);
#pragma unroll
      for (int o = 0; o < 4; o++) res[4][o] = one[0][o];
    }
    __syncthreads();   // (the next group overwrites hq)
  }
  float l1 = 0.f, ss = 0.f;
  const int gc = blockIdx.x * ST + tx;
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int lr = 4 * tg + o, gr = blockIdx.y * ST + lr;
    if (gr < H && gc < W) {
      const float mu1 = res[0][o], mu2 = res[1][o], e11 = res[2][o], e22 = res[3][o], e12 = res[4][o];
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
      const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
      const float inv = 1.0f / (B1 * B2);
      const float S = A1 * A2 * inv;
      ss += S;
      const size_t p = (size_t)gr * W + gc;
      // dS/dmu1 (mu1 also enters s1 and s12), dS/dE[x^2], dS/dE[xy]
      maps[(0 * 3 + ch) * hw + p] = 2.f * mu2 * (A2 - A1) * inv - S * (2.f * mu1 / B1 - 2.f * mu1 / B2);
      maps[(1 * 3 + ch) * hw + p] = -S / B2;
      maps[(2 * 3 + ch) * hw + p] = 2.f * A1 * inv;
      // the warp kernel below ADDS into the gradient of the shifted image (bilinear scatter): this pass, which visits
      // every (channel, pixel) exactly once, leaves it zero -- no 3 H W memset in front of the four launches
      if (a.dL_dshifted) a.dL_dshifted[ch * hw + p] = 0.f;
      l1 += fabsf(sin[0][lr + LR][tx + LR] - sin[1][lr + LR][tx + LR]);
    }
  }
  const float t0 = block_sum_256(l1, red);
  const float t1 = block_sum_256(ss, red);
  if (tid == 0) { add_sum(sums, 0, t0); add_sum(sums, 1, t1); }
}

// dL/dimage = cS * (w*dmu1 + 2x (w*de11) + y (w*de12)) + cL1 * sign(x - y)
// (the three maps one after the other through ONE tile + ONE row buffer: 12.7 KB of LDS instead of 38)
__global__ void __launch_bounds__(256) ssim_grad_kernel(LossBatch lb, Win win) {
  __shared__ float sm[1][SW][SW + 1];
  __shared__ float hq[1][SW][ST + 1];
  const PairArgs& a = lb.p[blockIdx.z / 3];
  const int ch = blockIdx.z % 3;
  const int W = a.W, H = a.H;
  if ((int)blockIdx.x * ST >= W || (int)blockIdx.y * ST >= H) return;
  const float* __restrict__ img = a.image;
  const float* __restrict__ gt = a.gt;
  const float* __restrict__ maps = a.maps;
  float* __restrict__ dL_dimage = a.dL_dimage;
  const float cS = a.cS, cL1 = a.cL1;
  const size_t hw = (size_t)H * W;
  const int tid = threadIdx.y * LT + threadIdx.x;
  const int r0 = blockIdx.y * ST - LR, c0 = blockIdx.x * ST - LR;
  const int tx = tid & 31, tg = tid >> 5;
  float res[3][4];
#pragma unroll
  for (int m = 0; m < 3; m++) {
    for (int i = tid; i < SW * SW; i += 256) {
      const int r = i / SW, c = i - r * SW, gr = r0 + r, gc = c0 + c;
      const bool in = gr >= 0 && gr < H && gc >= 0 && gc < W;
      sm[0][r][c] = in ? maps[(m * 3 + ch) * hw + (size_t)gr * W + gc] : 0.f;
    }
    __syncthreads();
    hpass4<1>(sm, hq, win, tid);
    __syncthreads();
    float one[1][4];
    vpass4<1>(hq, win, tx, tg, one);
#pragma unroll
    for (int o = 0; o < 4; o++) res[m][o] = one[0][o];
    // (the next map's tile load writes sm, which nobody reads any more; its hpass writes hq behind the next barrier pair)
    __syncthreads();
  }
  const int gc = blockIdx.x * ST + tx;
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int gr = blockIdx.y * ST + 4 * tg + o;
    if (gr < H && gc < W) {
      const size_t p = (size_t)gr * W + gc;
      const float xv = img[ch * hw + p], yv = gt[ch * hw + p];
      dL_dimage[ch * hw + p] = cS * (res[0][o] + 2.f * xv * res[1][o] + yv * res[2][o]) + cL1 * sgn(xv - yv);
    }
  }
}


struct Disp { float d, m; };
// disparity of pixel (r,c) and its warp-mask value ((x1-d)+(d-x0) where both taps are inside, else 0)
__device__ __forceinline__ Disp disparity_at(const PairArgs& a, int r, int c) {
  Disp o;
  o.d = 0.f; o.m = 0.f;
  if (r < 0 || r >= a.H || c < 0 || c >= a.W) return o;
  const float k_disp = a.t_dev ? a.focal_x * (-(*a.t_dev)) : a.k_disp;   // (the same fp32 product the host forms)
  const float d = k_disp / (a.depth[(size_t)r * a.W + c] + 1e-5f);
  o.d = d;
  if (!(fabsf(d) < 1.0e6f)) return o;
  const float x0 = floorf(d), x1 = x0 + 1.0f;
  const int c0 = c + (int)x0, c1 = c0 + 1;
  if (c0 < 0 || c0 >= a.W || c1 < 0 || c1 >= a.W) return o;
  o.m = (x1 - d) + (d - x0);
  return o;
}

__global__ void __launch_bounds__(256) binocular_kernel(LossBatch lb) {
  const PairArgs& a = lb.p[blockIdx.z];
  if ((int)blockIdx.x * LT >= a.W || (int)blockIdx.y * LT >= a.H) return;
  constexpr int HAL = 2, TW = LT + 2 * HAL;
  __shared__ float sD[TW][TW + 1];   // disparity * mask with a halo of 2
  __shared__ float sG[TW][TW + 1];   // sum over channels of gt, same halo (edge weights: one value per pixel)
  __shared__ float red4[4][4];
  const int W = a.W, H = a.H;
  const size_t hw = (size_t)H * W;
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * LT + tx;
  const int r = blockIdx.y * LT + ty, c = blockIdx.x * LT + tx;
  const bool in = r < H && c < W;
  const size_t p = in ? (size_t)r * W + c : 0;
  float s_alpha = 0.f, s_l1m = 0.f, s_sx = 0.f, s_sy = 0.f;
  if (in) {
    float ga = 0.f;
    if (a.alpha_weight) {
      const float al = a.alpha[p], w = a.alpha_weight[p];
      s_alpha = fabsf(al) * w;
      ga = sgn(al) * w * a.c_alpha;
    }
    a.dL_dalpha[p] = ga;
  }
  if (!a.shifted) {   // uniform
    if (in) a.dL_ddepth[p] = 0.f;
  } else {
    for (int i = tid; i < TW * TW; i += 256) {
      const int rr = i / TW, cc = i % TW;
      const Disp dd = disparity_at(a, (int)blockIdx.y * LT + rr - HAL, (int)blockIdx.x * LT + cc - HAL);
      sD[rr][cc] = dd.d * dd.m;
      const int gr = (int)blockIdx.y * LT + rr - HAL, gc = (int)blockIdx.x * LT + cc - HAL;
      float gsum = 0.f;
      if (gr >= 0 && gr < H && gc >= 0 && gc < W) {
        const size_t q = (size_t)gr * W + gc;
        gsum = (a.gt[q] + a.gt[hw + q]) + a.gt[2 * hw + q];
      }
      sG[rr][cc] = gsum;
    }
    __syncthreads();
    // the warp term: every lane of the wave runs the channel loop (the merge below talks to the neighbouring lanes)
    Disp me;
    me.d = 0.f; me.m = 0.f;
    if (in) me = disparity_at(a, r, c);
    const bool warp_ok = in && me.m != 0.f;   // both taps inside
    const float wx0 = warp_ok ? floorf(me.d) : 0.f;
    const int c0 = warp_ok ? c + (int)wx0 : -0x40000000, c1 = c0 + 1;
    const float w0 = (wx0 + 1.0f) - me.d, w1 = me.d - wx0;
    float dLdd = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      float v0 = 0.f, v1 = 0.f;
      bool has = false;
      if (warp_ok) {
        const float s0 = a.shifted[ch * hw + (size_t)r * W + c0], s1 = a.shifted[ch * hw + (size_t)r * W + c1];
        const float warped = w0 * s0 + w1 * s1;
        const float diff = warped * me.m - a.gt[ch * hw + p] * me.m;
        s_l1m += fabsf(diff);
        const float gW = sgn(diff) * me.m * a.c_l1m;   // dL/dwarped
        if (gW != 0.f) {
          has = true;
          v0 = w0 * gW;
          v1 = w1 * gW;
          dLdd += gW * (s1 - s0);
        }
      }
      // Neighbouring pixels of a row mostly share floor(disparity): my right tap is then my right neighbour's left tap.
      // Its contribution rides on my atomic and it skips its own (the scatter's atomics were 16 of the kernel's 36 us).
      const int n_c0 = __shfl_down(c0, 1, 64), l_c0 = __shfl_up(c0, 1, 64);
      const float n_v0 = __shfl_down(v0, 1, 64);
      const int n_has = __shfl_down((int)has, 1, 64), l_has = __shfl_up((int)has, 1, 64);
      const bool take = has && n_has && tx < LT - 1 && n_c0 == c1;          // (tx + 1 is the next lane of the same row)
      const bool given = has && l_has && tx > 0 && c0 == l_c0 + 1;
      if (take) v1 += n_v0;
#ifndef B3GS_LOSS_ABLATE_ATOMIC
      if (has && !given) atomicAdd(&a.dL_dshifted[ch * hw + (size_t)r * W + c0], v0);
      if (has) atomicAdd(&a.dL_dshifted[ch * hw + (size_t)r * W + c1], v1);
#endif
    }
    if (in) {
      // edge-aware smoothness of D' = d*m: |ex * dx(D')| + |ey * dy(D')| on the interior
      // g at location (rr,cc) along axis: value and d/dD' factor
      auto gx_at = [&](int rr, int cc, float& val) -> float {   // returns sign(v)*ex*c_smooth, val = |v|
        val = 0.f;
        if (rr < 1 || rr > H - 2 || cc < 1 || cc > W - 2) return 0.f;
        const int lr = rr - (int)blockIdx.y * LT + HAL, lc = cc - (int)blockIdx.x * LT + HAL;
        const float e = 0.5f * (sG[lr][lc + 1] - sG[lr][lc - 1]);   // sum_c dx(gt_c) = dx(sum_c gt_c)
        const float ex = __expf(fabsf(e) * -0.33f);
        const float v = ex * (0.5f * (sD[lr][lc + 1] - sD[lr][lc - 1]));
        val = fabsf(v);
        return sgn(v) * ex * a.c_smooth;
      };
      auto gy_at = [&](int rr, int cc, float& val) -> float {
        val = 0.f;
        if (rr < 1 || rr > H - 2 || cc < 1 || cc > W - 2) return 0.f;
        const int lr = rr - (int)blockIdx.y * LT + HAL, lc = cc - (int)blockIdx.x * LT + HAL;
        const float e = 0.5f * (sG[lr + 1][lc] - sG[lr - 1][lc]);
        const float ey = __expf(fabsf(e) * -0.33f);
        const float v = ey * (0.5f * (sD[lr + 1][lc] - sD[lr - 1][lc]));
        val = fabsf(v);
        return sgn(v) * ey * a.c_smooth;
      };
      float v, dDp = 0.f;
      (void)gx_at(r, c, v); s_sx = v;
      (void)gy_at(r, c, v); s_sy = v;
      // D'[r][c] is the +0.5 tap of location (r, c-1) and the -0.5 tap of (r, c+1); rows likewise; the
      // location's own row/column must be interior as well (dx is taken on rows 1..H-2, dy on columns 1..W-2)
      dDp += 0.5f * gx_at(r, c - 1, v);
      dDp -= 0.5f * gx_at(r, c + 1, v);
      dDp += 0.5f * gy_at(r - 1, c, v);
      dDp -= 0.5f * gy_at(r + 1, c, v);
      dLdd += dDp * me.m;
      a.dL_ddepth[p] = dLdd * (-me.d / (a.depth[p] + 1e-5f));
    }
  }
  // the four sums of the workgroup behind ONE pair of barriers
  float v4[4] = {s_alpha, s_l1m, s_sx, s_sy};
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v4[q] += __shfl_xor(v4[q], d, 64);
  __syncthreads();
  if ((tid & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) red4[q][tid >> 6] = v4[q];
  }
  __syncthreads();
  const float t0 = red4[0][0] + red4[0][1] + red4[0][2] + red4[0][3];
  const float t1 = red4[1][0] + red4[1][1] + red4[1][2] + red4[1][3];
  const float t2 = red4[2][0] + red4[2][1] + red4[2][2] + red4[2][3];
  const float t3 = red4[3][0] + red4[3][1] + red4[3][2] + red4[3][3];
  if (tid == 0) {
    if (t0 != 0.f) add_sum(a.sums, 5, t0);
    if (t1 != 0.f) add_sum(a.sums, 2, t1);
    if (t2 != 0.f) add_sum(a.sums, 3, t2);
    if (t3 != 0.f) add_sum(a.sums, 4, t3);
  }
}

__global__ void __launch_bounds__(64) loss_finalize_kernel(LossBatch lb) {
  const PairArgs& a = lb.p[blockIdx.x];
  float* __restrict__ slots = a.sums;
  const int W = a.W, H = a.H, has_shift = a.shifted != nullptr;
  const float lambda_dssim = a.lambda_dssim, lambda_smooth = a.lambda_smooth;
  float* __restrict__ parts = a.parts;
  float sums[8];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    float v = slots[q * SLOTS + threadIdx.x];
    slots[q * SLOTS + threadIdx.x] = 0.f;   // self-cleaning: zero for the next call (B3gsLossIO::workspace)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    sums[q] = v;
  }
  if (threadIdx.x != 0) return;
  const float hw = (float)H * (float)W;
  const float Ll1 = sums[0] / (3.f * hw), ssim = sums[1] / (3.f * hw);
  const float l1m = has_shift ? sums[2] / (3.f * hw) : 0.f;
  const float inner = (float)(H - 2) * (float)(W - 2);
  const float smooth = (has_shift && inner > 0.f) ? (sums[3] + sums[4]) / inner : 0.f;
  const float al = sums[5] / hw;
  parts[0] = ((1.f - lambda_dssim) * Ll1 + lambda_dssim * (1.f - ssim)) + (l1m + lambda_smooth * smooth) + al;
  parts[1] = Ll1; parts[2] = ssim; parts[3] = l1m; parts[4] = smooth; parts[5] = al; parts[6] = 0.f; parts[7] = 0.f;
}

}  // namespace

extern "C" size_t b3gs_loss_workspace_floats(int32_t W, int32_t H) {
  return 8 * SLOTS + (size_t)9 * (size_t)(W > 0 ? W : 0) * (size_t)(H > 0 ? H : 0);
}

extern "C" int b3gs_binocular_loss(const B3gsLossIO* io, b3gs_stream_t stream) {
  return b3gs_binocular_loss_batch(1, io, stream);
}

extern "C" int b3gs_binocular_loss_batch(int32_t npairs, const B3gsLossIO* ios, b3gs_stream_t stream) {
  if (npairs <= 0 || npairs > B3GS_MAX_FUSED_VIEWS || !ios)
    return b3gs_fail(B3GS_ERR_ARG, "b3gs_binocular_loss", "pair count must be 1..8 and ios non-NULL");
  hipStream_t s = (hipStream_t)stream;
  const Win win = make_window();
  LossBatch lb;
  lb.n = npairs;
  int gx = 0, gy = 0;
  for (int k = 0; k < npairs; k++) {
    const B3gsLossIO* io = ios + k;
    if (io->W <= 0 || io->H <= 0 || !io->image || !io->depth || !io->alpha || !io->gt_image || !io->dL_dimage ||
        !io->dL_ddepth || !io->dL_dalpha || !io->parts || !io->workspace || (io->shifted_image && !io->dL_dshifted))
      return b3gs_fail(B3GS_ERR_ARG, "b3gs_binocular_loss", "bad size or NULL image / gradient / workspace pointer");
    const int W = io->W, H = io->H;
    const size_t hw = (size_t)W * H;
    const float scale = io->grad_scale, inner = (float)(H - 2) * (float)(W - 2);
    PairArgs& a = lb.p[k];
    a.W = W; a.H = H;
    a.image = io->image; a.gt = io->gt_image; a.depth = io->depth; a.alpha = io->alpha;
    a.shifted = io->shifted_image; a.alpha_weight = io->alpha_weight;
    a.k_disp = io->focal_x * (-io->trans_dist);
    a.t_dev = io->trans_dist_dev;
    a.focal_x = io->focal_x;
    a.cS = -io->lambda_dssim * scale / (3.f * (float)hw);
    a.cL1 = (1.f - io->lambda_dssim) * scale / (3.f * (float)hw);
    a.c_l1m = scale / (3.f * (float)hw);
    a.c_smooth = inner > 0.f ? io->lambda_smooth * scale / inner : 0.f;
    a.c_alpha = scale / (float)hw;
    a.lambda_dssim = io->lambda_dssim; a.lambda_smooth = io->lambda_smooth;
    a.sums = io->workspace;
    a.maps = io->workspace + 8 * SLOTS;
    a.dL_dimage = io->dL_dimage; a.dL_ddepth = io->dL_ddepth; a.dL_dalpha = io->dL_dalpha; a.dL_dshifted = io->dL_dshifted;
    a.parts = io->parts;
    // (no memsets: the partial-sum slots are left zero by the previous call's last kernel -- the caller zeroes a NEW
    // workspace once --, the gradient of the shifted image is zeroed by the SSIM statistics pass)
    a.dL_dshifted = io->shifted_image ? io->dL_dshifted : nullptr;
    gx = (W + LT - 1) / LT > gx ? (W + LT - 1) / LT : gx;
    gy = (H + LT - 1) / LT > gy ? (H + LT - 1) / LT : gy;
  }
  const dim3 blk(LT, LT);
  const dim3 sgrid((gx * LT + ST - 1) / ST, (gy * LT + ST - 1) / ST, 3 * npairs);   // SSIM: 32x32 outputs per workgroup
  hipLaunchKernelGGL(ssim_stats_kernel, sgrid, blk, 0, s, lb, win);
  hipLaunchKernelGGL(ssim_grad_kernel, sgrid, blk, 0, s, lb, win);
  hipLaunchKernelGGL(binocular_kernel, dim3(gx, gy, npairs), blk, 0, s, lb);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(npairs), dim3(SLOTS), 0, s, lb);
  return b3gs_launch_status("b3gs_binocular_loss");
}
