// Per-tile alpha blending, forward and backward -- the two hot kernels of the rasterizer
// (the `render` stages of the extension the reference calls at
// gaussian_renderer/__init__.py:85-93 and differentiates at train.py:149; outputs: colour,
// un-normalised depth sum(z a T) and alpha sum(a T), consumed at train.py:101-106,131,141-143).
//
// CDNA4 design (wave64, not a 32-lane translation):
//  * one 256-thread workgroup per 16x16 tile = 4 waves, each wave owns one 8x8 pixel QUADRANT
//    (lane -> (lane&7, lane>>3)), so a wave-uniform test can discard a Gaussian for 64 pixels at once
//  * the tile's depth-sorted list is staged 256 Gaussians at a time through LDS: every lane
//    gathers one 64-byte record (one cache line per Gaussian), and while staging each wave
//    ballots "does this Gaussian's alpha>=1/255 footprint reach quadrant q?" for its 64 records.
//    The 4x4 64-bit masks go to LDS; the consumer wave walks only the set bits of its own
//    quadrant's masks with scalar bit ops (s_ff1/s_flbit) -- ballot compaction without moving data.
//    The footprint test is conservative, so the per-pixel arithmetic, skip rules and
//    n_contrib are exactly those of the sequential algorithm (oracle/tile_ref.c).
//  * record fields are read from LDS with broadcast ds_read_b128 (all lanes, one address)
//  * backward: lanes hold per-pixel partials; a 6-step DPP reduction folds the 64 lanes, the
//    four quadrant waves merge in LDS (ds_add_f32) and each (tile, Gaussian) issues ONE set of
//    global atomics instead of one per pixel
//  * workgroup -> tile map keeps raster-adjacent tiles (which share Gaussians) on one XCD's L2
#include "b3gs_internal.h"

namespace {

typedef unsigned long long u64;
constexpr int CHUNK = 256;

__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ u64 uniform_u64(u64 v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((u64)hi << 32) | lo;
}

// XCD-aware tile assignment: workgroup b is observed to run on XCD b % 8; give each XCD a
// contiguous band of tiles.  Placement only affects L2 hit rate, never results.
__device__ __forceinline__ int tile_of_block(int bid, int ntiles) {
  const int per = (ntiles + 7) >> 3;
  return (bid & 7) * per + (bid >> 3);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(t);
}
// sum over the 64 lanes; the total is valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);  // row_mirror        -> every lane holds its row-of-16 sum
  v = dpp_add<0x142, 0xA>(v);  // row_bcast15 into rows 1,3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast31 into rows 2,3 -> lane 63 = total
  return v;
}

struct TileShared {
  float4 A[CHUNK];  // x, y, cxx, cxy
  float4 B[CHUNK];  // cyy, opacity, r, g
  float4 C[CHUNK];  // b, depth, -, -
  u64 mask[4][4];   // [quadrant][producer wave]
};

// Stage one chunk: lane `tid` fetches list entry (first + tid); returns the Gaussian id (or -1).
__device__ __forceinline__ int stage_chunk(TileShared& sh, const uint32_t* __restrict__ point_list,
                                           const float4* __restrict__ rec, uint32_t first, uint32_t last,
                                           float tile_px, float tile_py) {
  const unsigned tid = threadIdx.x;
  const uint32_t idx = first + tid;
  int id = -1;
  bool hit[4] = {false, false, false, false};
  if (idx < last) {
    id = (int)point_list[idx];
    const float4* r = rec + 4 * (size_t)id;
    const float4 r0 = r[0], r1 = r[1], r2 = r[2];
    sh.A[tid] = r0;
    sh.B[tid] = r1;
    sh.C[tid] = make_float4(r2.x, r2.y, 0.f, 0.f);
    // footprint [x-ex, x+ex] x [y-ey, y+ey] against the four 8x8 quadrants (pixel centres
    // tile_p + {0..7} and tile_p + {8..15})
    const float lx = r0.x - r2.z - tile_px, hx = r0.x + r2.z - tile_px;
    const float ly = r0.y - r2.w - tile_py, hy = r0.y + r2.w - tile_py;
    const bool xl = (lx <= 7.0f) && (hx >= 0.0f), xr = (lx <= 15.0f) && (hx >= 8.0f);
    const bool yt = (ly <= 7.0f) && (hy >= 0.0f), yb = (ly <= 15.0f) && (hy >= 8.0f);
    hit[0] = xl && yt;
    hit[1] = xr && yt;
    hit[2] = xl && yb;
    hit[3] = xr && yb;
  }
  const unsigned w = tid >> 6;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const u64 m = __ballot(hit[q]);
    if ((tid & 63) == 0) sh.mask[q][w] = m;
  }
  return id;
}

__device__ __forceinline__ float blend_power(const float4& A, float cyy, float dx, float dy) {
  const float q = __builtin_fmaf(cyy * dy, dy, (A.z * dx) * dx);
  return __builtin_fmaf(-A.w * dx, dy, -0.5f * q);
}

__global__ void __launch_bounds__(256)
    render_fwd_kernel(int W, int H, int grid_x, int ntiles, const uint2* __restrict__ ranges,
                      const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
                      const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                      float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha) {
  __shared__ TileShared sh;
  const int tile = tile_of_block(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int px = tile_x * B3GS_TILE + (int)((w & 1) * 8 + (lane & 7));
  const int py = tile_y * B3GS_TILE + (int)((w >> 1) * 8 + (lane >> 3));
  const bool inside = px < W && py < H;
  const float fpx = (float)px, fpy = (float)py;
  const uint2 range = ranges[tile];
  const int nchunks = (int)((range.y - range.x + CHUNK - 1) / CHUNK);

  bool done = !inside;
  float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Dp = 0.f, Ac = 0.f;
  uint32_t last_contributor = 0;

  for (int c = 0; c < nchunks; c++) {
    if (__syncthreads_and(done)) break;
    stage_chunk(sh, point_list, rec, range.x + c * CHUNK, range.y, (float)(tile_x * B3GS_TILE), (float)(tile_y * B3GS_TILE));
    __syncthreads();
    if (__ballot(!done) == 0) continue;  // this quadrant is finished; keep pace with the barriers
#pragma unroll 1
    for (int pw = 0; pw < 4; pw++) {
      u64 m = uniform_u64(sh.mask[w][pw]);
      while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        const int gidx = pw * 64 + j;
        const float4 A = sh.A[gidx];
        const float4 B = sh.B[gidx];
        const float4 Cc = sh.C[gidx];
        const float dx = A.x - fpx, dy = A.y - fpy;
        const float power = blend_power(A, B.x, dx, dy);
        const float alpha = fminf(B3GS_ALPHA_MAX, B.y * __expf(power));
        const float test_T = T * (1.0f - alpha);
        const bool live = !done && !(power > 0.0f) && !(alpha < B3GS_ALPHA_MIN);
        if (live && test_T < B3GS_T_EPS) done = true;
        if (live && !done) {
          const float wgt = alpha * T;
          Cr = __builtin_fmaf(B.z, wgt, Cr);
          Cg = __builtin_fmaf(B.w, wgt, Cg);
          Cb = __builtin_fmaf(Cc.x, wgt, Cb);
          Dp = __builtin_fmaf(Cc.y, wgt, Dp);
          Ac += wgt;
          T = test_T;
          last_contributor = (uint32_t)(c * CHUNK + gidx + 1);
        }
        if (__ballot(!done) == 0) break;
      }
      if (__ballot(!done) == 0) break;
    }
  }
  if (inside) {
    const size_t pix = (size_t)py * W + px, hw = (size_t)H * W;
    final_T[pix] = T;
    n_contrib[pix] = last_contributor;
    out_color[pix] = __builtin_fmaf(T, bg[0], Cr);
    out_color[hw + pix] = __builtin_fmaf(T, bg[1], Cg);
    out_color[2 * hw + pix] = __builtin_fmaf(T, bg[2], Cb);
    out_depth[pix] = Dp;
    out_alpha[pix] = Ac;
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
struct TileSharedBwd {
  TileShared f;
  int id[CHUNK];
  float acc[10][CHUNK];   // mean2D.x, mean2D.y, conic xx, xy(half), yy, opacity, r, g, b, depth
  uint32_t touched[CHUNK];
};

__global__ void __launch_bounds__(256)
    render_bwd_kernel(int W, int H, int grid_x, int ntiles, const uint2* __restrict__ ranges,
                      const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
                      const float* __restrict__ bg, const float* __restrict__ final_T,
                      const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
                      const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha_img,
                      float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
                      float* __restrict__ dL_dcov3D) {
  __shared__ TileSharedBwd sh;
  const int tile = tile_of_block(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int px = tile_x * B3GS_TILE + (int)((w & 1) * 8 + (lane & 7));
  const int py = tile_y * B3GS_TILE + (int)((w >> 1) * 8 + (lane >> 3));
  const bool inside = px < W && py < H;
  const float fpx = (float)px, fpy = (float)py;
  const uint2 range = ranges[tile];
  const float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;

  uint32_t last = 0;
  float T_final = 0.f, dCr = 0.f, dCg = 0.f, dCb = 0.f, dD = 0.f, dA = 0.f;
  if (inside) {
    const size_t pix = (size_t)py * W + px, hw = (size_t)H * W;
    last = n_contrib[pix];
    T_final = final_T[pix];
    dCr = dL_dcolor[pix];
    dCg = dL_dcolor[hw + pix];
    dCb = dL_dcolor[2 * hw + pix];
    if (dL_ddepth) dD = dL_ddepth[pix];
    if (dL_dalpha_img) dA = dL_dalpha_img[pix];
  }
  const float bg_dot = (bg[0] * dCr + bg[1] * dCg) + bg[2] * dCb;
  float T = T_final;
  float Br = 0.f, Bg = 0.f, Bb = 0.f, Bd = 0.f, Ba = 0.f;  // composite "behind" the current Gaussian

  // deepest list position any pixel of the tile used
  __shared__ uint32_t s_max_last[4];
  {
    uint32_t m = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0) s_max_last[w] = m;
  }
#pragma unroll
  for (int k = 0; k < 10; k++) sh.acc[k][tid] = 0.f;
  sh.touched[tid] = 0;
  __syncthreads();
  const uint32_t max_last = max(max(s_max_last[0], s_max_last[1]), max(s_max_last[2], s_max_last[3]));
  if (max_last == 0) return;
  const uint32_t wave_last = __builtin_amdgcn_readfirstlane(s_max_last[w]);

  for (int c = (int)((max_last - 1) / CHUNK); c >= 0; c--) {
    const int id = stage_chunk(sh.f, point_list, rec, range.x + c * CHUNK, range.y, (float)(tile_x * B3GS_TILE),
                               (float)(tile_y * B3GS_TILE));
    sh.id[tid] = id;
    __syncthreads();
    if ((uint32_t)(c * CHUNK) < wave_last) {
#pragma unroll 1
      for (int pw = 3; pw >= 0; pw--) {
        u64 m = uniform_u64(sh.f.mask[w][pw]);
        while (m) {
          const int j = 63 - __builtin_clzll(m);
          m &= ~(1ull << j);
          const int gidx = pw * 64 + j;
          const uint32_t pos = (uint32_t)(c * CHUNK + gidx);
          if (pos >= wave_last) continue;
          const float4 A = sh.f.A[gidx];
          const float4 B = sh.f.B[gidx];
          const float4 Cc = sh.f.C[gidx];
          const float dx = A.x - fpx, dy = A.y - fpy;
          const float power = blend_power(A, B.x, dx, dy);
          const float G = __expf(power);
          const float alpha = fminf(B3GS_ALPHA_MAX, B.y * G);
          const bool live = (pos < last) && !(power > 0.0f) && !(alpha < B3GS_ALPHA_MIN);
          if (__ballot(live) == 0) continue;
          float p[10];
#pragma unroll
          for (int k = 0; k < 10; k++) p[k] = 0.f;
          if (live) {
            const float one_m_a = 1.0f - alpha;
            T = T / one_m_a;
            const float wgt = alpha * T;
            float dr = B.z - Br, dg = B.w - Bg, db = Cc.x - Bb, dd = Cc.y - Bd, da = 1.0f - Ba;
            float dL_da = dr * dCr;
            dL_da = __builtin_fmaf(dg, dCg, dL_da);
            dL_da = __builtin_fmaf(db, dCb, dL_da);
            dL_da = __builtin_fmaf(dd, dD, dL_da);
            dL_da = __builtin_fmaf(da, dA, dL_da);
            Br = __builtin_fmaf(alpha, dr, Br);
            Bg = __builtin_fmaf(alpha, dg, Bg);
            Bb = __builtin_fmaf(alpha, db, Bb);
            Bd = __builtin_fmaf(alpha, dd, Bd);
            Ba = __builtin_fmaf(alpha, da, Ba);
            dL_da = dL_da * T;
            dL_da = __builtin_fmaf(-T_final / one_m_a, bg_dot, dL_da);
            const float dL_dG = B.y * dL_da;
            const float gdx = G * dx, gdy = G * dy;
            p[0] = dL_dG * (-gdx * A.z - gdy * A.w) * half_w;
            p[1] = dL_dG * (-gdy * B.x - gdx * A.w) * half_h;
            p[2] = -0.5f * gdx * dx * dL_dG;
            p[3] = -0.5f * gdx * dy * dL_dG;
            p[4] = -0.5f * gdy * dy * dL_dG;
            p[5] = G * dL_da;
            p[6] = wgt * dCr;
            p[7] = wgt * dCg;
            p[8] = wgt * dCb;
            p[9] = wgt * dD;
          }
#pragma unroll
          for (int k = 0; k < 10; k++) p[k] = wave_sum_to_lane63(p[k]);
          if (lane == 63) {
#pragma unroll
            for (int k = 0; k < 10; k++) atomicAdd(&sh.acc[k][gidx], p[k]);
            sh.touched[gidx] = 1;
          }
        }
      }
    }
    __syncthreads();
    // one set of global atomics per (tile, Gaussian)
    if (sh.touched[tid]) {
      const size_t g = (size_t)sh.id[tid];
      unsafeAtomicAdd(&dL_dmeans2D[3 * g + 0], sh.acc[0][tid]);
      unsafeAtomicAdd(&dL_dmeans2D[3 * g + 1], sh.acc[1][tid]);
      unsafeAtomicAdd(&dL_dcov3D[6 * g + 0], sh.acc[2][tid]);
      unsafeAtomicAdd(&dL_dcov3D[6 * g + 1], sh.acc[3][tid]);
      unsafeAtomicAdd(&dL_dcov3D[6 * g + 2], sh.acc[4][tid]);
      unsafeAtomicAdd(&dL_dopacity[g], sh.acc[5][tid]);
      unsafeAtomicAdd(&dL_dcolors[3 * g + 0], sh.acc[6][tid]);
      unsafeAtomicAdd(&dL_dcolors[3 * g + 1], sh.acc[7][tid]);
      unsafeAtomicAdd(&dL_dcolors[3 * g + 2], sh.acc[8][tid]);
      unsafeAtomicAdd(&dL_dcov3D[6 * g + 3], sh.acc[9][tid]);
#pragma unroll
      for (int k = 0; k < 10; k++) sh.acc[k][tid] = 0.f;
      sh.touched[tid] = 0;
    }
    __syncthreads();
  }
}

}  // namespace

void b3gs_launch_render_forward(const B3gsScene& sc, const GeomView& g, const BinView& b, const ImgView& im,
                                float* out_color, float* out_depth, float* out_alpha, hipStream_t s) {
  const int gx = (sc.W + B3GS_TILE - 1) / B3GS_TILE, gy = (sc.H + B3GS_TILE - 1) / B3GS_TILE;
  const int ntiles = gx * gy;
  if (ntiles <= 0) return;
  const int nblocks = ((ntiles + 7) / 8) * 8;
  hipLaunchKernelGGL(render_fwd_kernel, dim3(nblocks), dim3(256), 0, s, sc.W, sc.H, gx, ntiles, im.ranges, b.val[0],
                     g.rec, sc.background, im.final_T, im.n_contrib, out_color, out_depth, out_alpha);
}

void b3gs_launch_render_backward(const B3gsScene& sc, const GeomView& g, const BinView& b, const ImgView& im,
                                 const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                                 float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dcov3D,
                                 hipStream_t s) {
  const int gx = (sc.W + B3GS_TILE - 1) / B3GS_TILE, gy = (sc.H + B3GS_TILE - 1) / B3GS_TILE;
  const int ntiles = gx * gy;
  if (ntiles <= 0) return;
  const int nblocks = ((ntiles + 7) / 8) * 8;
  hipLaunchKernelGGL(render_bwd_kernel, dim3(nblocks), dim3(256), 0, s, sc.W, sc.H, gx, ntiles, im.ranges, b.val[0],
                     g.rec, sc.background, im.final_T, im.n_contrib, dL_dcolor, dL_ddepth, dL_dalpha, dL_dmeans2D,
                     dL_dcolors, dL_dopacity, dL_dcov3D);
}
