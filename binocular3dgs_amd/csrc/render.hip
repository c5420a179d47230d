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
//    An exact test (minimum of the quadratic form over the quadrant's pixel rectangle vs ln(255 opacity)) follows
//    the bounding-box test: every candidate it removes saves a 64-lane evaluation in each direction.
//  * record fields are read from LDS with broadcast ds_read_b128 (all lanes, one address)
//  * backward: lanes hold per-pixel partials of 10 gradient components; they are transposed through a per-wave
//    LDS scratch (ds_write_addtid_b32 rows, 4 x ds_read_b128 per lane, 15 adds + 2 quad DPP) and each (quadrant,
//    Gaussian) issues ONE 10-lane global_atomic_add_f32 onto one 40-byte scratch row (details at the kernel)
//  * one launch covers all views of a training iteration (BlendBatch): ~11k tiles pack the 256 CUs, one view's
//    1900 tiles would fill them once and pay their own tail
//  * workgroup -> tile map keeps raster-adjacent tiles (which share Gaussians) on one XCD's L2
#include "b3gs_internal.h"
#include <cstdlib>
#include <cstring>

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// clear bit j of a wave-uniform 64-bit mask: one scalar instruction (the compiler's  m & (m - 1)  is three, and the
// blend loops are co-limited by the scalar unit)
__device__ __forceinline__ void clear_bit(u64& m, int j) {
#ifdef B3GS_NO_BITSET0
  m &= ~(1ull << j);
#else
  asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(j));
#endif
}
__device__ __forceinline__ u64 uniform_u64(u64 v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((u64)hi << 32) | lo;
}

constexpr uint32_t B3GS_ORDER_MAGIC = 0xB365A0D1u;
// XCD-aware tile assignment: workgroup b is observed to run on XCD b % 8; give each XCD a
// contiguous band of tiles.  Placement only affects L2 hit rate, never results.
__device__ __forceinline__ int tile_of_block(int bid, int ntiles) {
  const int per = (ntiles + 7) >> 3;
  return (bid & 7) * per + (bid >> 3);
}

// Pick the view a workgroup belongs to.  The table travels in the kernel arguments; a chain of
// wave-uniform selects (scalar moves, once per workgroup) avoids dynamic indexing of the argument
// block, which the compiler would otherwise copy to scratch memory.
__device__ __forceinline__ BlendView select_view(const BlendBatch& batch, int bid) {
  BlendView v = batch.v[0];
#pragma unroll
  for (int k = 1; k < B3GS_MAX_FUSED_VIEWS; k++)
    if (k < batch.n && bid >= batch.v[k].block_base) v = batch.v[k];
  return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(t);
}
// CHUNK = Gaussians staged per round (a multiple of 64, at most the 256 threads of the workgroup)
template <int CHUNK>
struct TileShared {
  float4 A[CHUNK];  // x, y, cxx, cxy
  float4 B[CHUNK];  // cyy, opacity, r, g
  float4 C[CHUNK];  // b, depth, Gaussian index (bits), -
  u64 mask[4][CHUNK / 64];   // [quadrant][producer wave]
};

// A tile's list is the concatenation of up to two segments (two-round binning, b3gs_internal.h BinJob::K1): positions
// [0, len1) live in point_list at r1.x, positions [len1, len1 + len2) in point_list2 at r2.x.
struct TileList {
  const uint32_t* list1;
  const uint32_t* list2;
  uint32_t first1, first2, len1, total;
};
__device__ __forceinline__ TileList tile_list(const BlendView& bv, int tile, bool with_segment2) {
  uint2 r1 = bv.ranges[tile];
  if (r1.y <= r1.x) r1 = make_uint2(0u, 0u);   // empty tiles hold (0xFFFFFFFF, 0)
  uint2 r2 = with_segment2 ? bv.ranges2[tile] : make_uint2(0u, 0u);
  if (r2.y <= r2.x) r2 = make_uint2(0u, 0u);
  TileList t;
  t.list1 = bv.point_list;
  t.list2 = bv.point_list2;
  t.first1 = r1.x;
  t.first2 = r2.x;
  t.len1 = r1.y - r1.x;
  t.total = t.len1 + (r2.y - r2.x);
  return t;
}

// Stage one chunk: lane `tid` fetches list position (q0 + tid); returns the Gaussian id (or -1).
template <int CHUNK>
__device__ __forceinline__ int stage_chunk(TileShared<CHUNK>& sh, const TileList& tl, uint32_t idx_mask,
                                           const float4* __restrict__ rec, uint32_t q0, float tile_px, float tile_py) {
  const unsigned tid = threadIdx.x;
  const uint32_t q = q0 + tid;
  int id = -1;
  // bit q of `hits`: the staged Gaussian can reach quadrant q.  (A bit mask and a rolled loop over the quadrants: the four
  // exact tests unrolled side by side were what set the register count of BOTH blend kernels -- 70 instead of 44 VGPRs for
  // the forward -- and staging runs once per 256 list entries.)
  uint32_t hits = 0u;
  if (CHUNK < 256 && tid >= (unsigned)CHUNK) return id;  // whole waves: wave-uniform exit
  if (q < tl.total) {
    const uint32_t word = q < tl.len1 ? tl.list1[tl.first1 + q] : tl.list2[tl.first2 + (q - tl.len1)];
    id = (int)(word & idx_mask);
    const float4* r = rec + 4 * (size_t)id;
    const float4 r0 = r[0], r1 = r[1], r2 = r[2];
    sh.A[tid] = r0;
    sh.B[tid] = r1;
    sh.C[tid] = make_float4(r2.x, r2.y, __int_as_float(id), 0.f);
    // footprint [x-ex, x+ex] x [y-ey, y+ey] against the four 8x8 quadrants (pixel centres
    // tile_p + {0..7} and tile_p + {8..15})
    const float lx = r0.x - r2.z - tile_px, hx = r0.x + r2.z - tile_px;
    const float ly = r0.y - r2.w - tile_py, hy = r0.y + r2.w - tile_py;
    const bool xl = (lx <= 7.0f) && (hx >= 0.0f), xr = (lx <= 15.0f) && (hx >= 8.0f);
    const bool yt = (ly <= 7.0f) && (hy >= 0.0f), yb = (ly <= 15.0f) && (hy >= 8.0f);
    hits = (uint32_t)(xl && yt) | ((uint32_t)(xr && yt) << 1) | ((uint32_t)(xl && yb) << 2) | ((uint32_t)(xr && yb) << 3);
    // Exact test for the quadrants the bounding box reaches: the smallest value of the quadratic form
    // over the quadrant's pixel rectangle against tau = ln(255 opacity) (alpha >= 1/255 <=> form <= tau).
    // One thread does this once per staged Gaussian; every candidate it removes saves a 64-lane evaluation
    // in each direction.  Conservative: continuous rectangle >= pixel centres, plus a rounding margin.
    if (hits) {
      const float cxx = r0.z, cxy = r0.w, cyy = r1.x;
      const float tau = __logf(255.0f * r1.y) * 1.0005f + 2e-3f;
      const float icx = __builtin_amdgcn_rcpf(cxx), icy = __builtin_amdgcn_rcpf(cyy);
      const float ox = tile_px - r0.x, oy = tile_py - r0.y;   // rectangle corner relative to the mean
#pragma unroll 1
      for (int qd = 0; qd < 4; qd++) {
        if (!((hits >> qd) & 1u)) continue;
        const float ax = ox + (float)((qd & 1) * 8), bx = ax + 7.0f;
        const float ay = oy + (float)((qd >> 1) * 8), by = ay + 7.0f;
        const bool inside = (ax <= 0.0f) && (bx >= 0.0f) && (ay <= 0.0f) && (by >= 0.0f);
        float fmin = 3.0e38f;
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float dx = e ? bx : ax;                                   // vertical edge
          const float dy = fminf(fmaxf(-cxy * dx * icy, ay), by);
          fmin = fminf(fmin, 0.5f * (cxx * dx * dx + cyy * dy * dy) + cxy * dx * dy);
          const float ey = e ? by : ay;                                   // horizontal edge
          const float ex = fminf(fmaxf(-cxy * ey * icx, ax), bx);
          fmin = fminf(fmin, 0.5f * (cxx * ex * ex + cyy * ey * ey) + cxy * ex * ey);
        }
        // (a NaN anywhere fails `!(fmin > tau)`'s complement only by keeping the candidate)
        if (!(inside || !(fmin > tau))) hits &= ~(1u << qd);
      }
    }
  }
  const unsigned w = tid >> 6;
#pragma unroll
  for (int qd = 0; qd < 4; qd++) {
    const u64 m = __ballot((hits >> qd) & 1u);
    if ((tid & 63) == 0) sh.mask[qd][w] = m;
  }
  return id;
}

__device__ __forceinline__ float blend_power(const float4& A, float cyy, float dx, float dy) {
  const float q = __builtin_fmaf(cyy * dy, dy, (A.z * dx) * dx);
  return __builtin_fmaf(-A.w * dx, dy, -0.5f * q);
}

// A blend launch covers one or several views: workgroups [block_base, next block_base) belong to view v.
// Batching the views of an iteration into one launch lets the dispatcher pack ~11k tiles over the
// machine (one view's 1900 tiles fill it exactly once, so every launch paid its own tail:
// measured 126 us for one view, 446 us for six in one launch).
// `bid` = default block index (tile of a view, see select_view / tile_of_block); `slot` = where a trace record goes
// sensitivity probes (tools only): extra instructions of one kind per candidate, to see which unit paces the loop
#if defined(B3GS_FWD_SENS_SALU)
#define B3GS_FWD_SENS asm volatile("s_add_u32 s2, s2, 1\n s_add_u32 s2, s2, 1\n s_add_u32 s2, s2, 1\n s_add_u32 s2, s2, 1" ::: "s2", "scc")
#elif defined(B3GS_FWD_SENS_VALU)
#define B3GS_FWD_SENS asm volatile("v_add_u32 v60, v60, 1\n v_add_u32 v61, v61, 1\n v_add_u32 v62, v62, 1\n v_add_u32 v63, v63, 1" ::: "v60", "v61", "v62", "v63")
#else
#define B3GS_FWD_SENS do { } while (0)
#endif
#ifndef B3GS_FWD_CHECK_EVERY
#define B3GS_FWD_CHECK_EVERY 4   /* candidates between two "is the quadrant finished" checks of the forward's hot loop */
#endif
template <int CHUNK, bool TRACE>
__device__ __forceinline__ void render_fwd_tile(const BlendView bv, int bid, unsigned slot, TileShared<CHUNK>& sh,
                                                uint32_t (&s_work)[4], unsigned long long* __restrict__ trace) {
  const unsigned long long t_start = TRACE ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long r_start = TRACE ? __builtin_amdgcn_s_memrealtime() : 0ull;
  unsigned n_iter = 0, n_chunks = 0;
  const int W = bv.W, H = bv.H, grid_x = bv.grid_x, ntiles = bv.ntiles;
  const float4* __restrict__ rec = bv.rec;
  const float* __restrict__ bg = bv.bg;
  float* __restrict__ final_T = bv.final_T;
  uint32_t* __restrict__ n_contrib = bv.n_contrib;
  float* __restrict__ out_color = bv.out_color;
  float* __restrict__ out_depth = bv.out_depth;
  float* __restrict__ out_alpha = bv.out_alpha;
  const int tile = tile_of_block(bid - bv.block_base, ntiles);
  if (tile >= ntiles) return;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int px = tile_x * B3GS_TILE + (int)((w & 1) * 8 + (lane & 7));
  const int py = tile_y * B3GS_TILE + (int)((w >> 1) * 8 + (lane >> 3));
  const bool inside = px < W && py < H;
  const float fpx = (float)px, fpy = (float)py;
  // round 0: every tile, segment 1 only, and the per-tile "all pixels terminated" flag; round 1 (after the second
  // binning round): only the tiles that received a segment 2, recomputed over segment 1 + segment 2
  // round 2 (b3gs_blend_forward_batch: re-blending a finished forward's state): every tile, both segments
  if (bv.round == 1 && *bv.open_count == 0u) return;   // no tile of this view was left open: nothing to redo
  const TileList tl = tile_list(bv, tile, bv.round != 0);
  if (bv.round == 1 && tl.total == tl.len1) return;
  const int nchunks = (int)((tl.total + CHUNK - 1) / CHUNK);

  // A finished pixel is encoded as Tw == 0 (working transmittance): every later weight is then exactly zero, the
  // saturation test fires again harmlessly, and "is anyone still active" is one compare -- no lane-mask
  // bookkeeping (the mask logic cost ~30 scalar instructions per candidate on the CU's shared scalar unit).
  // T holds the transmittance that is reported (the value before the Gaussian that saturated the pixel).
  float Tw = inside ? 1.0f : 0.0f;
  float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Dp = 0.f, Ac = 0.f;
  uint32_t last_contributor = 0;

  for (int c = 0; c < nchunks; c++) {
    if (__syncthreads_and(Tw == 0.0f)) break;
    stage_chunk(sh, tl, bv.idx_mask, rec, (uint32_t)(c * CHUNK), (float)(tile_x * B3GS_TILE), (float)(tile_y * B3GS_TILE));
    __syncthreads();
    if (TRACE) n_chunks++;
    if (__builtin_amdgcn_ballot_w64(Tw != 0.0f) == 0) continue;  // this quadrant is finished; keep pace with the barriers
#pragma unroll 1
    for (int pw = 0; pw < CHUNK / 64; pw++) {
      u64 m = uniform_u64(sh.mask[w][pw]);
#define B3GS_FWD_CANDIDATE(A, B, Cc, j)                                                                  \
      do {                                                                                               \
        if (TRACE) n_iter++;                                                                             \
        const int gidx = pw * 64 + (j);                                                                  \
        B3GS_FWD_SENS;                                                                                   \
        const float dx = A.x - fpx, dy = A.y - fpy;                                                      \
        const float power = blend_power(A, B.x, dx, dy);                                                 \
        const float alpha = fminf(B3GS_ALPHA_MAX, B.y * __expf(power));                                  \
        /* branch-free: a pixel this Gaussian does not touch blends weight 0, which leaves T, the sums and */ \
        /* last_contributor unchanged; so does a finished pixel (Tw == 0) */                             \
        const bool live = !(power > 0.0f) && !(alpha < B3GS_ALPHA_MIN);                                  \
        const float test_T = Tw * (1.0f - (live ? alpha : 0.0f));                                        \
        const bool stop = test_T < B3GS_T_EPS;  /* would saturate: not blended, pixel done (an active pixel has Tw >= eps) */ \
        const bool blend = live && !stop;                                                                \
        const float wgt = blend ? alpha * Tw : 0.0f;                                                     \
        Cr = __builtin_fmaf(B.z, wgt, Cr);                                                               \
        Cg = __builtin_fmaf(B.w, wgt, Cg);                                                               \
        Cb = __builtin_fmaf(Cc.x, wgt, Cb);                                                              \
        Dp = __builtin_fmaf(Cc.y, wgt, Dp);                                                              \
        Ac += wgt;                                                                                       \
        T = blend ? test_T : T;                                                                          \
        last_contributor = blend ? (uint32_t)(c * CHUNK + gidx + 1) : last_contributor;                  \
        Tw = stop ? 0.0f : test_T;                                                                       \
      } while (0)
      while (m) {
        const int j = __builtin_ctzll(m);
        clear_bit(m, j);
        {
          const float4 A = sh.A[pw * 64 + j], B = sh.B[pw * 64 + j];
          const float2 Cc = *reinterpret_cast<const float2*>(&sh.C[pw * 64 + j]);
          B3GS_FWD_CANDIDATE(A, B, Cc, j);
        }
        // "is any pixel of the quadrant still active" only after every B3GS_FWD_CHECK_EVERY-th candidate: a candidate
        // evaluated after the last pixel finished blends weight zero everywhere, and the check is a third of the loop's
        // scalar instructions (measured on MI355X, every 1 / 2 / 3 / 4 / 6 / 8: 282 / 278 / 274 / 273 / 273 / 275 us)
#pragma unroll
        for (int rep = 1; rep < B3GS_FWD_CHECK_EVERY; rep++) {
          if (m == 0) break;
          const int j2 = __builtin_ctzll(m);
          clear_bit(m, j2);
          const float4 A = sh.A[pw * 64 + j2], B = sh.B[pw * 64 + j2];
          const float2 Cc = *reinterpret_cast<const float2*>(&sh.C[pw * 64 + j2]);
          B3GS_FWD_CANDIDATE(A, B, Cc, j2);
        }
        if (__builtin_amdgcn_ballot_w64(Tw != 0.0f) == 0) break;
      }
      // (fetching the NEXT candidate's record before evaluating the current one -- the backward's scheme, two register sets
      // swapping roles -- measured 285 vs 273 us here: the forward's step is too short for the extra control flow)
#undef B3GS_FWD_CANDIDATE
      if (__builtin_amdgcn_ballot_w64(Tw != 0.0f) == 0) break;
    }
  }
  // (all threads of the workgroup are still here: no early return above)
  const int all_done = (bv.round == 0 && bv.open_rows != nullptr) ? __syncthreads_and(Tw == 0.0f) : 1;
  {   // the tile's backward work: the deepest list position any of its pixels used
    uint32_t m = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0) s_work[w] = m;
    __syncthreads();
    // GeomView::staged: the Gaussians at list positions below the tile's work are the only ones the blend backward of this
    // tile can flush a gradient for; their mark gets this forward's epoch (the chain rule's scan reads only marked rows).
    // (A tile re-blended by the second round marks again, over segment 1 + segment 2.)
    if (bv.staged) {
      const uint32_t wk = max(max(s_work[0], s_work[1]), max(s_work[2], s_work[3]));
      const uint8_t ep = (uint8_t)*bv.epoch;
      for (uint32_t q = tid; q < wk; q += 256u) {
        const uint32_t word = q < tl.len1 ? tl.list1[tl.first1 + q] : tl.list2[tl.first2 + (q - tl.len1)];
        bv.staged[word & bv.idx_mask] = ep;
      }
    }
    if (tid == 0) {
      const uint32_t work = max(max(s_work[0], s_work[1]), max(s_work[2], s_work[3]));
      bv.tile_work[tile] = work;
      if (bv.round == 0 && bv.open_rows != nullptr) {   // two-round forward
        const size_t word = (size_t)tile_y * bv.row_words + (tile_x >> 6);
        const unsigned long long bit = 1ull << (tile_x & 63);
        const bool predicted = bv.pred_rows && (bv.pred_rows[word] & bit);
        if (!all_done) {
          // an unterminated pixel: with the K1 prefix only, the tile is open for the second binning round; with its
          // complete list (predicted open) it is simply finished.  Either way it is predicted open next time.
          if (!predicted) {
            atomicOr(&bv.open_rows[word], bit);
            atomicAdd(bv.open_count, 1u);
          }
          if (bv.pred_next) atomicOr(&bv.pred_next[word], bit);
        } else if (predicted && bv.pred_next && work > 0u) {
          // terminated with the complete list: keep predicting it open unless the deepest Gaussian it used lies well
          // inside segment 1 (depth key below the one at rank 3/4 K1: hysteresis against repairing every other step)
          const uint32_t wq = work - 1u;
          const uint32_t lw = wq < tl.len1 ? tl.list1[tl.first1 + wq] : tl.list2[tl.first2 + (wq - tl.len1)];
          const float zdeep = reinterpret_cast<const float*>(rec + 4 * (size_t)(lw & bv.idx_mask) + 2)[1];
          if (!(__float_as_uint(zdeep) - bv.z_base < *bv.z_clear)) atomicOr(&bv.pred_next[word], bit);
        } else if (!predicted && bv.pred_next && 4u * work > 3u * tl.len1) {
          // terminated, but only in the last quarter of its segment-1 prefix: one optimiser step can push it over the
          // end (measured at the headline: a tile at that margin was left open every other iteration, and every such
          // miss costs the repair kernel's slow path) -- give it its complete list next time
          atomicOr(&bv.pred_next[word], bit);
        }
      }
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
  if (TRACE && lane == 0) {
    unsigned long long* t = trace + 4 * ((size_t)slot * 4 + w);
    t[0] = __builtin_readcyclecounter() - t_start;
    t[1] = (r_start << 32) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull);
    t[2] = n_iter;
    t[3] = n_chunks;
  }
}

template <int CHUNK, bool TRACE>
#ifndef B3GS_FWD_WAVES
#define B3GS_FWD_WAVES 1
#endif
__global__ void __launch_bounds__(256, B3GS_FWD_WAVES) render_fwd_kernel(BlendBatch batch, unsigned long long* __restrict__ trace) {
  __shared__ TileShared<CHUNK> sh;
  __shared__ uint32_t s_work[4];
  // longest-tile-first order of the previous backward of this batch shape, if the buffer holds one (BlendBatch::sig)
  int bid = (int)blockIdx.x;
  if (batch.order && batch.order[batch.sig_off] == B3GS_ORDER_MAGIC && batch.order[batch.sig_off + 1] == batch.sig)
    bid = 8 * (int)batch.order[(blockIdx.x & 7u) * (unsigned)batch.cls_size + (blockIdx.x >> 3)] + (int)(blockIdx.x & 7u);
  render_fwd_tile<CHUNK, TRACE>(select_view(batch, bid), bid, blockIdx.x, sh, s_work, trace);
}

// Second blend pass of a two-round forward (BlendView::round = 1) as a small persistent grid: in the steady state no
// tile was left open (image header word 3 of every view is zero) and the launch is one look at those words; otherwise
// the workgroups stride over all tiles and re-blend the ones that received a segment 2.
constexpr int REBLEND_GRID = 1024;
template <int CHUNK>
__global__ void __launch_bounds__(256) render_fwd_repair_kernel(BlendBatch batch, int total) {
  __shared__ TileShared<CHUNK> sh;
  __shared__ uint32_t s_work[4];
  bool any = false;
#pragma unroll
  for (int k = 0; k < B3GS_MAX_FUSED_VIEWS; k++)
    if (k < batch.n) any = any || *batch.v[k].open_count != 0u;
  // This is the last kernel of a two-round forward: the prediction the first blend pass collected (pred_next) becomes
  // the prediction of the NEXT forward into these image buffers.  Nothing below reads either bitmap (the second pass
  // does not predict), and the next forward's projection needs pred_rows complete before its first workgroup starts.
  // (one workgroup per view -- the grid holds at least eight per view: the six copies were six dependent round trips of workgroup
  // 0, most of what this launch costs when nothing is left to repair)
  if (blockIdx.x < (unsigned)B3GS_MAX_FUSED_VIEWS) {
#pragma unroll
    for (int k = 0; k < B3GS_MAX_FUSED_VIEWS; k++)
      if (k == (int)blockIdx.x && k < batch.n && batch.v[k].pred_next) {
        const BlendView& v = batch.v[k];
        unsigned long long* pr = const_cast<unsigned long long*>(v.pred_rows);
        const int nwords = v.row_words * (v.ntiles / v.grid_x);
        for (int i = (int)threadIdx.x; i < nwords; i += 256) {
          pr[i] = v.pred_next[i];
          v.pred_next[i] = 0ull;
        }
      }
  }
  if (!any) return;
  for (int bid = (int)blockIdx.x; bid < total; bid += (int)gridDim.x) {
    render_fwd_tile<CHUNK, false>(select_view(batch, bid), bid, 0u, sh, s_work, nullptr);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// ---- backward -------------------------------------------------------------------------------
// Cross-lane reduction is what this kernel is organised around: on gfx950 (tools/ubench/valu_rate.hip) a plain
// fp32 VALU op issues at ~2 cycles per wave64 instruction, v_pk_*_f32 at ~4 (packing buys nothing), v_add_f32_dpp,
// v_permlane32_swap, v_exp and v_rcp at ~8, so the textbook 6-step butterfly over 10 gradient components costs
// ~480 cycles per Gaussian -- more than twice the per-pixel arithmetic.  The partials are TRANSPOSED THROUGH LDS:
//   10 x ds_write_addtid_b32  row k of a per-wave [10][68]-float scratch <- component k of every lane
//                             (address = M0 + offset + 4*lane: no address VGPR, half the LDS-path cycles of ds_write_b32)
//    4 x ds_read_b128         lane (k,part) = (lane>>2, lane&3) reads elements [16 part, 16 part+16) of row k
//                             (row stride 68 dwords: the 16-lane groups of ds_read_b128 hit 64 distinct banks)
//   15 plain adds + 2 quad DPP adds -> lanes 4k..4k+3 hold component k summed over the 64 pixels
//    1 x global_atomic_add_f32 with 10 active lanes onto ONE 40-byte scratch row of this Gaussian
// Measured ablations: without the atomic the kernel ran 26 % faster while its 10 lanes went to four separate
// arrays; on one row the atomic is free.  PMC: VALU busy 66 % of all SIMD cycles (incl. the tail), i.e. the
// kernel is VALU/issue bound; trimming 15 % of the VALU instructions bought 1-2 %.
constexpr int RED_STRIDE = 68;
// B3GS_BWD_PIPELINE 1 consumes the reduction reads of Gaussian n only after evaluating Gaussian n+1 (the write -> read
// round trip overlaps the next evaluation) and was round 1's choice at 6 waves per SIMD (80 VGPRs).  Reducing at once
// frees the 16 registers the pending reads occupy: 69 VGPRs, 7 waves per SIMD, no spill -- measured 3 % faster
// (0.107-0.112 -> 0.102-0.105 ms per view, A/B on the same box, tools/ab.sh); 8 waves (64 VGPRs, 5 spilled) is no better.
#ifndef B3GS_BWD_PIPELINE
#define B3GS_BWD_PIPELINE 0
#endif
#ifndef B3GS_BWD_PREFETCH
#define B3GS_BWD_PREFETCH 1   /* fetch the next candidate's record (3 broadcast ds_read_b128) before evaluating the current one */
#endif
#ifndef B3GS_BWD_QWAVES
#define B3GS_BWD_QWAVES 8   /* the quadrant-wave kernel: 55 VGPRs */
#endif
#ifndef B3GS_BWD_WAVES
#define B3GS_BWD_WAVES 7  /* waves per SIMD the register allocator must leave room for */
#endif

template <int CHUNK>
struct TileSharedBwd {
  TileShared<CHUNK> f;
  float red[4][10 * RED_STRIDE];
};

struct BwdPixel {
  float fpx, fpy, T, T_final, bg_dot, dCr, dCg, dCb, dD, dA, Bw, half_w, half_h;
  uint32_t last;
};

// Per-pixel reverse step for one Gaussian; writes this lane's 10 partial gradients to p[].
// Branch-free: a lane the Gaussian does not touch uses alpha = G = 0, which leaves T and the
// "behind" composites unchanged and makes every partial an exact zero.
// dx, dy and the products u = cxx dx, v = cyy dy, nw = -cxy dx come from the evaluation of the exponent.
__device__ __forceinline__ void bwd_eval(BwdPixel& px, const float4& A, const float4& B, float col_b, float depth,
                                         float dx, float dy, float u, float v, float nw, float G, float alpha, bool live,
                                         float (&p)[10]) {
  G = live ? G : 0.0f;
  alpha = live ? alpha : 0.0f;
  // 1/(1-alpha): hardware reciprocal (1 ulp); alpha <= 0.99 keeps it well conditioned
  const float inv_one_m_a = __builtin_amdgcn_rcpf(1.0f - alpha);
  px.T = px.T * inv_one_m_a;
  const float wgt = alpha * px.T;
  // dL/dalpha = T * sum_ch (c_ch - behind_ch) dL/dpixel_ch over the five output channels (r, g, b, depth, alpha).
  // The composite behind the current Gaussian only ever enters through that dot product, and its update
  // behind' = behind + alpha (c - behind) is linear, so ONE scalar Bw = sum_ch behind_ch dL/dpixel_ch is carried
  // instead of five channels.
  float cw = B.z * px.dCr;
  cw = __builtin_fmaf(B.w, px.dCg, cw);
  cw = __builtin_fmaf(col_b, px.dCb, cw);
  cw = __builtin_fmaf(depth, px.dD, cw);
  cw += px.dA;
  const float diff = cw - px.Bw;
  px.Bw = __builtin_fmaf(alpha, diff, px.Bw);
  float dL_da = diff * px.T;
  dL_da = __builtin_fmaf(-px.T_final * inv_one_m_a, px.bg_dot, dL_da);
  // s = dL/d(exponent): mean gets -s Q d (scaled to NDC units), the conic entries -0.5 s d d^T
  const float s = G * (B.y * dL_da);
  p[0] = (s * px.half_w) * -__builtin_fmaf(A.w, dy, u);   // -(cxx dx + cxy dy)
  p[1] = (s * px.half_h) * (nw - v);                       // -(cyy dy + cxy dx)
  const float sh = -0.5f * s;
  const float shx = sh * dx;
  p[2] = shx * dx;
  p[3] = shx * dy;
  p[4] = (sh * dy) * dy;
  p[5] = G * dL_da;
  p[6] = wgt * px.dCr;
  p[7] = wgt * px.dCg;
  p[8] = wgt * px.dCb;
  p[9] = wgt * px.dD;
}

// sensitivity probes (tools only, like B3GS_FWD_SENS): four extra independent instructions of one kind per candidate
#if defined(B3GS_BWD_SENS_SALU)
#define B3GS_BWD_SENS asm volatile("s_add_u32 s2, s2, 1\n s_add_u32 s2, s2, 1\n s_add_u32 s2, s2, 1\n s_add_u32 s2, s2, 1" ::: "s2", "scc")
#elif defined(B3GS_BWD_SENS_VALU)
#define B3GS_BWD_SENS asm volatile("v_add_u32 v60, v60, 1\n v_add_u32 v61, v61, 1\n v_add_u32 v62, v62, 1\n v_add_u32 v63, v63, 1" ::: "v60", "v61", "v62", "v63")
#elif defined(B3GS_BWD_SENS_LDS)
#define B3GS_BWD_SENS asm volatile("ds_read_b128 v[60:63], %0\n s_waitcnt lgkmcnt(0)" :: "v"(0) : "v60", "v61", "v62", "v63", "memory")
#else
#define B3GS_BWD_SENS do { } while (0)
#endif
// where the reduced component of Gaussian g goes (redefined by the quadrant kernel's row mode)
#define B3GS_RED_ADDR(g) (red_base + __umul24((g), red_stride))
// ---- the candidate step of the blend backward, shared by the tile-workgroup kernel and the quadrant-wave kernel below.
// The macros use the surrounding kernel's locals: px, base, pending, q0..q3, pend_g, red_rd, red_base, red_stride,
// red_writer, red_m0 and (TRACE) n_iter / n_live / n_lanes.
#define B3GS_RETIRE_PENDING()                                                          \
  do {                                                                                 \
    float v = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w));       \
    v += ((q2.x + q2.y) + (q2.z + q2.w)) + ((q3.x + q3.y) + (q3.z + q3.w));            \
    /* sum over the four parts of component rk: two fused DPP adds, quad_perm [1,0,3,2] then [2,3,0,1].  As inline */ \
    /* asm: left to the compiler the second add sinks behind the writer branch and becomes v_mov + v_mov_dpp + v_add */ \
    /* (a VALU write needs two wait states before a DPP read of the same register) */   \
    asm("s_nop 1\n\t"                                                                   \
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
        "s_nop 1\n\t"                                                                   \
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" \
        : "+v"(v));                                                                      \
    if (red_writer) unsafeAtomicAdd(B3GS_RED_ADDR(pend_g), v);                         \
  } while (0)

  // one candidate: evaluate, and if any pixel of the quadrant is touched, reduce + queue its partials
#define B3GS_BWD_CANDIDATE(A, B, Cc, J)                                                          \
  do {                                                                                           \
    B3GS_BWD_SENS;                                                                               \
    const float dx_ = A.x - px.fpx, dy_ = A.y - px.fpy;                                          \
    const float u_ = A.z * dx_, v_ = B.x * dy_, nw_ = -A.w * dx_;                                \
    /* == blend_power(): same products, same order (the forward must agree on every skip decision) */ \
    const float power = __builtin_fmaf(nw_, dy_, -0.5f * __builtin_fmaf(v_, dy_, u_ * dx_));     \
    const float G = __expf(power);                                                               \
    const float alpha = fminf(B3GS_ALPHA_MAX, B.y * G);                                          \
    const bool live = (base + (uint32_t)(J) < px.last) && !(power > 0.0f) && !(alpha < B3GS_ALPHA_MIN); \
    if (TRACE) n_iter++;                                                                         \
    if (__builtin_amdgcn_ballot_w64(live) != 0) {                                                \
      if (TRACE) {                                                                               \
        const unsigned long long lm_ = __builtin_amdgcn_ballot_w64(live);                        \
        n_live++; n_lanes += (unsigned)__builtin_popcountll(lm_);                                \
        /* lane = 8 y + x: candidates whose live pixels sit in ONE half of the quadrant (rows 0-3 / 4-7, columns 0-3 / 4-7) */ \
        const bool top_ = !(lm_ >> 32), bot_ = !(lm_ & 0xFFFFFFFFull);                           \
        const bool lft_ = !(lm_ & 0xF0F0F0F0F0F0F0F0ull), rgt_ = !(lm_ & 0x0F0F0F0F0F0F0F0Full); \
        const unsigned code_ = top_ ? 1u : bot_ ? 2u : lft_ ? 3u : rgt_ ? 4u : 0u;                \
        n_half += code_ != 0u;                                                                   \
        /* ... and ADJACENT live candidates in complementary halves: the only pairs one loop trip could serve together */ \
        n_pair += (code_ != 0u && prev_half == (code_ ^ ((code_ <= 2u) ? 3u : 7u)));             \
        prev_half = (n_pair_last_ == n_pair) ? code_ : 0u;                                       \
        n_pair_last_ = n_pair;                                                                   \
      }                                                                                          \
      float p[10];                                                                               \
      bwd_eval(px, A, B, Cc.x, Cc.y, dx_, dy_, u_, v_, nw_, G, alpha, live, p);                  \
      B3GS_ROW_WRITES(p);                                                                        \
      if (B3GS_BWD_PIPELINE && pending) B3GS_RETIRE_PENDING();                                   \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                     \
      __builtin_amdgcn_wave_barrier();                                                           \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                     \
      q0 = red_rd[0]; q1 = red_rd[1]; q2 = red_rd[2]; q3 = red_rd[3];                            \
      pend_g = (uint32_t)__float_as_int(Cc.z);                                                   \
      if (B3GS_BWD_PIPELINE) pending = true;                                                     \
      else B3GS_RETIRE_PENDING();   /* experiment: reduce at once (16 fewer live VGPRs) */        \
    }                                                                                            \
  } while (0)
  // row k of the per-wave scratch <- component k of every lane.  ds_write_addtid_b32 (address = M0 + offset
  // + 4*lane, no address VGPR); s_nop: a SALU write of M0 needs one wait state before an LDS add-TID
  // instruction, which the assembler does not insert inside an asm block.
#define B3GS_ROW_WRITES(p)                                                                       \
  asm volatile(                                                                                  \
      "s_mov_b32 m0, %10\n\t"                                                                    \
      "s_nop 0\n\t"                                                                              \
      "ds_write_addtid_b32 %0 offset:0\n\t"                                                      \
      "ds_write_addtid_b32 %1 offset:272\n\t"                                                    \
      "ds_write_addtid_b32 %2 offset:544\n\t"                                                    \
      "ds_write_addtid_b32 %3 offset:816\n\t"                                                    \
      "ds_write_addtid_b32 %4 offset:1088\n\t"                                                   \
      "ds_write_addtid_b32 %5 offset:1360\n\t"                                                   \
      "ds_write_addtid_b32 %6 offset:1632\n\t"                                                   \
      "ds_write_addtid_b32 %7 offset:1904\n\t"                                                   \
      "ds_write_addtid_b32 %8 offset:2176\n\t"                                                   \
      "ds_write_addtid_b32 %9 offset:2448"                                                       \
      :                                                                                          \
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]),  \
        "v"(p[8]), "v"(p[9]), "s"(red_m0)                                                        \
      : "memory")


template <int CHUNK, bool TRACE>
__global__ void __launch_bounds__(256, B3GS_BWD_WAVES)
    render_bwd_tile_kernel(BlendBatch batch, unsigned long long* __restrict__ trace) {
  __shared__ TileSharedBwd<CHUNK> sh;
  __shared__ uint32_t s_max_last[4];
  const unsigned long long t_start = TRACE ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long r_start = TRACE ? __builtin_amdgcn_s_memrealtime() : 0ull;
  unsigned n_iter = 0, n_live = 0, n_lanes = 0, n_half = 0, n_pair = 0, prev_half = 0, n_pair_last_ = 0;
  // longest-tile-first inside the XCD class (BlendBatch::order); placement never affects results
  const int bid = batch.order ? 8 * (int)batch.order[(blockIdx.x & 7u) * (unsigned)batch.cls_size + (blockIdx.x >> 3)] + (int)(blockIdx.x & 7u)
                              : (int)blockIdx.x;
  const BlendView bv = select_view(batch, bid);
  const int W = bv.W, H = bv.H, grid_x = bv.grid_x, ntiles = bv.ntiles;
  const float4* __restrict__ rec = bv.rec;
  const float* __restrict__ bg = bv.bg;
  const float* __restrict__ final_T = bv.final_T;
  const uint32_t* __restrict__ n_contrib = bv.n_contrib;
  const float* __restrict__ dL_dcolor = bv.dL_dcolor;
  const float* __restrict__ dL_ddepth = bv.dL_ddepth;
  const float* __restrict__ dL_dalpha_img = bv.dL_dalpha;
  float* __restrict__ dL_dmeans2D = bv.dL_dmeans2D;
  float* __restrict__ dL_dcolors = bv.dL_dcolors;
  float* __restrict__ dL_dopacity = bv.dL_dopacity;
  float* __restrict__ dL_dcov3D = bv.dL_dcov3D;
  const unsigned cov_stride = bv.cov_stride;
  const int tile = tile_of_block(bid - bv.block_base, ntiles);
  if (tile >= ntiles) return;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ipx = tile_x * B3GS_TILE + (int)((w & 1) * 8 + (lane & 7));
  const int ipy = tile_y * B3GS_TILE + (int)((w >> 1) * 8 + (lane >> 3));
  const bool inside = ipx < W && ipy < H;
  const TileList tl = tile_list(bv, tile, true);   // n_contrib counts positions of segment 1 + segment 2

  BwdPixel px;
  px.fpx = (float)ipx;
  px.fpy = (float)ipy;
  px.half_w = 0.5f * (float)W;
  px.half_h = 0.5f * (float)H;
  px.last = 0;
  px.T_final = 0.f;
  px.dCr = px.dCg = px.dCb = px.dD = px.dA = 0.f;
  if (inside) {
    const size_t pix = (size_t)ipy * W + ipx, hw = (size_t)H * W;
    px.last = n_contrib[pix];
    px.T_final = final_T[pix];
    px.dCr = dL_dcolor[pix];
    px.dCg = dL_dcolor[hw + pix];
    px.dCb = dL_dcolor[2 * hw + pix];
    if (dL_ddepth) px.dD = dL_ddepth[pix];
    if (dL_dalpha_img) px.dA = dL_dalpha_img[pix];
  }
  px.bg_dot = (bg[0] * px.dCr + bg[1] * px.dCg) + bg[2] * px.dCb;
  px.T = px.T_final;
  px.Bw = 0.f;  // composite "behind" the current Gaussian, dotted with the pixel gradients

  // deepest list position any pixel of the quadrant / tile used
  {
    uint32_t m = px.last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0) s_max_last[w] = m;
  }
  __syncthreads();
  const uint32_t max_last = max(max(s_max_last[0], s_max_last[1]), max(s_max_last[2], s_max_last[3]));
  if (max_last == 0) return;
  const uint32_t wave_last = __builtin_amdgcn_readfirstlane(s_max_last[w]);
  // Every wave of every tile is resident from the start and a wave is one serial chain, so the kernel
  // ends with the longest chain: give waves with deeper lists issue priority over shorter ones.
  if (wave_last > 400) __builtin_amdgcn_s_setprio(3);
  else if (wave_last > 330) __builtin_amdgcn_s_setprio(2);
  else if (wave_last > 260) __builtin_amdgcn_s_setprio(1);

  // reduction role of this lane: component k = lane>>2 (valid for k < 10), part = lane&3; the
  // component's destination array, row stride (floats) and column
  const unsigned rk = lane >> 2, rpart = lane & 3;
  float* red_base;   // destination array + column of this lane's component
  unsigned red_stride;
  if (rk < 2) { red_base = dL_dmeans2D + rk; red_stride = bv.m2d_stride; }
  else if (rk < 5) { red_base = dL_dcov3D + (rk - 2); red_stride = cov_stride; }
  else if (rk == 5) { red_base = dL_dopacity; red_stride = bv.op_stride; }
  else if (rk < 9) { red_base = dL_dcolors + (rk - 6); red_stride = bv.col_stride; }
  else { red_base = dL_dcov3D + 3; red_stride = cov_stride; }
  const bool red_writer = (rk < 10u) && (rpart == 0);
  float* const red = sh.red[w];
  static_assert(RED_STRIDE * 4 == 272, "row offsets of the ds_write_addtid_b32 block");
  // LDS byte offset of this wave's scratch (an LDS pointer's value is its offset in the workgroup's allocation)
  const uint32_t red_m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)red);
  const float4* const red_rd = reinterpret_cast<const float4*>(red + (rk < 10 ? rk : 0) * RED_STRIDE + rpart * 16);

  // Software pipeline across Gaussians (a wave is an in-order machine and every tile's waves are
  // resident from the start, so the kernel's length is the longest wave's serial chain):
  //   * the LDS row reads of Gaussian n's reduction are ISSUED right after its row writes but
  //     CONSUMED (15 adds, 2 DPP, atomic) only after the evaluation of the next live Gaussian, so
  //     the write->read round trip overlaps ~150 VALU instructions.  No double buffer is needed: a
  //     wave's LDS operations complete in program order, so the pending reads capture their data
  //     before the next Gaussian's row writes execute.
  //   * the record of the next list entry is fetched before the current one is evaluated.
  float4 q0, q1, q2, q3;      // pending reduction reads
  uint32_t pend_g = 0;        // Gaussian they belong to (P < 2^24: 24-bit multiply for the row offset)
  bool pending = false;
  for (int c = (int)((max_last - 1) / CHUNK); c >= 0; c--) {
    stage_chunk(sh.f, tl, bv.idx_mask, rec, (uint32_t)(c * CHUNK), (float)(tile_x * B3GS_TILE), (float)(tile_y * B3GS_TILE));
    __syncthreads();
#pragma unroll 1
    for (int pw = CHUNK / 64 - 1; pw >= 0; pw--) {
      const uint32_t base = (uint32_t)(c * CHUNK + pw * 64);
      if (base >= wave_last) continue;
      u64 m = uniform_u64(sh.f.mask[w][pw]);
      const uint32_t lim = wave_last - base;  // positions >= wave_last were never reached by this quadrant
      if (lim < 64) m &= (1ull << lim) - 1ull;
      if (m == 0) continue;
      // Walk the set bits from the back.  The record of the NEXT candidate (three broadcast ds_read_b128:
      // A, B, C incl. the Gaussian index) is fetched before the current one is evaluated; the loop is
      // unrolled by two so the two register sets swap roles instead of being copied.
      const float4* const sA = sh.f.A + pw * 64;
      const float4* const sB = sh.f.B + pw * 64;
      const float4* const sC = sh.f.C + pw * 64;
#if B3GS_BWD_PREFETCH
      int j = 63 - __builtin_clzll(m);
      clear_bit(m, j);
      float4 A0 = sA[j], B0 = sB[j], C0 = sC[j], A1, B1, C1;
      while (true) {
        bool more = m != 0;
        int jn = more ? 63 - __builtin_clzll(m) : j;   // (re-reads the current record on the last entry)
        clear_bit(m, jn);
        A1 = sA[jn]; B1 = sB[jn]; C1 = sC[jn];
        B3GS_BWD_CANDIDATE(A0, B0, C0, j);
        if (!more) break;
        j = jn;
        more = m != 0;
        jn = more ? 63 - __builtin_clzll(m) : j;
        clear_bit(m, jn);
        A0 = sA[jn]; B0 = sB[jn]; C0 = sC[jn];
        B3GS_BWD_CANDIDATE(A1, B1, C1, j);
        if (!more) break;
        j = jn;
      }
#else
      while (m) {
        const int j = 63 - __builtin_clzll(m);
        clear_bit(m, j);
        const float4 A0 = sA[j], B0 = sB[j], C0 = sC[j];
        B3GS_BWD_CANDIDATE(A0, B0, C0, j);
      }
#endif
    }
    __syncthreads();
  }
  if (pending) B3GS_RETIRE_PENDING();
  if (TRACE && lane == 0) {
    unsigned long long* t = trace + 4 * ((size_t)blockIdx.x * 4 + w);
    t[0] = __builtin_readcyclecounter() - t_start;            // shader cycles this wave lived
    t[1] = (r_start << 32) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull);  // 100 MHz wall clock: start | end
    t[2] = (unsigned long long)n_iter | ((unsigned long long)(n_half & 0xFFFFu) << 32) | ((unsigned long long)(n_pair & 0xFFFFu) << 48);
    t[3] = (unsigned long long)n_live | ((unsigned long long)n_lanes << 32);   // live iterations | live lanes summed
  }
}

// ---------------------------------------------------------------------------------------------
// Blend backward with ONE WAVE PER WORKGROUP: workgroup = (tile, 8x8 quadrant).  The tile-workgroup kernel above
// couples the four quadrant waves of a tile through two barriers per 64 staged list entries (one wave stages, three
// wait; then all wait for the slowest quadrant) and holds the tile's wave slots until its heaviest quadrant is done
// (tools/bwd_trace_batched.py: 11.5 % of the wave-slot time is held by waves that have finished; jobs of 4 x ~100 us
// leave the machine emptying for the last 15 % of the launch).  Here every quadrant stages its own 64 entries -- the
// record gathers are repeated by the four quadrants of a tile (L1 / L2 hits: the four workgroups are dispatched
// back to back) and every lane tests ITS entry against ITS quadrant only, so the test work is the same -- walks
// only its own depth (wave_last), has no barrier at all, and is a job a quarter as long.
template <int SC>
struct WaveSharedBwd {
  float4 A[SC], B[SC], C[SC];
  float red[10 * RED_STRIDE];
};
// REGS: the staged records stay in the staging lanes' REGISTERS (lane t holds list entry base + t) and the record of
// candidate j is fetched with eleven v_readlane_b32 (wave-uniform: the values land in SGPRs) instead of three broadcast
// LDS reads.  The backward is bound by the LDS pipe -- per candidate and wave: 10 ds_write_addtid_b32 (2 cycles each) + 4
// ds_read_b128 (8 each) of the reduction + 2 ds_read_b128 + 1 ds_read_b96 of the record (8 each, broadcast or not) = 76
// cycles, x 4 SIMDs = 304 per CU and candidate-quad against the measured 290 -- so the record reads are a third of it.
// ROWS: every view's ten gradient components are one row of `row_len` floats per Gaussian (the fused path's scratch rows:
// conic xx, xy, yy, depth | mean2D x, y | colour r, g, b | opacity at columns 0..9 of dL_dcov3D's array): the atomic's
// address is  row array + (g * row_len + column) * 4  -- one scalar multiply and one vector add instead of a 24-bit
// multiply, a sign extension and a 64-bit shift-add per candidate.
template <bool TRACE, int SC, bool REGS, bool ROWS>
__global__ void __launch_bounds__(64, REGS ? B3GS_BWD_QWAVES : B3GS_BWD_WAVES)
    render_bwd_kernel(BlendBatch batch, unsigned long long* __restrict__ trace) {
  __shared__ WaveSharedBwd<REGS ? 1 : SC> sh;
  const unsigned long long t_start = TRACE ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long r_start = TRACE ? __builtin_amdgcn_s_memrealtime() : 0ull;
  unsigned n_iter = 0, n_live = 0, n_lanes = 0, n_half = 0, n_pair = 0, prev_half = 0, n_pair_last_ = 0;
#ifdef B3GS_BWD_QUAD_LINEAR
  const unsigned w = blockIdx.x & 3u, blk = blockIdx.x >> 2;     // quadrant, tile-level block
#else
  // workgroup g runs on XCD g % 8: the four quadrants of a tile-level block stay on that block's XCD (its L2 holds the
  // tile's records), i.e. quadrant = bits 3..4, tile-level block = (g >> 5) * 8 + g % 8
  const unsigned w = (blockIdx.x >> 3) & 3u, blk = ((blockIdx.x >> 5) << 3) | (blockIdx.x & 7u);
#endif
  // longest-tile-first inside the XCD class (BlendBatch::order); placement never affects results
  const int bid = batch.order ? 8 * (int)batch.order[(blk & 7u) * (unsigned)batch.cls_size + (blk >> 3)] + (int)(blk & 7u) : (int)blk;
  const BlendView bv = select_view(batch, bid);
  const int W = bv.W, H = bv.H, grid_x = bv.grid_x, ntiles = bv.ntiles;
  const float4* __restrict__ rec = bv.rec;
  const float* __restrict__ bg = bv.bg;
  float* __restrict__ dL_dmeans2D = bv.dL_dmeans2D;
  float* __restrict__ dL_dcolors = bv.dL_dcolors;
  float* __restrict__ dL_dopacity = bv.dL_dopacity;
  float* __restrict__ dL_dcov3D = bv.dL_dcov3D;
  const unsigned cov_stride = bv.cov_stride;
  const int tile = tile_of_block(bid - bv.block_base, ntiles);
  if (tile >= ntiles) return;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const unsigned lane = threadIdx.x;
  const int ipx = tile_x * B3GS_TILE + (int)((w & 1) * 8 + (lane & 7));
  const int ipy = tile_y * B3GS_TILE + (int)((w >> 1) * 8 + (lane >> 3));
  const bool inside = ipx < W && ipy < H;
  const TileList tl = tile_list(bv, tile, true);   // n_contrib counts positions of segment 1 + segment 2

  BwdPixel px;
  px.fpx = (float)ipx;
  px.fpy = (float)ipy;
  px.half_w = 0.5f * (float)W;
  px.half_h = 0.5f * (float)H;
  px.last = 0;
  px.T_final = 0.f;
  px.dCr = px.dCg = px.dCb = px.dD = px.dA = 0.f;
  if (inside) {
    const size_t pix = (size_t)ipy * W + ipx, hw = (size_t)H * W;
    px.last = bv.n_contrib[pix];
    px.T_final = bv.final_T[pix];
    px.dCr = bv.dL_dcolor[pix];
    px.dCg = bv.dL_dcolor[hw + pix];
    px.dCb = bv.dL_dcolor[2 * hw + pix];
    if (bv.dL_ddepth) px.dD = bv.dL_ddepth[pix];
    if (bv.dL_dalpha) px.dA = bv.dL_dalpha[pix];
  }
  px.bg_dot = (bg[0] * px.dCr + bg[1] * px.dCg) + bg[2] * px.dCb;
  px.T = px.T_final;
  px.Bw = 0.f;
  uint32_t wave_last = px.last;   // deepest list position any pixel of the quadrant used
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, d, 64));
  wave_last = __builtin_amdgcn_readfirstlane(wave_last);
  if (wave_last == 0) {
    if (TRACE && lane == 0) {   // (the trace buffer is reused: an idle quadrant leaves an empty record, not a stale one)
      unsigned long long* t = trace + 4 * ((size_t)blk * 4 + w);
      t[0] = 0ull;
      t[1] = (r_start << 32) | (r_start & 0xFFFFFFFFull);
      t[2] = 0ull;
      t[3] = 0ull;
    }
    return;
  }
  if (wave_last > 400) __builtin_amdgcn_s_setprio(3);
  else if (wave_last > 330) __builtin_amdgcn_s_setprio(2);
  else if (wave_last > 260) __builtin_amdgcn_s_setprio(1);

  const unsigned rk = lane >> 2, rpart = lane & 3;
  float* red_base;
  unsigned red_stride;
  if (rk < 2) { red_base = dL_dmeans2D + rk; red_stride = bv.m2d_stride; }
  else if (rk < 5) { red_base = dL_dcov3D + (rk - 2); red_stride = cov_stride; }
  else if (rk == 5) { red_base = dL_dopacity; red_stride = bv.op_stride; }
  else if (rk < 9) { red_base = dL_dcolors + (rk - 6); red_stride = bv.col_stride; }
  else { red_base = dL_dcov3D + 3; red_stride = cov_stride; }
  const bool red_writer = (rk < 10u) && (rpart == 0);
  // row mode: column of this lane's component inside the Gaussian's row (see ROWS above)
  const uint32_t red_col4 = 4u * (rk < 2 ? 4u + rk : (rk < 5 ? rk - 2u : (rk == 5 ? 9u : (rk < 9 ? rk : 3u))));
  const uint32_t row_bytes = 4u * cov_stride;
  char* const row0 = reinterpret_cast<char*>(dL_dcov3D);
#undef B3GS_RED_ADDR
#define B3GS_RED_ADDR(g) (ROWS ? reinterpret_cast<float*>(row0 + ((g) * row_bytes + red_col4)) : red_base + __umul24((g), red_stride))
  float* const red = sh.red;
  const uint32_t red_m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)red);
  const float4* const red_rd = reinterpret_cast<const float4*>(red + (rk < 10 ? rk : 0) * RED_STRIDE + rpart * 16);
  float4 q0, q1, q2, q3;
  uint32_t pend_g = 0;
  bool pending = false;

  // this quadrant's pixel rectangle relative to the tile origin
  const float qx0 = (float)((w & 1) * 8), qy0 = (float)((w >> 1) * 8);
  const float tile_px = (float)(tile_x * B3GS_TILE), tile_py = (float)(tile_y * B3GS_TILE);
  for (int c = (int)((wave_last - 1) / (uint32_t)SC); c >= 0; c--) {
    const uint32_t base = (uint32_t)c * (uint32_t)SC;
    // ---- stage 64 list entries; the lane's entry against THIS quadrant (bounding box, then the exact test of stage_chunk)
    const uint32_t q = base + lane;
    bool hit = false;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    int id = 0;
    if ((SC == 64 || lane < (unsigned)SC) && q < tl.total && q < wave_last) {
      const uint32_t word = q < tl.len1 ? tl.list1[tl.first1 + q] : tl.list2[tl.first2 + (q - tl.len1)];
      id = (int)(word & bv.idx_mask);
      const float4* r = rec + 4 * (size_t)id;
      r0 = r[0]; r1 = r[1]; r2 = r[2];
      if (!REGS) {
        sh.A[lane] = r0;
        sh.B[lane] = r1;
        sh.C[lane] = make_float4(r2.x, r2.y, __int_as_float(id), 0.f);
      }
      const float ax = (tile_px + qx0) - r0.x, bx = ax + 7.0f;   // quadrant rectangle relative to the mean
      const float ay = (tile_py + qy0) - r0.y, by = ay + 7.0f;
      hit = (ax <= r2.z) && (bx >= -r2.z) && (ay <= r2.w) && (by >= -r2.w);
      if (hit) {
        const float cxx = r0.z, cxy = r0.w, cyy = r1.x;
        const float tau = __logf(255.0f * r1.y) * 1.0005f + 2e-3f;
        const float icx = __builtin_amdgcn_rcpf(cxx), icy = __builtin_amdgcn_rcpf(cyy);
        const bool in_rect = (ax <= 0.0f) && (bx >= 0.0f) && (ay <= 0.0f) && (by >= 0.0f);
        float fmin = 3.0e38f;
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float dx = e ? bx : ax;                                   // vertical edge
          const float dy = fminf(fmaxf(-cxy * dx * icy, ay), by);
          fmin = fminf(fmin, 0.5f * (cxx * dx * dx + cyy * dy * dy) + cxy * dx * dy);
          const float ey = e ? by : ay;                                   // horizontal edge
          const float ex = fminf(fmaxf(-cxy * ey * icx, ax), bx);
          fmin = fminf(fmin, 0.5f * (cxx * ex * ex + cyy * ey * ey) + cxy * ex * ey);
        }
        hit = in_rect || !(fmin > tau);
      }
    }
    u64 m = __ballot(hit);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (REGS) {
      // the candidate's Gaussian index from the staging lane (one v_readlane_b32), its record through the SCALAR cache
      // (s_load from the record array: nothing in this launch writes it; the staging lanes' gathers have just pulled the
      // line into L2).  Measured alternatives: all eleven values by v_readlane_b32 (803 us: a cross-lane read costs ~8
      // cycles of the vector ALU); both s_loads forced into one issue by inline asm (same time).
      typedef const float __attribute__((address_space(4))) cfloat_k;
      while (m) {
        const int j = 63 - __builtin_clzll(m);
        clear_bit(m, j);
        const int idj = __builtin_amdgcn_readlane(id, j);
        cfloat_k* rs = (cfloat_k*)(uintptr_t)(rec + 4 * (size_t)idj);
        const float4 A0 = make_float4(rs[0], rs[1], rs[2], rs[3]);
        const float4 B0 = make_float4(rs[4], rs[5], rs[6], rs[7]);
        const float4 C0 = make_float4(rs[8], rs[9], __int_as_float(idj), 0.f);
        B3GS_BWD_CANDIDATE(A0, B0, C0, j);
      }
    } else if (m != 0) {
      const float4* const sA = sh.A;
      const float4* const sB = sh.B;
      const float4* const sC = sh.C;
      int j = 63 - __builtin_clzll(m);
      clear_bit(m, j);
      float4 A0 = sA[j], B0 = sB[j], C0 = sC[j], A1, B1, C1;
      while (true) {
        bool more = m != 0;
        int jn = more ? 63 - __builtin_clzll(m) : j;   // (re-reads the current record on the last entry)
        clear_bit(m, jn);
        A1 = sA[jn]; B1 = sB[jn]; C1 = sC[jn];
        B3GS_BWD_CANDIDATE(A0, B0, C0, j);
        if (!more) break;
        j = jn;
        more = m != 0;
        jn = more ? 63 - __builtin_clzll(m) : j;
        clear_bit(m, jn);
        A0 = sA[jn]; B0 = sB[jn]; C0 = sC[jn];
        B3GS_BWD_CANDIDATE(A1, B1, C1, j);
        if (!more) break;
        j = jn;
      }
    }
    // (the next round's record stores follow this round's record reads in the wave's in-order LDS queue)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (pending) B3GS_RETIRE_PENDING();
  if (TRACE && lane == 0) {
    unsigned long long* t = trace + 4 * ((size_t)blk * 4 + w);
    t[0] = __builtin_readcyclecounter() - t_start;
    t[1] = (r_start << 32) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull);
    t[2] = (unsigned long long)n_iter | ((unsigned long long)(n_half & 0xFFFFu) << 32) | ((unsigned long long)(n_pair & 0xFFFFu) << 48);
    t[3] = (unsigned long long)n_live | ((unsigned long long)n_lanes << 32);
  }
}
#undef B3GS_BWD_CANDIDATE
#undef B3GS_ROW_WRITES
#undef B3GS_RETIRE_PENDING
#undef B3GS_RED_ADDR

#ifndef B3GS_FWD_CHUNK
#define B3GS_FWD_CHUNK 256
#endif
constexpr int FWD_CHUNK = B3GS_FWD_CHUNK;
unsigned long long* g_bwd_trace = nullptr;
size_t g_bwd_trace_words = 0;
unsigned long long* trace_buffer(int nblocks) {  // debug (B3GS_FWD_TRACE / B3GS_BWD_TRACE): per-wave cycle trace
  if (g_bwd_trace && g_bwd_trace_words < (size_t)nblocks * 16) {
    (void)hipFree(g_bwd_trace);
    g_bwd_trace = nullptr;
  }
  if (!g_bwd_trace) {
    (void)hipMalloc((void**)&g_bwd_trace, (size_t)nblocks * 16 * sizeof(unsigned long long));
    g_bwd_trace_words = (size_t)nblocks * 16;
  }
  return g_bwd_trace;
}
constexpr int BWD_CHUNK = 64;

}  // namespace

namespace {
int blocks_of(const BlendView& v) { return ((v.ntiles + 7) / 8) * 8; }
}  // namespace

namespace {
// where the signature of an order array sits, and what it says: (views, total blocks) of the batch it was built for
bool order_fits(const BlendBatch& batch, int total, uint32_t* sig_off, uint32_t* sig) {
  const size_t words = (size_t)B3GS_MAX_FUSED_VIEWS * (size_t)(batch.v[0].ntiles + 8);
  *sig_off = (uint32_t)(words - 2);
  *sig = (uint32_t)total * 16u + (uint32_t)batch.n;
  return (size_t)total <= words - 2;
}
}  // namespace

void b3gs_launch_blend_forward(BlendBatch batch, hipStream_t s) {
  int total = 0;
  for (int k = 0; k < batch.n; k++) {
    batch.v[k].block_base = total;
    total += blocks_of(batch.v[k]);
  }
  if (total <= 0) return;
  static const bool lpt = getenv("B3GS_NO_LPT") == nullptr && getenv("B3GS_NO_FWD_LPT") == nullptr;
  batch.order = nullptr;
  batch.cls_size = total / 8;
  if (lpt && batch.order_buf && order_fits(batch, total, &batch.sig_off, &batch.sig)) batch.order = batch.order_buf;
  if (batch.v[0].round == 1) {   // second pass of a two-round forward: small persistent grid, usually nothing to do
    batch.order = nullptr;
    hipLaunchKernelGGL((render_fwd_repair_kernel<FWD_CHUNK>), dim3(total < REBLEND_GRID ? total : REBLEND_GRID), dim3(256), 0, s,
                       batch, total);
    return;
  }
  static const bool fwd_trace = getenv("B3GS_FWD_TRACE") != nullptr;   // (debug switches: read once per process)
  if (fwd_trace)  // debug: per-wave cycle trace (tools/bwd_trace.py fwd)
    hipLaunchKernelGGL((render_fwd_kernel<FWD_CHUNK, true>), dim3(total), dim3(256), 0, s, batch, trace_buffer(total));
  else
    hipLaunchKernelGGL((render_fwd_kernel<FWD_CHUNK, false>), dim3(total), dim3(256), 0, s, batch, nullptr);
}

namespace {
// One workgroup per XCD class c: sort the class's cls_size default positions by decreasing tile work (counting sort
// over 256 buckets of work / max work; the order inside a bucket is arbitrary: it only schedules).
constexpr int ORDER_CACHE = 4096;
// view_groups > 1: the views of the batch are cut into that many consecutive groups and the order is group-major,
// longest-tile-first inside a group (256 / groups work levels each): the tiles in flight then belong to n / groups views, whose
// 64-byte records and 40-byte scratch rows (104 MB per view at 1M Gaussians) are what L2 and the 256-MB Infinity Cache have to
// hold -- all six views of the headline are 624 MB.  Round 6, same box, alternating: 662 iters/s with one group, 669-672 with two,
// 667-671 with three, 668-670 with six (blend backward 86-92 -> 83-84 us per view); scratch rows padded to 64 bytes -- a larger
// footprint, no line-straddling atomics -- made the same kernel 7 % SLOWER, which is what pointed here.
__global__ void __launch_bounds__(256) blend_order_kernel(BlendBatch batch, uint32_t* __restrict__ order, int cls_size,
                                                          int view_groups) {
  __shared__ uint32_t hist[256], cursor[256], s_max[4], tmp[8];
  const int c = (int)blockIdx.x;
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const uint32_t levels = 256u / (uint32_t)view_groups;
  // work (low 24 bits) | group of the tile's view (high bits)
  auto work_of = [&](int q) -> uint32_t {
    const int bid = 8 * q + c;
    int k = 0;
    for (int j = 1; j < batch.n; j++) k = bid >= batch.v[j].block_base ? j : k;
    const BlendView bv = select_view(batch, bid);
    const int tile = tile_of_block(bid - bv.block_base, bv.ntiles);
    const uint32_t wk = tile < bv.ntiles ? min(bv.tile_work[tile], 0xFFFFFFu) : 0u;
    return wk | ((uint32_t)(k * view_groups / batch.n) << 24);
  };
  // the work of the class's tiles is read once into LDS (the three passes below recompute nothing); classes larger than
  // the cache fall back to reading it again
  __shared__ uint32_t s_work[ORDER_CACHE];
  for (int q = (int)threadIdx.x; q < cls_size && q < ORDER_CACHE; q += 256) s_work[q] = work_of(q);
  __syncthreads();
  auto work_at = [&](int q) -> uint32_t { return q < ORDER_CACHE ? s_work[q] : work_of(q); };
  uint32_t m = 0;
  for (int q = (int)threadIdx.x; q < cls_size; q += 256) m = max(m, work_at(q) & 0xFFFFFFu);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
  if (lane == 0) s_max[w] = m;
  hist[threadIdx.x] = 0;
  __syncthreads();
  const float scale = (float)(levels - 1u) / (float)max(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])), 1u);
  auto bucket_of = [&](uint32_t wk) -> uint32_t {   // group-major, heaviest first inside the group
    return (wk >> 24) * levels + (levels - 1u) - (uint32_t)((float)(wk & 0xFFFFFFu) * scale);
  };
  for (int q = (int)threadIdx.x; q < cls_size; q += 256) atomicAdd(&hist[bucket_of(work_at(q))], 1u);
  __syncthreads();
  {   // exclusive scan of the 256 bucket counts (bucket 0 = heaviest)
    const uint32_t v = hist[threadIdx.x];
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
      if (lane >= (unsigned)d) inc += o;
    }
    if (lane == 63) tmp[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (unsigned k = 0; k < w; k++) base += tmp[k];
    cursor[threadIdx.x] = base + inc - v;
  }
  __syncthreads();
  for (int q = (int)threadIdx.x; q < cls_size; q += 256) {
    const uint32_t b = bucket_of(work_at(q));
    order[(size_t)c * cls_size + atomicAdd(&cursor[b], 1u)] = (uint32_t)q;
  }
  if (c == 0 && threadIdx.x == 0) {   // the next forward of this batch shape may reuse the order
    order[batch.sig_off] = B3GS_ORDER_MAGIC;
    order[batch.sig_off + 1] = batch.sig;
  }
}
}  // namespace

void b3gs_launch_blend_backward(BlendBatch batch, hipStream_t s) {
  int total = 0;
  for (int k = 0; k < batch.n; k++) {
    batch.v[k].block_base = total;
    total += blocks_of(batch.v[k]);
  }
  if (total <= 0) return;
  // longest-tile-first: the order array lives in view 0's image buffer (sized for 8 views of its tile count; batches of
  // mixed resolutions that do not fit keep the default order)
  static const bool lpt = getenv("B3GS_NO_LPT") == nullptr;
  uint32_t* order = batch.order_buf;
  batch.order = nullptr;
  batch.cls_size = total / 8;
  if (lpt && order && order_fits(batch, total, &batch.sig_off, &batch.sig)) {
    // default: at most three views per group (6 views: two groups; pairs stay together); B3GS_BWD_VIEW_GROUPS overrides (A/B)
    static const int groups_env = getenv("B3GS_BWD_VIEW_GROUPS") ? atoi(getenv("B3GS_BWD_VIEW_GROUPS")) : 0;
    const int want = groups_env > 0 ? groups_env : (batch.n + 2) / 3;
    const int groups = want < 1 ? 1 : (want > batch.n ? batch.n : want);
    hipLaunchKernelGGL(blend_order_kernel, dim3(8), dim3(256), 0, s, batch, order, batch.cls_size, groups);
    batch.order = order;
  }
  // B3GS_BWD_TRACE=1: per-wave {cycles, wall start|end, iterations, live iterations} (tools/bwd_trace.py)
  static const bool bwd_trace = getenv("B3GS_BWD_TRACE") != nullptr;
  unsigned long long* trace = bwd_trace ? trace_buffer(total) : nullptr;
  // B3GS_BWD_CHUNK (64/128/256) is a tuning knob for experiments; 64 measured best on MI355X
  static const int bwd_chunk = getenv("B3GS_BWD_CHUNK") ? atoi(getenv("B3GS_BWD_CHUNK")) : BWD_CHUNK;
  // default: one wave per workgroup (quadrant), records through the scalar cache; B3GS_BWD_KERNEL=tile selects the
  // tile-workgroup kernel (four quadrant waves sharing an LDS stage), =wave the quadrant kernel with LDS-staged records
  static const char* which = getenv("B3GS_BWD_KERNEL");
  const bool tile_wg = which && !strcmp(which, "tile"), wave_lds = which && !strcmp(which, "wave");
  if (!tile_wg) {
    bool rows = true;   // every view: one row per Gaussian holding the ten components at the scratch-row columns
    for (int k = 0; k < batch.n; k++) {
      const BlendView& v = batch.v[k];
      rows = rows && v.m2d_stride == v.cov_stride && v.col_stride == v.cov_stride && v.op_stride == v.cov_stride &&
             v.dL_dmeans2D == v.dL_dcov3D + 4 && v.dL_dcolors == v.dL_dcov3D + 6 && v.dL_dopacity == v.dL_dcov3D + 9 &&
             (uint64_t)v.cov_stride * 4u * (uint64_t)(v.idx_mask == 0xFFFFFFFFu ? 0x7FFFFFFFu : v.idx_mask) < 0xFFFFFFFFull;
    }
    if (trace) hipLaunchKernelGGL((render_bwd_kernel<true, 64, true, false>), dim3(total * 4), dim3(64), 0, s, batch, trace);
    else if (wave_lds) hipLaunchKernelGGL((render_bwd_kernel<false, 64, false, false>), dim3(total * 4), dim3(64), 0, s, batch, trace);
    else if (rows) hipLaunchKernelGGL((render_bwd_kernel<false, 64, true, true>), dim3(total * 4), dim3(64), 0, s, batch, trace);
    else hipLaunchKernelGGL((render_bwd_kernel<false, 64, true, false>), dim3(total * 4), dim3(64), 0, s, batch, trace);
    return;
  }
  if (trace) hipLaunchKernelGGL((render_bwd_tile_kernel<64, true>), dim3(total), dim3(256), 0, s, batch, trace);
  else if (bwd_chunk == 256) hipLaunchKernelGGL((render_bwd_tile_kernel<256, false>), dim3(total), dim3(256), 0, s, batch, trace);
  else if (bwd_chunk == 128) hipLaunchKernelGGL((render_bwd_tile_kernel<128, false>), dim3(total), dim3(256), 0, s, batch, trace);
  else hipLaunchKernelGGL((render_bwd_tile_kernel<64, false>), dim3(total), dim3(256), 0, s, batch, trace);
}

BlendView b3gs_blend_view(const B3gsScene& sc, const GeomView& g, const BinView& b, const ImgView& im) {
  BlendView v;
  memset(&v, 0, sizeof(v));
  v.W = sc.W;
  v.H = sc.H;
  v.grid_x = (sc.W + B3GS_TILE - 1) / B3GS_TILE;
  v.ntiles = v.grid_x * ((sc.H + B3GS_TILE - 1) / B3GS_TILE);
  v.ranges = im.ranges;
  v.point_list = b.val[0];
  v.ranges2 = im.ranges2;      // empty unless a second binning round ran (b3gs_launch_round2_batch); segment 2 sits
  v.point_list2 = b.val[0];    // behind segment 1 in the same array, its ranges are absolute positions
  v.tile_work = im.tile_work;
  v.open_rows = nullptr;
  v.pred_rows = nullptr;
  v.pred_next = nullptr;
  v.z_clear = nullptr;
  v.z_base = 0u;
  v.open_count = im.header + 3;
  v.row_words = (v.grid_x + 63) / 64;
  v.round = 0;
  const int idx_bits = b3gs_packed_idx_bits(sc.P, sc.W, sc.H);
  v.idx_mask = (idx_bits < 0 || idx_bits >= 32) ? 0xFFFFFFFFu : ((1u << idx_bits) - 1u);
  v.rec = g.rec;
  static const bool no_staged = getenv("B3GS_NO_STAGED") != nullptr;   // (A/B switch, read once: the scan then reads every row)
  v.staged = no_staged ? nullptr : g.staged;
  v.epoch = g.header + B3GS_GEOM_EPOCH;
  v.bg = sc.background;
  v.final_T = im.final_T;
  v.n_contrib = im.n_contrib;
  return v;
}

// debug only: copy the last backward's per-wave cycle trace to the host (returns words copied)
extern "C" size_t b3gs_debug_bwd_trace(unsigned long long* host, size_t max_words) {
  if (!g_bwd_trace) return 0;
  (void)hipDeviceSynchronize();
  size_t n = g_bwd_trace_words < max_words ? g_bwd_trace_words : max_words;
  (void)hipMemcpy(host, g_bwd_trace, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  return n;
}
