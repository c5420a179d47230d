// Tile binning: depth order, instance emission, stable tile split, per-tile ranges -- for one view or
// for a BATCH of views per launch (blockIdx.y = view).
//
// The published rasterizer (and oracle/tile_ref.c) sorts N (tile<<32 | depth-bits) 64-bit keys in
// one stable radix sort, 6 byte-passes over N at 800x600.  The same permutation is produced here
// with far less HBM traffic by sorting in two levels:
//   1. stable LSD radix sort of the P Gaussians by their 32 depth bits (4 passes over P;
//      Gaussians behind the near plane carry key 0xFFFFFFFF and sink to the end; ties keep index order),
//   2. emit the (tile, index) instances in that depth order (wave-cooperative expansion:
//      coalesced writes regardless of how many tiles one Gaussian covers),
//   3. stable LSD radix sort of the N instances by tile id only (ceil(log2(tiles)/8) passes,
//      2 at 800x600 and 1600x1600) -- stability keeps the depth/index order inside each tile.
// Resulting point_list is bit-identical to the 64-bit sort (tests/test_gpu_parity.py).
//
// Batching: one view's pass is ~250 workgroups of 4 waves on a 256-CU part (one wave per SIMD: pure
// latency).  The training iteration renders 6 views of the same Gaussians, so every launch here takes
// up to B3GS_MAX_FUSED_VIEWS jobs.  Views whose view-space depth is identical -- the binocular pairs:
// the shifted camera moves along the camera x axis only (utils/pose_utils.py:148-163), so row z of
// the view matrix is unchanged -- share ONE depth sort (`order_from`).
//
// All kernels read the element count from device memory (N is produced on the device), so the
// same launches serve the sync-free forward; grids are sized from a host-side bound.
#include "b3gs_internal.h"
#include <utility>
#include <cstdlib>

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ u64 lanemask_lt() {
  const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}
__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const unsigned lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if (lane >= (unsigned)d) v += o;
  }
  return v;
}

// exclusive scan of one value per thread over a 256-thread workgroup (4 waves); returns the
// exclusive prefix and writes the workgroup total to *total.  `tmp` is 8 words of LDS.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* tmp, uint32_t* total) {
  const unsigned lane = lane_id(), w = threadIdx.x >> 6;
  uint32_t inc = wave_incl_scan(v);
  if (lane == 63) tmp[w] = inc;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (unsigned k = 0; k < 4; k++) {
    uint32_t t = tmp[k];
    if (k < w) base += t;
  }
  *total = tmp[0] + tmp[1] + tmp[2] + tmp[3];
  __syncthreads();
  return base + inc - v;
}
// ---- batch descriptors: kernel arguments by value, blockIdx.y selects the job ------------------
struct SortJob {
  const uint32_t* kin;
  const uint32_t* vin;   // null: value = element index
  uint32_t* kout;
  uint32_t* vout;        // null: keys only (packed tile|index keys)
  const uint32_t* n_ptr; // null: n = n_cap
  uint32_t n_cap;
  uint32_t nblk;
  int32_t shift_base;    // added to the pass shift (position of the tile id inside a packed key)
  uint32_t* hist;        // [256 * nblk] digit-major + 256 totals
  uint2* ranges;         // non-null on the LAST pass of the tile sort: per-tile [begin, end) by atomic min / max
  const uint32_t* off_ptr;   // null, or a device word: the n elements start at that offset of kin / vin / kout / vout
                             // (segment 2 of the tile lists sits behind segment 1 in the same arrays; ranges are absolute)
  // depth sort of a two-round forward: one bit per element (bit i of word i / 64) that the FIRST pass -- the one that
  // creates the values -- puts into bits 30 / 31 of the value, so that the flag travels with the order (ORDER_* below)
  const unsigned long long* flag[2];
  // ABI 7 (depth sort of a view with a depth_order_hint): the job's launches run only when this word is non-zero (a depth
  // key differed from the hinted forward's); while it is zero they exit at once and adopt_order_kernel copies the order
  const int32_t* gate = nullptr;
};
__device__ __forceinline__ bool sort_job_idle(const SortJob& job) {
  return job.gate && __builtin_nontemporal_load(job.gate) == 0;
}
// A depth order's words: Gaussian index | flags.  The flags say "the rect of this Gaussian reaches a tile that is predicted
// open" for the view that owns the order (A) and for its binocular partner (B): the scan behind segment 1 reads them with
// the order itself instead of gathering GeomView::pflag at a random index per lane (measured: that gather -- 6M separate L2
// requests per iteration -- was 25 of the scan's 67 us).
constexpr uint32_t ORDER_IDX = 0x3FFFFFFFu, ORDER_A = 1u << 30, ORDER_B = 1u << 31;
// The flag words of the 64-element rows one wave ranks (row r = elements [first + 64 r, first + 64 r + 64), first % 64 == 0):
// lane r holds the words of row r -- one coalesced load per flag array and wave; a per-element lookup was two more
// (wave-uniform) vector loads per key and cost the first depth-sort pass 9 us.
struct WaveFlags {
  unsigned long long a, b;
};
__device__ __forceinline__ WaveFlags load_wave_flags(const SortJob& job, uint32_t first, int rows, uint32_t n, unsigned lane) {
  WaveFlags f{0ull, 0ull};
  if (job.flag[0] && (int)lane < rows) {
    const size_t wi = (size_t)(first >> 6) + lane;
    if (wi * 64u < n) {
      f.a = job.flag[0][wi];
      if (job.flag[1]) f.b = job.flag[1][wi];
    }
  }
  return f;
}
template <int R>
__device__ __forceinline__ uint32_t order_value(const WaveFlags& f, unsigned lane, uint32_t gi) {
  const uint32_t alo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f.a, R), ahi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(f.a >> 32), R);
  const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f.b, R), bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(f.b >> 32), R);
  const uint32_t sh = lane & 31u;
  const uint32_t wa = lane < 32u ? alo : ahi, wb = lane < 32u ? blo : bhi;
  return gi | (((wa >> sh) & 1u) * ORDER_A) | (((wb >> sh) & 1u) * ORDER_B);
}
struct SortBatch {
  int32_t n;
  SortJob j[B3GS_MAX_FUSED_VIEWS];
};

struct ScanJob {
  const uint32_t* order;    // depth order (own or the donor view's)
  const uint2* rect;        // tile rectangles by Gaussian (element i at rect[i * rect_stride])
  int32_t rect_stride;
  int32_t partner;          // >= 0: job whose rects are interleaved with this one's ([P][2]) and which shares the depth
                            //       order: this job's workgroups gather both with ONE 16-byte load; -2: done by its partner
  uint2* srect;             // the same in depth order (written by the first scan launch: ONE gather per view)
  uint32_t* soffs;          // segment 1: [ceil(K1 / 256)] sums of the 256-Gaussian sub-blocks (the emission scans inside them)
  uint32_t* chunk_sums;     // [SCAN_MAX_CHUNKS] out: instances before every chunk (scan_chunk_offsets)
  uint4* part_sum;          // [SCAN_MAX_CHUNKS] instances of every chunk, as up to four partial sums (a tile of segment 1 is
  uint4* part_vis;          //   gathered by four workgroups: ScanBatch::split); the same for the visible Gaussians
  uint32_t* header;
  uint32_t* img_header;
  int32_t* n_out;
  uint32_t* scount;         // two-round: instance count of every Gaussian behind K1 (its tiles predicted open), depth order
  OpenMap pred;             // two-round: the tiles predicted open
  int32_t* high_water;      // optional: max(N) over the forwards since the host last looked
  int32_t* overflow_flag;   // optional: set when N exceeds n_bound (the lists of this view are truncated)
  uint32_t n_bound;
  const unsigned long long* pflag;   // two-round: bit g set = Gaussian g's rect reaches a predicted-open tile (projection)
  uint32_t* flist;          // two-round: per 4096-Gaussian tile behind K1, its flagged Gaussians in depth order (a pair shares
  uint32_t* fcount;         //   the donor's list: entries carry a flag per view) and their number
  uint32_t* tsum;           // instances of every 4096-Gaussian tile of the depth order
  int32_t order_flags;      // two-round: the order words carry this view's (and its partner's) flag bits (ORDER_A / ORDER_B)
};
struct ScanBatch {
  int32_t n, P, K1, tiles_per_chunk, nchunks;   // K1 == P: one round (every Gaussian emits its whole rect)
  // split > 1 (needs tiles_per_chunk == 1): the first dense_chunks chunks -- segment 1, every Gaussian's rect gathered at a
  // random address -- are served by `split` workgroups each.  One workgroup per 4096-Gaussian tile left the gathers of
  // segment 1 on 31 of the 245 chunks of a view: those workgroups ran 42 us (the misses one CU can keep in flight) while
  // the rest of the chip waited.
  int32_t split, dense_chunks;
  uint32_t* repair_barrier;   // two-round forward: arrival counter of the repair kernel's grid barrier, reset here
  ScanJob j[B3GS_MAX_FUSED_VIEWS];
};

struct EmitJob {
  const uint32_t* order;
  const uint32_t* soffs;
  const uint2* srect;
  uint32_t* tile_out;   // tile id, or (tile << idx_bits) | index when idx_out is null
  uint32_t* idx_out;
  uint32_t n_cap;
  int32_t grid_x;
  int32_t idx_bits;
  const uint32_t* scount;   // round 2: tiles of the rect still open; round 1 of a two-round forward: tiles predicted
                            //   open, for the Gaussians behind K1 (the others emit their whole rect)
  OpenMap open;             // the bitmap that goes with scount
  const uint32_t* open_count;   // round 2: their number (0: nothing to emit)
  const uint32_t* off_ptr;      // round 2: device word N1: the instances go behind segment 1 in the same arrays
  const uint32_t* chunk_base;   // round 1: exclusive instance offset of every scan chunk (soffs holds the sub-block sums)
  const uint32_t* flist;        // round 1 of a two-round forward: the flagged Gaussians of every 4096-Gaussian tile behind K1,
  const uint32_t* fcount;       //   in depth order (ScanJob::flist; a pair shares the donor's list), their number per tile,
  const uint32_t* tsum;         //   and the instance total of every tile (a later tile of the same chunk starts behind it)
};
struct EmitBatch {
  int32_t n, P, first, subs;    // Gaussians [first, P) of the depth order; round 1: 256-Gaussian sub-blocks per scan chunk
  int32_t K1;                   // round 1: Gaussians behind K1 go to the predicted-open tiles only (K1 == P: one round)
  int32_t compact_blocks;       // round 1: workgroups [0, compact_blocks) walk the flagged list of one 4096-Gaussian tile behind
  int32_t dense_blocks;         //   K1 each (the longer ones: first in the grid), the next dense_blocks expand the 256-Gaussian
  int32_t tiles_per_chunk;      //   sub-blocks of segment 1
  EmitJob j[B3GS_MAX_FUSED_VIEWS];
};

struct RangeJob {
  const uint32_t* tile_sorted;
  const uint32_t* n_ptr;
  uint2* ranges;
  uint32_t n_cap;
  int32_t shift;        // tile id = key >> shift
  const uint32_t* off_ptr;   // as SortJob::off_ptr
};

__device__ __forceinline__ uint32_t rect_area(uint2 rc) {
  return ((rc.y & 0xFFFFu) - (rc.x & 0xFFFFu)) * ((rc.y >> 16) - (rc.x >> 16));
}
struct RangeBatch {
  int32_t n;
  RangeJob j[B3GS_MAX_FUSED_VIEWS];
};

// k-th open tile of the rect in row-major order (k < open_tiles(rc))
__device__ __forceinline__ uint32_t kth_open_tile(uint2 rc, uint32_t k, const OpenMap& om) {
  const uint32_t x0 = rc.x & 0xFFFFu, y0 = rc.x >> 16, x1 = rc.y & 0xFFFFu, y1 = rc.y >> 16;
  for (uint32_t y = y0; y < y1; y++)
    for (uint32_t wb = x0 >> 6; wb <= (x1 - 1u) >> 6; wb++) {
      uint32_t c0;
      unsigned long long m = open_bits(om, y, wb, x0, x1, &c0);
      const uint32_t c = (uint32_t)__builtin_popcountll(m);
      if (k < c) {
        for (uint32_t j = 0; j < k; j++) m &= m - 1ull;
        return y * om.grid_x + c0 + (uint32_t)__builtin_ctzll(m);
      }
      k -= c;
    }
  return y0 * om.grid_x + x0;   // not reached
}

// keys (and values) of the rows a wave ranks: wave w owns the slab [w ITEMS 64, (w + 1) ITEMS 64) of the tile, row r = 64
// consecutive elements.  A pass without input values creates them: element index | the flags of SortJob::flag.
template <bool HAS_VAL, int... R>
__device__ __forceinline__ void sort_load_rows(std::integer_sequence<int, R...>, uint32_t (&key)[sizeof...(R)],
                                               uint32_t (&val)[sizeof...(R)], const uint32_t* __restrict__ keys_in,
                                               const uint32_t* __restrict__ vals_in, bool want_val, const WaveFlags& wf,
                                               uint32_t tile_base, uint32_t tile_n, unsigned w, unsigned lane) {
  constexpr int ITEMS = (int)sizeof...(R);
  ((void)([&] {
     const uint32_t li = w * (ITEMS * 64) + R * 64 + lane;
     const bool valid = li < tile_n;
     const uint32_t gi = tile_base + li;
     key[R] = valid ? keys_in[gi] : 0xFFFFFFFFu;
     val[R] = (HAS_VAL && valid && want_val) ? (vals_in ? vals_in[gi] : order_value<R>(wf, lane, gi)) : 0u;
   }()),
   ...);
}

// ---------------------------------------------------------------------------------------------
// scan of tiles_touched in depth order: soffs[s] = sum_{s' <= s} tiles_touched[order[s']]
// two launches: per-chunk sums + per-256-Gaussian sub-block sums, scan of chunk sums (+ N, V to the headers); the
// emission kernel finishes the scan inside its own sub-block
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096
constexpr int SCAN_MAX_CHUNKS = 2048;

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
  return v;
}

// One pass gathers the rects in depth order (the random access of the binning), stores them sorted, and produces the
// chunk sums AND the sums of every 256-Gaussian sub-block (iteration r of the strided loop = sub-block r): the emission
// kernel scans inside its own sub-block, so no per-Gaussian offset array is written or read.
constexpr int SCAN_MAX_SUBS = 64;
// Sub-blocks of a chunk handled per step of scan_chunk_sums: the index loads, the flag / rect gathers that depend on them
// and the stores of that many items per thread are in flight together.  The kernel is one dependent chain of memory round
// trips per step with ~3 workgroups per CU: measured on MI355X: 4, 8 and 16 within 1 % (the rect gathers of the flagged half of the Gaussians bound the kernel).
#ifndef B3GS_SCAN_BATCH
#define B3GS_SCAN_BATCH 4
#endif
#ifndef B3GS_SCAN_SPLIT
#define B3GS_SCAN_SPLIT 4    /* workgroups per 4096-Gaussian tile of segment 1 (divides 16 / B3GS_SCAN_BATCH) */
#endif
constexpr int SCAN_SPLIT = B3GS_SCAN_SPLIT;
constexpr int SCAN_BATCH = B3GS_SCAN_BATCH;
constexpr int SCAN_PRED_WORDS = 512;   // LDS copy of a tile bitmap (4 KB): up to 512 tile rows of <= 64 tiles, 256 of <= 128, ...   // sub-blocks per chunk: 16 * tiles_per_chunk (P < 2^24 keeps tiles_per_chunk <= 2)
// Behind segment 1 (two-round forward) a Gaussian only matters when its rect reaches a predicted-open tile: the
// projection left that as one bit per Gaussian, ~5 % of them are flagged.  Letting every lane handle its own Gaussian
// made every wave run the rect gather + bitmap popcounts for two or three live lanes (measured: +29 us here and +28 us in
// the emission for that 5 %), so a 4096-Gaussian tile behind segment 1 is processed in two steps: the flags of its 16
// sub-blocks are ballot-compacted IN DEPTH ORDER into an LDS list, then the list is walked with dense lanes.  The list
// (local index | view flags) also goes to memory: the emission visits exactly these entries (EmitJob::flist).
constexpr uint32_t FL_IDX = 0xFFFu, FL_A = 1u << 12, FL_B = 1u << 13;

#ifdef B3GS_BIN_TRACE   // (tools/bin_trace.py: per-workgroup wall-clock stamps of the scan and the emission; never in the product build)
__device__ unsigned long long g_bin_trace[2][8192][8];
#define BIN_TRACE(k, wg, slot, val) do { if (threadIdx.x == 0 && (wg) < 8192u) g_bin_trace[k][wg][slot] = (val); } while (0)
#define BIN_NOW() wall_clock64()
extern "C" size_t b3gs_debug_bin_trace(unsigned long long* host) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bin_trace), sizeof(g_bin_trace));
  return sizeof(g_bin_trace) / 8;
}
#else
#define BIN_TRACE(k, wg, slot, val) do { } while (0)
#define BIN_NOW() 0ull
#endif

__global__ void __launch_bounds__(SCAN_THREADS) scan_chunk_sums(ScanBatch sb) {
  __shared__ uint32_t tmp[8];
  __shared__ uint32_t ssub[2][4][SCAN_MAX_SUBS];
  __shared__ unsigned long long s_pred[2][SCAN_PRED_WORDS];
  __shared__ unsigned long long s_mask[SCAN_ITEMS][4];   // flagged lanes of (sub-block, wave) of the tile being compacted
  __shared__ uint32_t s_base[SCAN_ITEMS][4];
  __shared__ uint32_t s_list[SCAN_TILE];
  __shared__ uint32_t s_H;
  const ScanJob& job = sb.j[blockIdx.y];
  [[maybe_unused]] const uint32_t twg = blockIdx.y * gridDim.x + blockIdx.x;
  BIN_TRACE(0, twg, 0, BIN_NOW());
  BIN_TRACE(0, twg, 5, 0ull);
  if (job.partner == -2) return;   // this view's rects are gathered by its partner's workgroups
  const bool pair = job.partner >= 0;
  const ScanJob& pj = pair ? sb.j[job.partner] : job;
  const uint32_t* __restrict__ order = job.order;
  const int subs = sb.tiles_per_chunk * SCAN_ITEMS;
  // which chunk, and which of its 256-Gaussian sub-blocks [r_lo, r_hi) (all, unless this is one of the `split` workgroups of
  // a segment-1 chunk)
  uint32_t chunk = blockIdx.x, part = 0;
  int r_lo = 0, r_hi = SCAN_ITEMS;
  if (sb.split > 1) {
    if ((int)blockIdx.x < sb.dense_chunks * sb.split) {
      chunk = blockIdx.x / (uint32_t)sb.split;
      part = blockIdx.x % (uint32_t)sb.split;
      r_lo = (int)part * (SCAN_ITEMS / sb.split);
      r_hi = r_lo + SCAN_ITEMS / sb.split;
    } else {
      chunk = (uint32_t)sb.dense_chunks + (blockIdx.x - (uint32_t)(sb.dense_chunks * sb.split));
    }
  }
  if ((int)chunk >= sb.nchunks) return;
  const int64_t begin = (int64_t)chunk * sb.tiles_per_chunk * SCAN_TILE;
  const int64_t end = min((int64_t)sb.P, begin + (int64_t)sb.tiles_per_chunk * SCAN_TILE);
  const unsigned lane = lane_id(), w = threadIdx.x >> 6;
  const u64 lt = lanemask_lt();
  uint32_t sum = 0, vis = 0, sum2 = 0, vis2 = 0;
  // the bitmaps of the tiles predicted open (304 bytes at 800x600) are staged in LDS when they fit
  OpenMap pa = job.pred, pb2 = pj.pred;
  if (end > (int64_t)sb.K1) {
    const uint32_t na = pa.grid_y * pa.row_words, nb = pb2.grid_y * pb2.row_words;
    if (na <= (uint32_t)SCAN_PRED_WORDS && nb <= (uint32_t)SCAN_PRED_WORDS) {
      for (uint32_t k = threadIdx.x; k < na; k += SCAN_THREADS) s_pred[0][k] = pa.rows[k];
      if (pair)
        for (uint32_t k = threadIdx.x; k < nb; k += SCAN_THREADS) s_pred[1][k] = pb2.rows[k];
      __syncthreads();
      pa.rows = s_pred[0];
      pb2.rows = s_pred[1];
    }
  }
  BIN_TRACE(0, twg, 1, BIN_NOW());
  const uint2* __restrict__ rect1 = job.rect;
  const uint4* __restrict__ rect2 = reinterpret_cast<const uint4*>(job.rect);   // pair: [P][2] rects, one 16-byte gather
  const size_t stride = (size_t)job.rect_stride;
  uint2* __restrict__ sa = job.srect;
  uint2* __restrict__ sb2 = pj.srect;
  for (int t = 0; t < sb.tiles_per_chunk; t++) {
    const int64_t tb = begin + (int64_t)t * SCAN_TILE;
    if (tb >= end) break;
    const int64_t te = min(end, tb + (int64_t)SCAN_TILE);
    const size_t tile = (size_t)chunk * sb.tiles_per_chunk + t;
    uint32_t ts_a = 0, ts_b = 0;   // this thread's share of the tile's instance totals
    if (tb < (int64_t)sb.K1) {
      // ---- segment 1 (K1 is a multiple of the tile size, or P): every Gaussian emits its whole rect.
      // tiles_touched == area of the rectangle (preprocess keeps them consistent); sub-blocks in batches: the index
      // loads, then the dependent rect gathers (the random access of the binning), are in flight together
      for (int r0 = r_lo; r0 < r_hi; r0 += SCAN_BATCH) {
        uint32_t oi[SCAN_BATCH];
        uint4 rr[SCAN_BATCH];
#pragma unroll
        for (int k = 0; k < SCAN_BATCH; k++) {
          const int64_t i = tb + (int64_t)(r0 + k) * SCAN_THREADS + threadIdx.x;
          oi[k] = i < te ? (order[i] & ORDER_IDX) : 0u;
        }
#pragma unroll
        for (int k = 0; k < SCAN_BATCH; k++) {
          const int64_t i = tb + (int64_t)(r0 + k) * SCAN_THREADS + threadIdx.x;
          rr[k] = make_uint4(0u, 0u, 0u, 0u);
          if (i < te) {
            if (pair) rr[k] = rect2[oi[k]];
            else { const uint2 q = rect1[oi[k] * stride]; rr[k].x = q.x; rr[k].y = q.y; }
          }
        }
#pragma unroll
        for (int k = 0; k < SCAN_BATCH; k++) {
          const int r = r0 + k;
          const int64_t i = tb + (int64_t)r * SCAN_THREADS + threadIdx.x;
          uint32_t ta = 0, tb_ = 0;
          if (i < te) {
            const uint2 ra = make_uint2(rr[k].x, rr[k].y), rb = make_uint2(rr[k].z, rr[k].w);
            sa[i] = ra;
            ta = rect_area(ra);
            if (pair) { sb2[i] = rb; tb_ = rect_area(rb); }
          }
          ts_a += ta; vis += (ta != 0);
          ts_b += tb_; vis2 += (tb_ != 0);
          const uint32_t wa = wave_sum(ta), wb = wave_sum(tb_);
          if (lane == 0) { ssub[0][w][t * SCAN_ITEMS + r] = wa; ssub[1][w][t * SCAN_ITEMS + r] = wb; }
        }
      }
    } else {
      // ---- behind segment 1: compact the flagged Gaussians of the tile (in depth order), then walk them densely
      uint32_t fl[SCAN_ITEMS];
#pragma unroll
      for (int r = 0; r < SCAN_ITEMS; r++) {
        const int64_t i = tb + (int64_t)r * SCAN_THREADS + threadIdx.x;
        uint32_t f = 0u;
        if (i < te) {
          const uint32_t o = order[i];
          if (job.order_flags) {   // the flags came with the order (own depth sort)
            f = (o & ORDER_A) ? FL_A : 0u;
            if (pair && (o & ORDER_B)) f |= FL_B;
          } else {                 // a borrowed order carries the donor's flags: look this view's up
            const uint32_t g = o & ORDER_IDX;
            f = (uint32_t)((job.pflag[g >> 6] >> (g & 63u)) & 1ull) * FL_A;
            if (pair) f |= (uint32_t)((pj.pflag[g >> 6] >> (g & 63u)) & 1ull) * FL_B;
          }
        }
        fl[r] = f;
        const u64 m = __ballot(f != 0u);
        if (lane == 0) { s_mask[r][w] = m; ssub[0][w][t * SCAN_ITEMS + r] = 0u; ssub[1][w][t * SCAN_ITEMS + r] = 0u; }
      }
      __syncthreads();
      BIN_TRACE(0, twg, 2, BIN_NOW());
      if (threadIdx.x < 64u) {   // 16 x 4 cells in depth order (sub-block major, then wave): exclusive prefix of their counts
        const uint32_t c = (uint32_t)__popcll(s_mask[threadIdx.x >> 2][threadIdx.x & 3u]);
        const uint32_t inc = wave_incl_scan(c);
        s_base[threadIdx.x >> 2][threadIdx.x & 3u] = inc - c;
        if (threadIdx.x == 63u) s_H = inc;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < SCAN_ITEMS; r++)
        if (fl[r]) s_list[s_base[r][w] + (uint32_t)__popcll(s_mask[r][w] & lt)] = (uint32_t)(r * SCAN_THREADS + threadIdx.x) | fl[r];
      __syncthreads();
      const uint32_t Hn = s_H;
      BIN_TRACE(0, twg, 3, BIN_NOW());
      BIN_TRACE(0, twg, 7, (unsigned long long)Hn);
      uint32_t* __restrict__ flist = job.flist + tile * SCAN_TILE;
      for (uint32_t e = threadIdx.x; e < Hn; e += SCAN_THREADS) {
        const uint32_t ent = s_list[e];
        const int64_t i = tb + (int64_t)(ent & FL_IDX);
        const uint32_t g = order[i] & ORDER_IDX;
        uint2 ra, rb = make_uint2(0u, 0u);
        if (pair) { const uint4 q = rect2[g]; ra = make_uint2(q.x, q.y); rb = make_uint2(q.z, q.w); }
        else ra = rect1[g * stride];
        const uint32_t ta = (ent & FL_A) ? open_tiles(ra, pa) : 0u;
        const uint32_t tb_ = (pair && (ent & FL_B)) ? open_tiles(rb, pb2) : 0u;
        sa[i] = ra;
        job.scount[i] = ta;
        if (pair) { sb2[i] = rb; pj.scount[i] = tb_; }
        flist[e] = ent;
        ts_a += ta; vis += (ta != 0);
        ts_b += tb_; vis2 += (tb_ != 0);
      }
      if (threadIdx.x == 0) job.fcount[tile] = Hn;
      __syncthreads();   // (s_mask / s_list are reused by the next tile of the chunk)
      BIN_TRACE(0, twg, 4, BIN_NOW());
    }
    // the tile's instance totals (the emission of a later tile of the same chunk starts behind them)
    uint32_t tot_a, tot_b;
    block_excl_scan_256(ts_a, tmp, &tot_a);
    block_excl_scan_256(ts_b, tmp, &tot_b);
    if (threadIdx.x == 0 && r_hi - r_lo == SCAN_ITEMS) {   // (split tiles: tiles_per_chunk == 1, nobody reads their tsum)
      job.tsum[tile] = tot_a;
      if (pair) pj.tsum[tile] = tot_b;
    }
    sum += ts_a;
    sum2 += ts_b;
  }
  uint32_t tot, tot2;
  block_excl_scan_256(sum, tmp, &tot);      // (its barriers also publish ssub)
  block_excl_scan_256(vis, tmp, &tot2);
  // this workgroup's share of the chunk total; a workgroup that serves the whole chunk writes the other shares as zero
  const bool whole = r_hi - r_lo == SCAN_ITEMS;
  const bool my_subs = (int)threadIdx.x < subs && (whole || ((int)threadIdx.x >= r_lo && (int)threadIdx.x < r_hi));
  if (threadIdx.x == 0) {
    if (whole) {
      job.part_sum[chunk] = make_uint4(tot, 0u, 0u, 0u);
      job.part_vis[chunk] = make_uint4(tot2, 0u, 0u, 0u);
    } else {
      reinterpret_cast<uint32_t*>(job.part_sum + chunk)[part] = tot;
      reinterpret_cast<uint32_t*>(job.part_vis + chunk)[part] = tot2;
    }
  }
  if (my_subs)
    job.soffs[(size_t)chunk * subs + threadIdx.x] =
        (ssub[0][0][threadIdx.x] + ssub[0][1][threadIdx.x]) + (ssub[0][2][threadIdx.x] + ssub[0][3][threadIdx.x]);
  if (pair) {
    block_excl_scan_256(sum2, tmp, &tot);
    block_excl_scan_256(vis2, tmp, &tot2);
    if (threadIdx.x == 0) {
      if (whole) {
        pj.part_sum[chunk] = make_uint4(tot, 0u, 0u, 0u);
        pj.part_vis[chunk] = make_uint4(tot2, 0u, 0u, 0u);
      } else {
        reinterpret_cast<uint32_t*>(pj.part_sum + chunk)[part] = tot;
        reinterpret_cast<uint32_t*>(pj.part_vis + chunk)[part] = tot2;
      }
    }
    if (my_subs)
      pj.soffs[(size_t)chunk * subs + threadIdx.x] =
          (ssub[1][0][threadIdx.x] + ssub[1][1][threadIdx.x]) + (ssub[1][2][threadIdx.x] + ssub[1][3][threadIdx.x]);
  }
  BIN_TRACE(0, twg, 5, BIN_NOW());
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_chunk_offsets(ScanBatch sb) {
  __shared__ uint32_t tmp[8];
  const ScanJob& job = sb.j[blockIdx.x];
  const int nchunks = sb.nchunks;
  if (blockIdx.x == 0 && threadIdx.x == 0 && sb.repair_barrier) {
    *sb.repair_barrier = 0u;
    sb.repair_barrier[3] += 1u;   // two-round forwards into this image buffer (header word 11; word 10 counts the repaired ones)
  }
  // nchunks <= 2048: 8 per thread, sequential
  uint32_t loc[8], s = 0, v = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int idx = threadIdx.x * 8 + k;
    uint4 ps = make_uint4(0u, 0u, 0u, 0u), pv = ps;
    if (idx < nchunks) { ps = job.part_sum[idx]; pv = job.part_vis[idx]; }
    loc[k] = (ps.x + ps.y) + (ps.z + ps.w);
    s += loc[k];
    v += (pv.x + pv.y) + (pv.z + pv.w);
  }
  uint32_t tot, totv;
  uint32_t base = block_excl_scan_256(s, tmp, &tot);
  block_excl_scan_256(v, tmp, &totv);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int idx = threadIdx.x * 8 + k;
    if (idx < nchunks) job.chunk_sums[idx] = base;
    base += loc[k];
  }
  if (threadIdx.x == 0) {
    job.header[0] = tot;   // N (segment 1)
    job.header[1] = totv;  // V
    job.header[2] = 0u;    // N2: set by the second binning round, if one runs
    job.header[B3GS_GEOM_EPOCH] = job.header[B3GS_GEOM_EPOCH] % 255u + 1u;   // this forward's staged-mark value (1..255)
    if (job.img_header) { job.img_header[0] = tot; job.img_header[1] = totv; job.img_header[2] = 0u; job.img_header[3] = 0u; }
    if (job.n_out) *job.n_out = (int32_t)tot;
    if (job.high_water) atomicMax(job.high_water, (int32_t)min(tot, 0x7FFFFFFFu));
    if (job.overflow_flag && tot > job.n_bound) atomicOr(job.overflow_flag, 1);
  }
}

// ---------------------------------------------------------------------------------------------
// stable LSD radix pass on (u32 key, u32 value) pairs, 8-bit digit at `shift`
//   hist:    per-workgroup digit counts, stored digit-major  hist[d * nblk + blk]
//   rowscan: one workgroup per digit: exclusive prefix over workgroups, digit total -> totals[d]
//   scatter: wave-striped stable ranking (ballot match), LDS reorder, coalesced run writes
// n is read from *n_ptr and clamped to n_cap.
// (A single-launch chained-scan pass with decoupled look-back was measured on MI355X and is SLOWER
//  here -- sort stage 352 us vs 274 us per view: a dependent kernel boundary costs ~1.5 us, an
//  agent-scope hand-off 1-2 us PER look-back hop -- so the pass stays three launches.)
// ---------------------------------------------------------------------------------------------
// Workgroup g of a launch runs on XCD g % 8 (the grids of the sort kernels are padded to a multiple of 8 in x), and every XCD
// has an L2 of its own.  A radix tile leaves a run of 8-16 keys (32-64 bytes, unaligned) per digit and the runs of
// CONSECUTIVE tiles are neighbours in the output; the histogram kernel leaves 16 bytes per digit row and four consecutive
// workgroups complete a line.  With tile = workgroup id the neighbours land in eight different L2s and leave as partial
// lines; with a contiguous band of tiles per XCD they meet in one L2 and leave as whole lines (round 5, same-box A/B at
// the headline: radix9_scatter 27.4 -> 21.8 us; -DB3GS_SORT_NO_XCD_BANDS restores tile = workgroup id).
__device__ __forceinline__ bool xcd_band_tile(uint32_t g, uint32_t used, uint32_t* tile) {
#ifndef B3GS_SORT_NO_XCD_BANDS
  const uint32_t per = (used + 7u) >> 3;
  *tile = (g & 7u) * per + (g >> 3);
  return (g >> 3) < per && *tile < used;
#else
  *tile = g;
  return g < used;
#endif
}
#ifndef B3GS_SORT_NO_XCD_BANDS
#define B3GS_SORT_GRID_X(n) ((((n) + 7u) & ~7u))
#else
#define B3GS_SORT_GRID_X(n) (n)
#endif

// One workgroup of 1024 threads counts FOUR consecutive radix tiles side by side (256 threads each, same latency as one
// tile per workgroup): the digit-major store is then one 16-byte word per digit row instead of four scattered 4-byte
// writes -- those partial writes, not the key reads or the LDS atomics, are half of this kernel's time (fit over the
// 256- and 2048-digit variants: 21 us + 12.8 us per million stores).  Rows are padded to a multiple of 16 words.
constexpr int HIST_TILES = 4;
__host__ __device__ inline uint32_t hist_stride(uint32_t nblk) { return (nblk + 15u) & ~15u; }

__global__ void __launch_bounds__(HIST_TILES * B3GS_SORT_THREADS) radix_hist(SortBatch sb, int pass_shift) {
  __shared__ uint32_t h[HIST_TILES][256];
  const SortJob& job = sb.j[blockIdx.y];
  const uint32_t sub = threadIdx.x >> 8, t = threadIdx.x & 255u;
  if (sort_job_idle(job)) return;
  const int shift = pass_shift + job.shift_base;
  const uint32_t off = job.off_ptr ? min(*job.off_ptr, job.n_cap) : 0u;
  const uint32_t* __restrict__ keys = job.kin + off;
  const uint32_t n = job.n_ptr ? min(*job.n_ptr, job.n_cap - off) : job.n_cap - off;
  // The grid is sized by the CAPACITY (n lives on the device); workgroups past n write nothing: row scan / scatter only
  // look at the first ceil(n / tile) columns
  const uint32_t used = min(job.nblk, (uint32_t)(((uint64_t)n + B3GS_SORT_TILE - 1) / B3GS_SORT_TILE));
  uint32_t grp;
  if (!xcd_band_tile(blockIdx.x, (used + HIST_TILES - 1) / HIST_TILES, &grp)) return;
  const uint32_t blk0 = grp * HIST_TILES;
  h[sub][t] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)(blk0 + sub) * B3GS_SORT_TILE;
#pragma unroll
  for (int k = 0; k < B3GS_SORT_ITEMS; k++) {
    const uint64_t i = base + k * B3GS_SORT_THREADS + t;
    if (i < n) atomicAdd(&h[sub][(keys[i] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  if (sub == 0)
    *reinterpret_cast<uint4*>(job.hist + (size_t)t * hist_stride(job.nblk) + blk0) = make_uint4(h[0][t], h[1][t], h[2][t], h[3][t]);
}

__device__ __forceinline__ void radix_rowscan_body(const SortBatch& sb, uint32_t bx, uint32_t by, uint32_t* tmp) {
  const SortJob& job = sb.j[by];
  if (sort_job_idle(job)) return;
  const uint32_t nblk = job.nblk;
  const uint32_t off = job.off_ptr ? min(*job.off_ptr, job.n_cap) : 0u;
  const uint32_t n = job.n_ptr ? min(*job.n_ptr, job.n_cap - off) : job.n_cap - off;
  const uint32_t used = min(nblk, (uint32_t)(((uint64_t)n + B3GS_SORT_TILE - 1) / B3GS_SORT_TILE));   // columns the histogram pass wrote
  uint32_t* row = job.hist + (size_t)bx * hist_stride(nblk);
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < used; b0 += 256) {
    uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < used ? row[i] : 0u;
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(v, tmp, &tot);
    if (i < used) row[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) job.hist[(size_t)256 * hist_stride(nblk) + bx] = carry;  // totals
}
__global__ void __launch_bounds__(256) radix_rowscan(SortBatch sb) {
  __shared__ uint32_t tmp[8];
  radix_rowscan_body(sb, blockIdx.x, blockIdx.y, tmp);
}

// HAS_VAL = false: keys only (packed tile|index words): no value staging buffer, 22 KB instead of 38 KB of LDS
// per workgroup (7 instead of 4 workgroups per CU)
// BITS: significant bits of this pass's digit (the last tile-split pass sees only the top bits of the tile id: 3 at
// 800x600) -- the ballot ranking costs one round per bit
template <bool HAS_VAL>
struct ScatterShared {
  uint32_t wave_cnt[4][256];
  uint32_t blk_start[256];  // first slot of digit d inside this workgroup's reorder buffer
  uint32_t gbase[256];      // global destination of that first slot
  uint32_t tmp[8];
  uint32_t s_key[B3GS_SORT_TILE];
  uint32_t s_val[HAS_VAL ? B3GS_SORT_TILE : 1];
};
// (bx, by) = (radix tile, job): blockIdx of the stand-alone launch, a loop index inside the repair kernel
template <bool HAS_VAL, int BITS>
__device__ __forceinline__ void radix_scatter_body(const SortBatch& sb, int pass_shift, uint32_t bx, uint32_t by,
                                                   ScatterShared<HAS_VAL>& sh) {
  uint32_t (&wave_cnt)[4][256] = sh.wave_cnt;
  uint32_t (&blk_start)[256] = sh.blk_start;
  uint32_t (&gbase)[256] = sh.gbase;
  uint32_t (&tmp)[8] = sh.tmp;
  uint32_t (&s_key)[B3GS_SORT_TILE] = sh.s_key;
  uint32_t (&s_val)[HAS_VAL ? B3GS_SORT_TILE : 1] = sh.s_val;

  const SortJob& job = sb.j[by];
  if (bx >= job.nblk || sort_job_idle(job)) return;
  const int shift = pass_shift + job.shift_base;
  const uint32_t off = job.off_ptr ? min(*job.off_ptr, job.n_cap) : 0u;
  const uint32_t* __restrict__ keys_in = job.kin + off;
  const uint32_t* __restrict__ vals_in = job.vin ? job.vin + off : nullptr;
  uint32_t* __restrict__ keys_out = job.kout + off;
  uint32_t* __restrict__ vals_out = job.vout ? job.vout + off : nullptr;
  const uint32_t nblk = job.nblk;
  const uint32_t* __restrict__ hist = job.hist;
  const uint32_t hstride = hist_stride(nblk);
  const uint32_t* __restrict__ totals = job.hist + (size_t)256 * hstride;
  const uint32_t n = job.n_ptr ? min(*job.n_ptr, job.n_cap - off) : job.n_cap - off;
  const uint32_t tile_base = bx * B3GS_SORT_TILE;
  if (tile_base >= n) return;  // uniform per workgroup
  const uint32_t tile_n = min((uint32_t)B3GS_SORT_TILE, n - tile_base);
  const unsigned lane = lane_id(), w = threadIdx.x >> 6;
  const u64 lt = lanemask_lt();

#pragma unroll
  for (int k = 0; k < 4; k++) wave_cnt[k][threadIdx.x] = 0;
  __syncthreads();

  // wave w owns the contiguous slab [w*1024, (w+1)*1024) of the tile, 16 rounds of 64
  uint32_t key[B3GS_SORT_ITEMS], val[B3GS_SORT_ITEMS], rank[B3GS_SORT_ITEMS];
  WaveFlags wf{0ull, 0ull};
  if (HAS_VAL && vals_out && !vals_in) wf = load_wave_flags(job, tile_base + w * (B3GS_SORT_ITEMS * 64), B3GS_SORT_ITEMS, n, lane);
  sort_load_rows<HAS_VAL>(std::make_integer_sequence<int, B3GS_SORT_ITEMS>{}, key, val, keys_in, vals_in, vals_out != nullptr, wf,
                          tile_base, tile_n, w, lane);
#pragma unroll
  for (int r = 0; r < B3GS_SORT_ITEMS; r++) {
    const uint32_t li = w * (B3GS_SORT_ITEMS * 64) + r * 64 + lane;
    const bool valid = li < tile_n;
    const uint32_t d = (key[r] >> shift) & 0xFF;
    u64 m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
      const bool bit = (d >> b) & 1u;
      const u64 bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    // lanes in m share digit d (and are valid); stable rank = earlier lanes of the group
    const uint32_t before = (uint32_t)__popcll(m & lt);
    const uint32_t cnt = wave_cnt[w][d];
    rank[r] = cnt + before;
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) wave_cnt[w][d] = cnt + (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // per digit (thread = digit): exclusive prefix over the 4 waves, workgroup digit start, global base
  {
    const uint32_t d = threadIdx.x;
    uint32_t c0 = wave_cnt[0][d], c1 = wave_cnt[1][d], c2 = wave_cnt[2][d], c3 = wave_cnt[3][d];
    uint32_t tot = c0 + c1 + c2 + c3, dummy;
    uint32_t start = block_excl_scan_256(tot, tmp, &dummy);
    uint32_t dig_total = totals[d], dummy2;
    uint32_t dig_base = block_excl_scan_256(dig_total, tmp, &dummy2);
    wave_cnt[0][d] = start;
    wave_cnt[1][d] = start + c0;
    wave_cnt[2][d] = start + c0 + c1;
    wave_cnt[3][d] = start + c0 + c1 + c2;
    blk_start[d] = start;
    gbase[d] = dig_base + hist[(size_t)d * hstride + bx];
  }
  __syncthreads();

#pragma unroll
  for (int r = 0; r < B3GS_SORT_ITEMS; r++) {
    const uint32_t li = w * (B3GS_SORT_ITEMS * 64) + r * 64 + lane;
    if (li < tile_n) {
      const uint32_t d = (key[r] >> shift) & 0xFF;
      const uint32_t p = wave_cnt[w][d] + rank[r];
      s_key[p] = key[r];
      if (HAS_VAL && vals_out) s_val[p] = val[r];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < B3GS_SORT_ITEMS; k++) {
    const uint32_t p = k * B3GS_SORT_THREADS + threadIdx.x;
    if (p < tile_n) {
      const uint32_t kk = s_key[p];
      const uint32_t d = (kk >> shift) & 0xFF;
      const uint32_t dst = gbase[d] + (p - blk_start[d]);
      keys_out[dst] = kk;
      if (HAS_VAL && vals_out) vals_out[dst] = s_val[p];
      if (job.ranges) {
        // Last pass: the keys of one tile are contiguous in this workgroup's reorder buffer (grouped by the
        // top digit, ordered by the lower ones from the earlier passes) and land on consecutive addresses,
        // so run boundaries give the tile's range inside this workgroup; min / max merge the workgroups
        // (a few atomics per workgroup).  Ranges start at (0xFFFFFFFF, 0) = empty (preprocess).
        const uint32_t tile = kk >> job.shift_base;
        if (p == 0 || (s_key[p - 1] >> job.shift_base) != tile) atomicMin(&job.ranges[tile].x, off + dst);
        if (p == tile_n - 1 || (s_key[p + 1] >> job.shift_base) != tile) atomicMax(&job.ranges[tile].y, off + dst + 1u);
      }
    }
  }
}
template <bool HAS_VAL, int BITS>
__global__ void __launch_bounds__(B3GS_SORT_THREADS) radix_scatter(SortBatch sb, int pass_shift) {
  __shared__ ScatterShared<HAS_VAL> sh;
  const SortJob& job = sb.j[blockIdx.y];
  if (sort_job_idle(job)) return;      // (a gated-off job's pointers are not read; the body checked this first)
  const uint32_t off = job.off_ptr ? min(*job.off_ptr, job.n_cap) : 0u;
  const uint32_t n = job.n_ptr ? min(*job.n_ptr, job.n_cap - off) : job.n_cap - off;
  uint32_t tile;
  // (tile count in 64 bits like radix_hist: n + tile - 1 must not wrap for n near 2^32)
  if (!xcd_band_tile(blockIdx.x, min(job.nblk, (uint32_t)(((uint64_t)n + B3GS_SORT_TILE - 1) / B3GS_SORT_TILE)), &tile)) return;
  radix_scatter_body<HAS_VAL, BITS>(sb, pass_shift, tile, blockIdx.y, sh);
}

// ---------------------------------------------------------------------------------------------
// Depth sort in THREE passes of 9-bit digits (B3gsForwardView::depth_key_bits = 27).  A visible Gaussian has view z >
// 0.2 (B3GS_NEAR), so its key -- the float bits of z -- is at least KEY27_BASE = bits(0.2f); keys of z up to ~13107
// lie within 2^27 of it.  The first pass maps  key -> key - KEY27_BASE  (culled keys 0xFFFFFFFF -> 2^27 - 1, the
// largest value) on the fly: order-preserving, ties keep their index order, so the resulting permutation is the one
// the four 8-bit passes over all 32 bits produce.  The assumption is CHECKED, not trusted: the projection raises bit 1
// of the caller's overflow word when a visible key falls outside the span (preprocess.hip), which drops that step on
// the device like a capacity overflow and makes the host fall back to the full 32-bit sort.
// Same structure as the 8-bit kernels above with 512 digits (thread t owns digits t and t + 256).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t KEY27_BASE = 0x3E4CCCCDu;          // float bits of B3GS_NEAR
constexpr uint32_t KEY27_SPAN = 1u << 27;
__device__ __forceinline__ uint32_t key27(uint32_t k) {
  if (k == 0xFFFFFFFFu) return KEY27_SPAN - 1u;       // culled: sinks to the end
  const uint32_t d = k > KEY27_BASE ? k - KEY27_BASE : 0u;
  return d < KEY27_SPAN - 2u ? d : KEY27_SPAN - 2u;   // (out-of-span keys are flagged by the projection)
}

// ITEMS = keys per thread of a radix tile: 16 (4096-key tiles) for batches of views; 8 (2048-key tiles: twice the workgroups,
// half the serial ranking rounds each) when ONE view is sorted alone and 245 tiles of 4096 would leave the CUs one
// workgroup each (the reference-shaped surface renders view by view)
template <int ITEMS>
__global__ void __launch_bounds__(HIST_TILES * B3GS_SORT_THREADS) radix9_hist(SortBatch sb, int shift, int xform) {
  constexpr uint32_t TILE = ITEMS * B3GS_SORT_THREADS;
  __shared__ uint32_t h[HIST_TILES][512];
  const SortJob& job = sb.j[blockIdx.y];
  const uint32_t sub = threadIdx.x >> 8, t = threadIdx.x & 255u;
  const uint32_t nblk = (job.n_cap + TILE - 1) / TILE;
  uint32_t grp;
  if (!xcd_band_tile(blockIdx.x, (nblk + HIST_TILES - 1) / HIST_TILES, &grp) || sort_job_idle(job)) return;
  const uint32_t blk0 = grp * HIST_TILES;
  const uint32_t* __restrict__ keys = job.kin;
  const uint32_t n = job.n_cap;
  if ((uint64_t)blk0 * TILE >= n) return;
  h[sub][t] = 0;
  h[sub][t + 256] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)(blk0 + sub) * TILE;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint64_t i = base + k * B3GS_SORT_THREADS + t;
    if (i < n) {
      uint32_t kk = keys[i];
      if (xform) kk = key27(kk);
      atomicAdd(&h[sub][(kk >> shift) & 0x1FF], 1u);
    }
  }
  __syncthreads();
  if (sub == 0) {
    const uint32_t hs = hist_stride(nblk);
    *reinterpret_cast<uint4*>(job.hist + (size_t)t * hs + blk0) = make_uint4(h[0][t], h[1][t], h[2][t], h[3][t]);
    *reinterpret_cast<uint4*>(job.hist + (size_t)(t + 256) * hs + blk0) =
        make_uint4(h[0][t + 256], h[1][t + 256], h[2][t + 256], h[3][t + 256]);
  }
}

template <int ITEMS>
__global__ void __launch_bounds__(256) radix9_rowscan(SortBatch sb) {
  __shared__ uint32_t tmp[8];
  constexpr uint32_t TILE = ITEMS * B3GS_SORT_THREADS;
  const SortJob& job = sb.j[blockIdx.y];
  if (sort_job_idle(job)) return;
  const uint32_t nblk = (job.n_cap + TILE - 1) / TILE;
  const uint32_t used = nblk;
  uint32_t* row = job.hist + (size_t)blockIdx.x * hist_stride(nblk);
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < used; b0 += 256) {
    uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < used ? row[i] : 0u;
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(v, tmp, &tot);
    if (i < used) row[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) job.hist[(size_t)512 * hist_stride(nblk) + blockIdx.x] = carry;  // totals
}

template <int ITEMS>
__global__ void __launch_bounds__(B3GS_SORT_THREADS) radix9_scatter(SortBatch sb, int shift, int xform) {
  constexpr uint32_t TILE = ITEMS * B3GS_SORT_THREADS;
  __shared__ uint32_t wave_cnt[4][512];
  __shared__ uint32_t blk_start[512];
  __shared__ uint32_t gbase[512];
  __shared__ uint32_t tmp[8];
  __shared__ uint32_t s_key[TILE];
  __shared__ uint32_t s_val[TILE];
  const SortJob& job = sb.j[blockIdx.y];
  const uint32_t nblk = (job.n_cap + TILE - 1) / TILE;
  uint32_t bx;
  if (!xcd_band_tile(blockIdx.x, nblk, &bx) || sort_job_idle(job)) return;
  const uint32_t* __restrict__ keys_in = job.kin;
  const uint32_t* __restrict__ vals_in = job.vin;
  uint32_t* __restrict__ keys_out = job.kout;
  uint32_t* __restrict__ vals_out = job.vout;
  const uint32_t* __restrict__ hist = job.hist;
  const uint32_t hstride = hist_stride(nblk);
  const uint32_t* __restrict__ totals = job.hist + (size_t)512 * hstride;
  const uint32_t n = job.n_cap;
  const uint32_t tile_base = bx * TILE;
  if (tile_base >= n) return;
  const uint32_t tile_n = min(TILE, n - tile_base);
  const unsigned lane = lane_id(), w = threadIdx.x >> 6;
  const u64 lt = lanemask_lt();
#pragma unroll
  for (int k = 0; k < 4; k++) { wave_cnt[k][threadIdx.x] = 0; wave_cnt[k][threadIdx.x + 256] = 0; }
  __syncthreads();
  uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
  WaveFlags wf{0ull, 0ull};
  if (!vals_in) wf = load_wave_flags(job, tile_base + w * (ITEMS * 64), ITEMS, n, lane);
  sort_load_rows<true>(std::make_integer_sequence<int, ITEMS>{}, key, val, keys_in, vals_in, true, wf, tile_base, tile_n, w, lane);
  if (xform) {
#pragma unroll
    for (int r = 0; r < ITEMS; r++)
      if (w * (ITEMS * 64) + r * 64 + lane < tile_n) key[r] = key27(key[r]);
  }
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t li = w * (ITEMS * 64) + r * 64 + lane;
    const bool valid = li < tile_n;
    const uint32_t d = (key[r] >> shift) & 0x1FF;
    u64 m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 9; b++) {
      const bool bit = (d >> b) & 1u;
      const u64 bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(m & lt);
    const uint32_t cnt = wave_cnt[w][d];
    rank[r] = cnt + before;
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) wave_cnt[w][d] = cnt + (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {   // per digit (thread t owns digits t and t + 256): prefix over the 4 waves, workgroup digit start, global base
    uint32_t c[2][4], tot[2], dt[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t d = threadIdx.x + 256u * h;
#pragma unroll
      for (int k = 0; k < 4; k++) c[h][k] = wave_cnt[k][d];
      tot[h] = (c[h][0] + c[h][1]) + (c[h][2] + c[h][3]);
      dt[h] = totals[d];
    }
    uint32_t lo_tot, hi_tot, lo_dtot, hi_dtot;
    const uint32_t s0 = block_excl_scan_256(tot[0], tmp, &lo_tot);
    const uint32_t s1 = block_excl_scan_256(tot[1], tmp, &hi_tot) + lo_tot;
    const uint32_t g0 = block_excl_scan_256(dt[0], tmp, &lo_dtot);
    const uint32_t g1 = block_excl_scan_256(dt[1], tmp, &hi_dtot) + lo_dtot;
    const uint32_t start[2] = {s0, s1}, dbase[2] = {g0, g1};
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t d = threadIdx.x + 256u * h;
      wave_cnt[0][d] = start[h];
      wave_cnt[1][d] = start[h] + c[h][0];
      wave_cnt[2][d] = start[h] + c[h][0] + c[h][1];
      wave_cnt[3][d] = start[h] + c[h][0] + c[h][1] + c[h][2];
      blk_start[d] = start[h];
      gbase[d] = dbase[h] + hist[(size_t)d * hstride + bx];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t li = w * (ITEMS * 64) + r * 64 + lane;
    if (li < tile_n) {
      const uint32_t d = (key[r] >> shift) & 0x1FF;
      const uint32_t p = wave_cnt[w][d] + rank[r];
      s_key[p] = key[r];
      s_val[p] = val[r];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    const uint32_t p = k * B3GS_SORT_THREADS + threadIdx.x;
    if (p < tile_n) {
      const uint32_t kk = s_key[p];
      const uint32_t d = (kk >> shift) & 0x1FF;
      const uint32_t dst = gbase[d] + (p - blk_start[d]);
      keys_out[dst] = kk;
      vals_out[dst] = s_val[p];
    }
  }
}

// ABI 7: a view whose depth keys all equal those of an earlier forward (B3gsForwardView::depth_order_hint) takes that
// forward's depth order instead of sorting: order words and sorted keys are copied (8 bytes read + 8 written per Gaussian,
// ~4 us at 1M); in a two-round forward the predicted-tile flags of THIS view (and of its scan partner) replace the ones the
// words carried.  Runs behind the gated sort launches; does nothing when the keys differed (the sort ran).
struct AdoptJob {
  const uint32_t* src_val;
  const uint32_t* src_key;
  uint32_t* dst_val;
  uint32_t* dst_key;
  const int32_t* word;                 // adopt while *word == 0 (null: always -- a trusted hint, no sort was launched)
  const unsigned long long* flag[2];   // null: no flags (one-round forward)
  uint32_t n;
};
struct AdoptBatch {
  int32_t n;
  AdoptJob j[B3GS_MAX_FUSED_VIEWS];
};
__global__ void __launch_bounds__(256) adopt_order_kernel(AdoptBatch ab) {
  const AdoptJob& job = ab.j[blockIdx.y];
  if (job.word && __builtin_nontemporal_load(job.word) != 0) return;
  const uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u;
  if (i >= job.n) return;
  if (i + 4u <= job.n) {
    uint4 v = *reinterpret_cast<const uint4*>(job.src_val + i);
    const uint4 k = *reinterpret_cast<const uint4*>(job.src_key + i);
    uint32_t w[4] = {v.x & ORDER_IDX, v.y & ORDER_IDX, v.z & ORDER_IDX, v.w & ORDER_IDX};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (job.flag[0] && ((job.flag[0][w[c] >> 6] >> (w[c] & 63u)) & 1ull)) w[c] |= ORDER_A;
      if (job.flag[1] && ((job.flag[1][(w[c] & ORDER_IDX) >> 6] >> (w[c] & 63u)) & 1ull)) w[c] |= ORDER_B;
    }
    *reinterpret_cast<uint4*>(job.dst_val + i) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4*>(job.dst_key + i) = k;
  } else {
    for (uint32_t e = i; e < job.n; e++) {
      uint32_t w = job.src_val[e] & ORDER_IDX;
      if (job.flag[0] && ((job.flag[0][w >> 6] >> (w & 63u)) & 1ull)) w |= ORDER_A;
      if (job.flag[1] && ((job.flag[1][(w & ORDER_IDX) >> 6] >> (w & 63u)) & 1ull)) w |= ORDER_B;
      job.dst_val[e] = w;
      job.dst_key[e] = job.src_key[e];
    }
  }
}

// one 9-bit pass over all jobs (depth keys only: every job has values, n = n_cap = P)
template <int ITEMS>
void radix9_pass_t(SortBatch& sb, int shift, bool xform, hipStream_t s) {
  uint32_t max_blk = 0;
  for (int k = 0; k < sb.n; k++) {
    const uint32_t nb = (sb.j[k].n_cap + ITEMS * B3GS_SORT_THREADS - 1) / (ITEMS * B3GS_SORT_THREADS);
    max_blk = nb > max_blk ? nb : max_blk;
  }
  if (sb.n <= 0 || max_blk == 0) return;
  hipLaunchKernelGGL(radix9_hist<ITEMS>, dim3(B3GS_SORT_GRID_X((max_blk + HIST_TILES - 1) / HIST_TILES), sb.n), dim3(HIST_TILES * B3GS_SORT_THREADS),
                     0, s, sb, shift, xform ? 1 : 0);
  hipLaunchKernelGGL(radix9_rowscan<ITEMS>, dim3(512, sb.n), dim3(256), 0, s, sb);
  hipLaunchKernelGGL(radix9_scatter<ITEMS>, dim3(B3GS_SORT_GRID_X(max_blk), sb.n), dim3(B3GS_SORT_THREADS), 0, s, sb, shift, xform ? 1 : 0);
}
void radix9_pass(SortBatch& sb, int shift, bool xform, hipStream_t s) {
  static const int force = getenv("B3GS_SORT9_ITEMS") ? atoi(getenv("B3GS_SORT9_ITEMS")) : 0;   // (A/B switch)
  const int items = force ? force : (sb.n == 1 ? 8 : 16);
  if (items == 8) radix9_pass_t<8>(sb, shift, xform, s);
  else radix9_pass_t<16>(sb, shift, xform, s);
}

// one pass over all jobs; swaps every job's in/out buffers afterwards (vin becomes non-null)
template <int BITS>
void launch_scatter(const SortBatch& sb, bool any_val, uint32_t max_blk, int shift, hipStream_t s) {
  if (any_val) hipLaunchKernelGGL((radix_scatter<true, BITS>), dim3(B3GS_SORT_GRID_X(max_blk), sb.n), dim3(B3GS_SORT_THREADS), 0, s, sb, shift);
  else hipLaunchKernelGGL((radix_scatter<false, BITS>), dim3(B3GS_SORT_GRID_X(max_blk), sb.n), dim3(B3GS_SORT_THREADS), 0, s, sb, shift);
}

// `bits`: how many bits of the digit at `shift` can be non-zero in any key of the batch (8 unless the caller knows)
void radix_pass(SortBatch& sb, int shift, hipStream_t s, int bits = 8) {
  uint32_t max_blk = 0;
  for (int k = 0; k < sb.n; k++) max_blk = sb.j[k].nblk > max_blk ? sb.j[k].nblk : max_blk;
  if (sb.n <= 0 || max_blk == 0) return;
  hipLaunchKernelGGL(radix_hist, dim3(B3GS_SORT_GRID_X((max_blk + HIST_TILES - 1) / HIST_TILES), sb.n), dim3(HIST_TILES * B3GS_SORT_THREADS), 0, s, sb, shift);
  hipLaunchKernelGGL(radix_rowscan, dim3(256, sb.n), dim3(256), 0, s, sb);
  bool any_val = false;
  for (int k = 0; k < sb.n; k++) any_val = any_val || sb.j[k].vout != nullptr;
  if (bits <= 2) launch_scatter<2>(sb, any_val, max_blk, shift, s);
  else if (bits <= 4) launch_scatter<4>(sb, any_val, max_blk, shift, s);
  else if (bits <= 6) launch_scatter<6>(sb, any_val, max_blk, shift, s);
  else launch_scatter<8>(sb, any_val, max_blk, shift, s);
}

// ---------------------------------------------------------------------------------------------
// instance emission in depth order (wave-cooperative expansion)
// ---------------------------------------------------------------------------------------------
constexpr int EMIT_PRED_WORDS = 256;   // LDS copy of the predicted-open bitmap (2 KB) when it fits
struct EmitShared {
  uint32_t s_end[4][64];
  uint32_t s_tmp[8];
  unsigned long long s_open[EMIT_PRED_WORDS];
};
template <bool ROUND2>
__device__ __forceinline__ void emit_instances_body(const EmitBatch& eb, uint32_t bx, uint32_t by, EmitShared& sh) {
  uint32_t (&s_end)[4][64] = sh.s_end;
  uint32_t (&s_tmp)[8] = sh.s_tmp;
  const EmitJob& job = eb.j[by];
  const int P = eb.P;
  const uint32_t off = (ROUND2 && job.off_ptr) ? min(*job.off_ptr, job.n_cap) : 0u;
  const uint32_t n_cap = job.n_cap - off;
  uint32_t* __restrict__ tile_out = job.tile_out + off;
  uint32_t* __restrict__ idx_out = job.idx_out ? job.idx_out + off : nullptr;
  const unsigned lane = lane_id(), w = threadIdx.x >> 6;
  if (!ROUND2 && (int)bx < eb.compact_blocks) {
    // One 4096-Gaussian tile behind segment 1 of a two-round forward: only its flagged Gaussians can emit anything (into
    // the tiles predicted open).  The scan left them as a list in depth order: dense lanes walk it 256 at a time; every
    // lane writes the few instances of its own Gaussian (no cooperative expansion, no search).
    const uint32_t tile = (uint32_t)(eb.K1 / SCAN_TILE) + bx;
    const uint32_t Hn = job.fcount[tile];
    if (Hn == 0u) return;
    const uint32_t chunk = tile / (uint32_t)eb.tiles_per_chunk;
    uint32_t carry = job.chunk_base[chunk];
    for (uint32_t k = chunk * (uint32_t)eb.tiles_per_chunk; k < tile; k++) carry += job.tsum[k];
    const uint32_t* __restrict__ flist = job.flist + (size_t)tile * SCAN_TILE;
    // the bitmap of the predicted tiles from LDS: a lane reads one word per row of its rect, one after the other
    OpenMap om = job.open;
    const uint32_t nw = om.grid_y * om.row_words;
    if (nw <= (uint32_t)EMIT_PRED_WORDS) {
      for (uint32_t k = threadIdx.x; k < nw; k += 256u) sh.s_open[k] = om.rows[k];
      __syncthreads();
      om.rows = sh.s_open;
    }
    // Two 256-entry rounds per trip: the list entries of both, then the dependent gathers of both, are in flight together
    // (a round is a chain of two memory round trips; most tiles hold fewer than 512 entries)
    for (uint32_t e0 = 0; e0 < Hn; e0 += 512u) {
      uint32_t sidx[2], cnt[2] = {0u, 0u}, gid[2] = {0u, 0u};
      uint2 rc[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
      bool in[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t e = e0 + 256u * h + threadIdx.x;
        in[h] = e < Hn;
        sidx[h] = in[h] ? tile * (uint32_t)SCAN_TILE + (flist[e] & FL_IDX) : 0u;
      }
#pragma unroll
      for (int h = 0; h < 2; h++)
        if (in[h]) {
          cnt[h] = job.scount[sidx[h]];       // 0 when the Gaussian is flagged for the other view of the pair only
          gid[h] = job.order[sidx[h]] & ORDER_IDX;
          rc[h] = job.srect[sidx[h]];
        }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if (e0 + 256u * h >= Hn) break;
        uint32_t tot;
        uint32_t pos = carry + block_excl_scan_256(cnt[h], s_tmp, &tot);
        carry += tot;
        if (cnt[h] != 0u) {
          const uint32_t x0 = rc[h].x & 0xFFFFu, y0 = rc[h].x >> 16, x1 = rc[h].y & 0xFFFFu, y1 = rc[h].y >> 16;
          for (uint32_t y = y0; y < y1; y++)
            for (uint32_t wb = x0 >> 6; wb <= (x1 - 1u) >> 6; wb++) {
              uint32_t c0;
              unsigned long long m = open_bits(om, y, wb, x0, x1, &c0);
              while (m) {
                const uint32_t t = y * om.grid_x + c0 + (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                if (pos < n_cap) {
                  if (idx_out) { tile_out[pos] = t; idx_out[pos] = gid[h]; }
                  else tile_out[pos] = (t << job.idx_bits) | gid[h];
                }
                pos++;
              }
            }
        }
      }
    }
    return;
  }
  if (!ROUND2) bx -= (uint32_t)eb.compact_blocks;   // the (longer) list workgroups come first in the grid
  const int s = eb.first + (int)(bx * 256 + threadIdx.x);
  uint32_t gid = 0, cnt = 0, end = 0;
  uint2 rc = make_uint2(0, 0);
  if (ROUND2) {
    // almost every Gaussian of the second round lies in finished tiles only: look at the counts first (4 bytes each)
    if (*job.open_count == 0u) return;
    if (s < P) cnt = job.scount[s];
    if (__ballot(cnt != 0) == 0) return;
  }
  const uint32_t own_sum = ROUND2 ? 1u : job.soffs[bx];   // (looked at behind the loads below: one round trip, not two)
  if (s < P) {
    if (ROUND2) {
      gid = job.order[s] & ORDER_IDX;
      rc = job.srect[s];
      end = job.soffs[s];
    } else {   // (a dense block of round 1 lies inside segment 1: K1 is a multiple of the 4096-Gaussian tile, or P)
      gid = job.order[s] & ORDER_IDX;
      rc = job.srect[s];
      cnt = rect_area(rc);
    }
  }
  if (!ROUND2) {
    // (lane k < sub reads the sum of sub-block k: one load per wave instead of a serial loop of up to 15 dependent ones)
    const uint32_t chunk = bx / (uint32_t)eb.subs, sub = bx % (uint32_t)eb.subs;
    const uint32_t before = (lane < sub && lane < (uint32_t)eb.subs) ? job.soffs[(size_t)chunk * eb.subs + lane] : 0u;
    uint32_t base = job.chunk_base[chunk];
    if (own_sum == 0u) return;   // nothing in this sub-block (culled tail, finished tiles only)
    base += wave_sum(before);
    // inclusive end of every Gaussian's instance run: offset of this 256-Gaussian sub-block (chunk base + the sums of
    // the sub-blocks before it: uniform scalar loads) + the scan inside the sub-block
    if (sub > 64u)   // (more than 64 sub-blocks per chunk: P > 2^24, not reachable through the scan's own limit)
      for (uint32_t k = 64u; k < sub; k++) base += job.soffs[(size_t)chunk * eb.subs + k];
    uint32_t tot;
    end = base + block_excl_scan_256(cnt, s_tmp, &tot) + cnt;
  }
  // lanes past P inherit the last valid end so the search array stays monotone
  const uint32_t wave_end = __shfl(end, 63 - (int)__builtin_clzll(__ballot(s < P) | 1ull), 64);
  if (s >= P) end = wave_end;
  s_end[w][lane] = end;
  const uint32_t wave_begin = __shfl(end - cnt, 0, 64);
  __builtin_amdgcn_wave_barrier();
  const bool wave_active = __ballot(cnt != 0) != 0;  // waves of culled Gaussians emit nothing

  const uint32_t x0 = rc.x & 0xFFFFu, y0 = rc.x >> 16, x1 = rc.y & 0xFFFFu;
  const uint32_t rw = x1 - x0;
  for (uint32_t j0 = wave_begin; wave_active && j0 < wave_end; j0 += 64) {
    const uint32_t j = j0 + lane;
    // smallest src with s_end[src] > j
    uint32_t lo = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      if (s_end[w][lo + step - 1] <= j) lo += step;
    }
    lo = min(lo, 63u);
    const uint32_t src_end = __shfl(end, (int)lo, 64);
    const uint32_t src_cnt = __shfl(cnt, (int)lo, 64);
    const uint32_t src_gid = __shfl(gid, (int)lo, 64);
    const uint32_t src_x0 = __shfl(x0, (int)lo, 64), src_y0 = __shfl(y0, (int)lo, 64), src_rw = __shfl(rw, (int)lo, 64);
    // round 2: the source's instances are the still-open tiles of its rect
    const bool masked = ROUND2;
    uint2 src_rc = make_uint2(0u, 0u);
    if (ROUND2) {
      src_rc.x = __shfl(rc.x, (int)lo, 64);
      src_rc.y = __shfl(rc.y, (int)lo, 64);
    }
    if (j < wave_end && j < n_cap) {
      const uint32_t k = j - (src_end - src_cnt);
      uint32_t tile;
      if (masked) {
        tile = kth_open_tile(src_rc, k, job.open);
      } else {
        const uint32_t ry = k / src_rw, rx = k - ry * src_rw;
        tile = (src_y0 + ry) * (uint32_t)job.grid_x + (src_x0 + rx);
      }
      if (idx_out) {
        tile_out[j] = tile;
        idx_out[j] = src_gid;
      } else {
        tile_out[j] = (tile << job.idx_bits) | src_gid;
      }
    }
  }
}
template <bool ROUND2>
__global__ void __launch_bounds__(256) emit_instances(EmitBatch eb) {
  __shared__ EmitShared sh;
  [[maybe_unused]] const uint32_t twg = blockIdx.y * gridDim.x + blockIdx.x;
  uint32_t bx = blockIdx.x, by = blockIdx.y;
  if (!ROUND2) {
    // round 1 is a 1-D grid: the list workgroups of ALL views first (they run longest and would otherwise start behind
    // every other view's workgroups), then the sub-blocks of segment 1; the view is the fast index in both parts
    const uint32_t nc = (uint32_t)eb.compact_blocks * (uint32_t)eb.n;
    if (blockIdx.x < nc) { bx = blockIdx.x / (uint32_t)eb.n; by = blockIdx.x % (uint32_t)eb.n; }
    else { const uint32_t d = blockIdx.x - nc; bx = (uint32_t)eb.compact_blocks + d / (uint32_t)eb.n; by = d % (uint32_t)eb.n; }
    BIN_TRACE(1, twg, 0, BIN_NOW());
    BIN_TRACE(1, twg, 7, (unsigned long long)((int)bx < eb.compact_blocks));
  }
  emit_instances_body<ROUND2>(eb, bx, by, sh);
  if (!ROUND2) BIN_TRACE(1, twg, 5, BIN_NOW());
}

// ---------------------------------------------------------------------------------------------
// round 2 of the two-round binning: count / scan of the Gaussians [K1, P) of the depth order, restricted to the
// tiles that segment 1 did not finish
// ---------------------------------------------------------------------------------------------
struct Scan2Job {
  const uint32_t* order;    // depth order and the rects by Gaussian: the first scan stored the rects of the Gaussians behind
  const uint2* rect;        //   segment 1 only where they reach a predicted tile, so this (rare) round gathers them itself
  int32_t rect_stride;
  uint2* srect;             // rects in depth order: completed here for [K1, P) (the emission reads them)
  uint32_t* scount;
  uint32_t* soffs;
  uint32_t* chunk_sums;     // [SCAN_MAX_CHUNKS]
  uint32_t* header;         // header[0] = N1 (read), header[2] = N2 (written)
  uint32_t* img_header;
  int32_t* n_out;           // *n_out = N1 + N2
  OpenMap open;
  int32_t* high_water;      // as ScanJob's, for N1 + N2
  int32_t* overflow_flag;
  uint32_t n_bound;
};
struct Scan2Batch {
  int32_t n, P, K1, tiles_per_chunk, nchunks;
  Scan2Job j[B3GS_MAX_FUSED_VIEWS];
};

__device__ __forceinline__ void scan2_chunk_sums_body(const Scan2Batch& sb, uint32_t bx, uint32_t by, uint32_t* tmp) {
  const Scan2Job& job = sb.j[by];
  if (job.img_header[3] == 0u) {   // no tile to repair: segment 2 is empty, the chunk sum is all that is needed
    if (threadIdx.x == 0) job.chunk_sums[bx] = 0u;
    return;
  }
  const int64_t begin = (int64_t)sb.K1 + (int64_t)bx * sb.tiles_per_chunk * SCAN_TILE;
  const int64_t end = min((int64_t)sb.P, begin + (int64_t)sb.tiles_per_chunk * SCAN_TILE);
  uint32_t sum = 0;
  const uint32_t* __restrict__ order = job.order;
  const uint2* __restrict__ rect = job.rect;
  const size_t stride = (size_t)job.rect_stride;
  uint2* __restrict__ srect = job.srect;
  uint32_t* __restrict__ scount = job.scount;
  for (int64_t i0 = begin + threadIdx.x; i0 < end; i0 += 4 * SCAN_THREADS) {   // four rect gathers in flight per thread
    uint32_t oi[4];
    uint2 rc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t i = i0 + (int64_t)k * SCAN_THREADS;
      oi[k] = i < end ? (order[i] & ORDER_IDX) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t i = i0 + (int64_t)k * SCAN_THREADS;
      rc[k] = i < end ? rect[oi[k] * stride] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t i = i0 + (int64_t)k * SCAN_THREADS;
      if (i < end) {
        const uint32_t t = open_tiles(rc[k], job.open);
        srect[i] = rc[k];
        scount[i] = t;
        sum += t;
      }
    }
  }
  uint32_t tot;
  block_excl_scan_256(sum, tmp, &tot);
  if (threadIdx.x == 0) job.chunk_sums[bx] = tot;
}

__device__ __forceinline__ void scan2_chunk_offsets_body(const Scan2Batch& sb, uint32_t bx, uint32_t* tmp) {
  const Scan2Job& job = sb.j[bx];
  const int nchunks = sb.nchunks;
  uint32_t loc[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int idx = threadIdx.x * 8 + k;
    loc[k] = idx < nchunks ? job.chunk_sums[idx] : 0u;
    s += loc[k];
  }
  uint32_t tot;
  uint32_t base = block_excl_scan_256(s, tmp, &tot);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int idx = threadIdx.x * 8 + k;
    if (idx < nchunks) job.chunk_sums[idx] = base;
    base += loc[k];
  }
  if (threadIdx.x == 0) {
    job.header[2] = tot;   // N2
    if (job.img_header) job.img_header[2] = tot;
    const uint32_t n12 = job.header[0] + tot;
    if (job.n_out) *job.n_out = (int32_t)n12;
    if (job.high_water) atomicMax(job.high_water, (int32_t)min(n12, 0x7FFFFFFFu));
    if (job.overflow_flag && n12 > job.n_bound) atomicOr(job.overflow_flag, 1);
  }
}

__device__ __forceinline__ void scan2_chunk_apply_body(const Scan2Batch& sb, uint32_t bx, uint32_t by, uint32_t* wave_tot) {
  const Scan2Job& job = sb.j[by];
  if (job.img_header[3] == 0u) return;   // no open tile: no segment 2, nobody reads the offsets
  const uint32_t* __restrict__ scount = job.scount;
  uint32_t* __restrict__ soffs = job.soffs;
  const int P = sb.P;
  const unsigned lane = lane_id(), w = threadIdx.x >> 6;
  uint32_t carry = job.chunk_sums[bx];
  const int64_t begin = (int64_t)sb.K1 + (int64_t)bx * sb.tiles_per_chunk * SCAN_TILE;
  for (int t = 0; t < sb.tiles_per_chunk; t++) {
    const int64_t tb = begin + (int64_t)t * SCAN_TILE;
    if (tb >= P) break;
    const int64_t wb = tb + (int64_t)w * (SCAN_ITEMS * 64);
    uint32_t inc[SCAN_ITEMS], run = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ITEMS; r++) {
      const int64_t i = wb + r * 64 + lane;
      const uint32_t v = i < P ? scount[i] : 0u;
      inc[r] = run + wave_incl_scan(v);
      run = __shfl(inc[r], 63, 64);
    }
    if (lane == 0) wave_tot[w] = run;
    __syncthreads();
    uint32_t base = carry, tot = 0;
#pragma unroll
    for (unsigned k = 0; k < 4; k++) {
      const uint32_t x = wave_tot[k];
      if (k < w) base += x;
      tot += x;
    }
#pragma unroll
    for (int r = 0; r < SCAN_ITEMS; r++) {
      const int64_t i = wb + r * 64 + lane;
      if (i < P) soffs[i] = base + inc[r];
    }
    carry += tot;
    __syncthreads();
  }
}

// stand-alone launches of the round-2 scans (B3GS_ROUND2_LEGACY=1, and tile grids deeper than two radix passes)
__global__ void __launch_bounds__(SCAN_THREADS) scan2_chunk_sums(Scan2Batch sb) {
  __shared__ uint32_t tmp[8];
  scan2_chunk_sums_body(sb, blockIdx.x, blockIdx.y, tmp);
}
__global__ void __launch_bounds__(SCAN_THREADS) scan2_chunk_offsets(Scan2Batch sb) {
  __shared__ uint32_t tmp[8];
  scan2_chunk_offsets_body(sb, blockIdx.x, tmp);
}
__global__ void __launch_bounds__(SCAN_THREADS) scan2_chunk_apply(Scan2Batch sb) {
  __shared__ uint32_t wave_tot[4];
  scan2_chunk_apply_body(sb, blockIdx.x, blockIdx.y, wave_tot);
}

__global__ void __launch_bounds__(256) tile_ranges(RangeBatch rb) {
  const RangeJob& job = rb.j[blockIdx.y];
  const uint32_t off = job.off_ptr ? min(*job.off_ptr, job.n_cap) : 0u;
  const uint32_t* __restrict__ tile_sorted = job.tile_sorted + off;
  const uint32_t n = min(*job.n_ptr, job.n_cap - off);
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int sh = job.shift;
  const uint32_t t = tile_sorted[j] >> sh;
  if (j == 0 || (tile_sorted[j - 1] >> sh) != t) job.ranges[t].x = off + j;
  if (j == n - 1 || (tile_sorted[j + 1] >> sh) != t) job.ranges[t].y = off + j + 1;
}

// ---------------------------------------------------------------------------------------------
// Second binning round as ONE launch ("repair kernel").  After the blend forward over segment 1 the image header says
// how many tiles are still open and were not predicted open (word 3).  In the steady state that number is zero -- the
// prediction has settled -- and the ten launches of the round (three scans, emission, two radix passes of three kernels)
// did nothing but start and exit (~60-90 us per iteration at the headline workload).  Here one persistent grid of
// REPAIR_GRID workgroups looks at the counts once: all zero -> exit (one launch boundary); otherwise it walks the same
// phases with the same device code as the stand-alone kernels, workgroups striding over each phase's blocks, and an
// agent-scope grid barrier between phases (monotonic counter; lane 0 releases before it arrives and acquires after
// the poll, MI355X_MICROARCH.md "barrier-counter").  REPAIR_GRID = one workgroup per CU: resident at once with room to
// spare (<= 40 KB of LDS, 256 threads), so the barrier cannot starve; the spin is bounded all the same and a time-out
// is reported through the status word instead of hanging the queue.
// ---------------------------------------------------------------------------------------------
constexpr int REPAIR_GRID = 256;
struct RepairArgs {
  int32_t* overflow[B3GS_MAX_FUSED_VIEWS];   // every view's sticky overflow word (may be null): bit 2 <- barrier time-out
  uint32_t spin_limit;                       // polls of one barrier before the launch gives up
  Scan2Batch sc;
  EmitBatch eb;
  SortBatch tb[2];          // tile-split passes (in / out already swapped for pass 1; ranges set on the last one)
  int32_t passes, bits[2];
  uint32_t emit_blocks, max_blk;
  uint32_t* barrier;        // [2] arrival counter (zeroed by scan_chunk_offsets of the same forward) | status (bit 0: time-out)
};

// counter[0] = arrivals of this forward's launch (reset by the forward's first scan); its top bit is raised by the
// workgroup that gives up: every workgroup still polling -- or arriving later, when the grid was not co-resident -- then
// leaves at once instead of waiting out its own limit
constexpr uint32_t BARRIER_GAVE_UP = 0x80000000u;
__device__ __forceinline__ bool grid_barrier(uint32_t* counter, uint32_t target, uint32_t spin_limit) {
  __shared__ uint32_t s_ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0, ok = 1;
    for (;;) {
      const uint32_t c = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c & BARRIER_GAVE_UP) { ok = 0; break; }
      if (c >= target) break;
      __builtin_amdgcn_s_sleep(8);
      if (++spins > spin_limit) {   // ~1 s at the default limit: a workgroup of the grid never became resident
        __hip_atomic_fetch_or(counter, BARRIER_GAVE_UP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0u;
}

// one radix tile of 4096 keys counted by 256 threads (the stand-alone radix_hist counts four per 1024-thread workgroup)
__device__ __forceinline__ void radix_hist_tile_body(const SortBatch& sb, int pass_shift, uint32_t blk, uint32_t by, uint32_t* h) {
  const SortJob& job = sb.j[by];
  if (blk >= job.nblk) return;
  const int shift = pass_shift + job.shift_base;
  const uint32_t off = job.off_ptr ? min(*job.off_ptr, job.n_cap) : 0u;
  const uint32_t* __restrict__ keys = job.kin + off;
  const uint32_t n = job.n_ptr ? min(*job.n_ptr, job.n_cap - off) : job.n_cap - off;
  if ((uint64_t)blk * B3GS_SORT_TILE >= n) return;
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blk * B3GS_SORT_TILE;
#pragma unroll
  for (int k = 0; k < B3GS_SORT_ITEMS; k++) {
    const uint64_t i = base + k * B3GS_SORT_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  job.hist[(size_t)threadIdx.x * hist_stride(job.nblk) + blk] = h[threadIdx.x];
}

template <bool HAS_VAL>
union RepairShared {
  uint32_t tmp[8];
  uint32_t hist[256];
  EmitShared emit;
  ScatterShared<HAS_VAL> scatter;
};

template <bool HAS_VAL>
__global__ void __launch_bounds__(256) repair_kernel(RepairArgs ra) {
  __shared__ RepairShared<HAS_VAL> sh;
  // anything to repair?  (word 3 of every view's image header, final since the blend forward completed)
  bool any = false;
  for (int v = 0; v < ra.sc.n; v++) any = any || ra.sc.j[v].img_header[3] != 0u;
  if (!any) {
    // segment 2 is empty: N2 = 0 was stored by the first scan, *n_out already holds N1
    return;
  }
  const uint32_t G = gridDim.x, wg = blockIdx.x;
  uint32_t phase = 0;
  bool ok = true;
  if (wg == 0 && threadIdx.x == 0) ra.barrier[2] += 1u;   // forwards whose prediction missed (the host watches the rate)
  // A time-out leaves the repaired tiles of this forward wrong: besides the status word (read by the host at its next
  // check) every view's sticky overflow word gets bit 2, so that b3gs_adam_step(skip_if_nonzero) and the densification
  // statistics drop this step ON THE DEVICE like a step rendered from truncated lists.
  if (__hip_atomic_load(ra.barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & BARRIER_GAVE_UP) return;   // a late workgroup of a grid that gave up
#define REPAIR_SYNC()                                                   \
  do {                                                                  \
    ok = ok && grid_barrier(ra.barrier, ++phase * G, ra.spin_limit);    \
    if (!ok) {                                                          \
      if (threadIdx.x == 0) {                                           \
        atomicOr(ra.barrier + 1, 1u);                                   \
        for (int v = 0; v < ra.sc.n; v++)                               \
          if (ra.overflow[v]) atomicOr(ra.overflow[v], 4);              \
      }                                                                 \
      return;                                                           \
    }                                                                   \
  } while (0)
  const uint32_t nv = (uint32_t)ra.sc.n;
  for (uint32_t b = wg; b < (uint32_t)ra.sc.nchunks * nv; b += G) {
    scan2_chunk_sums_body(ra.sc, b % (uint32_t)ra.sc.nchunks, b / (uint32_t)ra.sc.nchunks, sh.tmp);
    __syncthreads();
  }
  REPAIR_SYNC();
  for (uint32_t b = wg; b < nv; b += G) {
    scan2_chunk_offsets_body(ra.sc, b, sh.tmp);
    __syncthreads();
  }
  REPAIR_SYNC();
  for (uint32_t b = wg; b < (uint32_t)ra.sc.nchunks * nv; b += G) {
    scan2_chunk_apply_body(ra.sc, b % (uint32_t)ra.sc.nchunks, b / (uint32_t)ra.sc.nchunks, sh.tmp);
    __syncthreads();
  }
  REPAIR_SYNC();
  for (uint32_t b = wg; b < ra.emit_blocks * nv; b += G) {
    emit_instances_body<true>(ra.eb, b % ra.emit_blocks, b / ra.emit_blocks, sh.emit);
    __syncthreads();
  }
  // radix tiles that hold instances of segment 2 (N2 is on the device since the second phase; the host only knows
  // the capacity): the widest view bounds the loops below
  uint32_t used_blk = 0;
  for (uint32_t v = 0; v < nv; v++) {
    const SortJob& j = ra.tb[0].j[v];
    const uint32_t off = min(*j.off_ptr, j.n_cap);
    const uint32_t n = min(*j.n_ptr, j.n_cap - off);
    used_blk = max(used_blk, (n + (uint32_t)B3GS_SORT_TILE - 1u) / (uint32_t)B3GS_SORT_TILE);
  }
  used_blk = min(used_blk, ra.max_blk);
  for (int p = 0; p < ra.passes; p++) {
    const SortBatch& tb = ra.tb[p];
    REPAIR_SYNC();
    for (uint32_t b = wg; b < used_blk * nv; b += G) {
      radix_hist_tile_body(tb, 8 * p, b % used_blk, b / used_blk, sh.hist);
      __syncthreads();
    }
    REPAIR_SYNC();
    for (uint32_t b = wg; b < 256u * nv; b += G) {
      radix_rowscan_body(tb, b & 255u, b >> 8, sh.tmp);
      __syncthreads();
    }
    REPAIR_SYNC();
    for (uint32_t b = wg; b < used_blk * nv; b += G) {
      radix_scatter_body<HAS_VAL, 8>(tb, 8 * p, b % used_blk, b / used_blk, sh.scatter);
      __syncthreads();
    }
  }
#undef REPAIR_SYNC
}

int tile_sort_passes(int W, int H) {
  const int tbits = b3gs_tile_bits(W, H);
  return tbits == 0 ? 0 : (tbits + 7) / 8;
}

}  // namespace

// the view whose rects sit in the odd slots of view d's [P][2] rect array and which borrows d's depth order: d's scan
// workgroups gather both rects with one load and keep ONE flagged list for the pair (or -1)
static int scan_partner_of(const BinJob* jobs, int nviews, int d) {
  const uint2* rd = jobs[d].rect ? jobs[d].rect : jobs[d].g.rect;
  const int sd = jobs[d].rect ? jobs[d].rect_stride : 1;
  for (int v = 0; v < nviews; v++) {
    const uint2* rv = jobs[v].rect ? jobs[v].rect : jobs[v].g.rect;
    const int sv = jobs[v].rect ? jobs[v].rect_stride : 1;
    if (jobs[v].order_from == d && sv == 2 && sd == 2 && rv == rd + 1) return v;
  }
  return -1;
}

static const uint32_t* depth_order_of(const BinJob* jobs, int v) {
  const BinJob& jb = jobs[v];
  return jb.order_from == -1 ? jb.g.sval[0] : (jb.order_from >= 0 ? jobs[jb.order_from].g.sval[0] : jb.order);
}

// stable sort of n (u32 key, element index) pairs by all 32 key bits: keys -> skey[1] -> skey[0] -> skey[1] ->
// skey[0]; sorted keys end in skey[0], the permutation in sval[0] (also used by the k-nearest-neighbour init)
void b3gs_launch_sort_u32_index(const uint32_t* keys, uint32_t* const skey[2], uint32_t* const sval[2], uint32_t n,
                                uint32_t* hist, hipStream_t s) {
  SortBatch db;
  db.n = 1;
  db.j[0] = SortJob{keys, nullptr, skey[1], sval[1], nullptr, n, b3gs_sort_blocks((int64_t)n), 0, hist, nullptr, nullptr};
  for (int pass = 0; pass < 4; pass++) {
    radix_pass(db, 8 * pass, s);
    const int dst = pass & 1;  // destination of the NEXT pass
    db.j[0].kin = skey[dst ^ 1]; db.j[0].vin = sval[dst ^ 1];
    db.j[0].kout = skey[dst]; db.j[0].vout = sval[dst];
  }
}

void b3gs_launch_binning_batch(int32_t P, int nviews, const BinJob* jobs, hipStream_t s) {
  b3gs_launch_depth_order_batch(P, nviews, jobs, s);
  b3gs_launch_tile_lists_batch(P, nviews, jobs, s);
}

void b3gs_launch_depth_order_batch(int32_t P, int nviews, const BinJob* jobs, hipStream_t s) {
  if (nviews <= 0) return;
  if (P <= 0) {
    for (int v = 0; v < nviews; v++) {
      (void)hipMemsetAsync(jobs[v].g.header, 0, 8, s);
      if (jobs[v].im.header) (void)hipMemsetAsync(jobs[v].im.header, 0, 8, s);
      if (jobs[v].n_out) (void)hipMemsetAsync(jobs[v].n_out, 0, 4, s);
    }
    return;
  }
  // ---- 1. depth order for every view that does not borrow another view's order; result in sval[0] / skey[0]:
  //         4 passes of 8 bits  depth_key -> [1] -> [0] -> [1] -> [0], or -- when every job of the batch vouches for
  //         the 27-bit key span (BinJob::key_bits == 27) -- 3 passes of 9 bits  depth_key -> [0] -> [1] -> [0]
  SortBatch db;
  db.n = 0;
  AdoptBatch adopt;
  adopt.n = 0;
  int sorted_view[B3GS_MAX_FUSED_VIEWS];
  const uint32_t pblk = b3gs_sort_blocks((int64_t)P);
  bool span27 = true;
  for (int v = 0; v < nviews; v++) span27 = span27 && jobs[v].key_bits == 27;
  const int npass = span27 ? 3 : 4;
  const int first_dst = span27 ? 0 : 1;
  const int K1 = b3gs_seg1_count(jobs[0], P);
  for (int v = 0; v < nviews; v++) {
    if (jobs[v].order_from != -1) continue;
    const GeomView& g = jobs[v].g;
    const unsigned long long* flag[2] = {nullptr, nullptr};
    if (K1 < P) {   // two rounds: the predicted-tile flags of this view and of its scan partner travel with the order
      const int pv = scan_partner_of(jobs, nviews, v);
      flag[0] = g.pflag;
      flag[1] = pv >= 0 ? jobs[pv].g.pflag : nullptr;
    }
    const bool hinted = jobs[v].hint_sval && jobs[v].hint_word;
    if (hinted)     // adopt an earlier forward's order while the keys are equal (trusted: always, the sort is not launched)
      adopt.j[adopt.n++] = AdoptJob{jobs[v].hint_sval, jobs[v].hint_skey, g.sval[0], g.skey[0],
                                    jobs[v].hint_trusted ? nullptr : jobs[v].hint_word, {flag[0], flag[1]}, (uint32_t)P};
    if (hinted && jobs[v].hint_trusted) continue;
    sorted_view[db.n] = v;
    SortJob& sj = db.j[db.n++];
    sj = SortJob{g.depth_key, nullptr, g.skey[first_dst], g.sval[first_dst], nullptr, (uint32_t)P, pblk, 0, g.hist,
                 nullptr, nullptr, {flag[0], flag[1]}};
    if (hinted) sj.gate = jobs[v].hint_word;
  }
  for (int pass = 0; pass < npass; pass++) {
    if (span27) radix9_pass(db, 9 * pass, pass == 0, s);
    else radix_pass(db, 8 * pass, s);
    for (int k = 0; k < db.n; k++) {
      const GeomView& g = jobs[sorted_view[k]].g;
      const int src = (first_dst + pass) & 1;   // where this pass wrote: the next one reads it and writes the other
      SortJob& j = db.j[k];
      j.kin = g.skey[src];
      j.vin = g.sval[src];
      j.kout = g.skey[src ^ 1];
      j.vout = g.sval[src ^ 1];
    }
  }
  if (adopt.n > 0)
    hipLaunchKernelGGL(adopt_order_kernel, dim3((P + 1023) / 1024, adopt.n), dim3(256), 0, s, adopt);

  // ---- 2. scan of tiles_touched in depth order -> soffs, N, V (segment 1 = the first K1 Gaussians of the order)
  const int total_tiles = (P + SCAN_TILE - 1) / SCAN_TILE;
  ScanBatch sc;
  sc.n = nviews;
  sc.P = P;
  sc.K1 = K1;
  sc.tiles_per_chunk = (total_tiles + SCAN_MAX_CHUNKS - 1) / SCAN_MAX_CHUNKS;
  sc.nchunks = (total_tiles + sc.tiles_per_chunk - 1) / sc.tiles_per_chunk;
  sc.repair_barrier = (K1 < P && jobs[0].im.header) ? jobs[0].im.header + B3GS_HDR_REPAIR_BARRIER : nullptr;
  if (sc.tiles_per_chunk * SCAN_ITEMS > SCAN_MAX_SUBS) {   // LDS sub-block sums of scan_chunk_sums (P < 2^24 keeps it at 32)
    (void)b3gs_fail(B3GS_ERR_ARG, "b3gs_launch_depth_order_batch", "too many Gaussians for the scan's sub-block table");
    return;
  }
  for (int v = 0; v < nviews; v++) {
    const BinJob& jb = jobs[v];
    const uint2* rect = jb.rect ? jb.rect : jb.g.rect;
    const int32_t rstride = jb.rect ? jb.rect_stride : 1;
    sc.j[v] = ScanJob{depth_order_of(jobs, v), rect, rstride, -1, jb.g.srect, jb.g.soffs, jb.g.scan_tmp,
                      reinterpret_cast<uint4*>(jb.g.scan_tmp + SCAN_MAX_CHUNKS), reinterpret_cast<uint4*>(jb.g.scan_tmp + 5 * SCAN_MAX_CHUNKS),
                      jb.g.header, jb.im.header, jb.n_out, jb.g.scount,
                      open_map(jb.im.pred_rows, jb.W, jb.H), jb.high_water, jb.overflow_flag,
                      (uint32_t)(jb.n_bound > 0 ? (jb.n_bound < 0xFFFFFFFFll ? jb.n_bound : 0xFFFFFFFFll) : 0), jb.g.pflag,
                      jb.g.flist, jb.g.fcount, jb.g.tsum, (K1 < P && jb.order_from == -1) ? 1 : 0};
  }
  // a view that borrows view d's depth order AND whose rects sit in the odd slots of d's [P][2] array is folded
  // into d's gather
  for (int d = 0; d < nviews; d++) {
    const int v = jobs[d].order_from == -1 ? scan_partner_of(jobs, nviews, d) : -1;
    if (v < 0) continue;
    sc.j[d].partner = v;
    sc.j[v].partner = -2;
  }
  sc.split = sc.tiles_per_chunk == 1 ? SCAN_SPLIT : 1;
  sc.dense_chunks = K1 < P ? K1 / SCAN_TILE : sc.nchunks;
  const int scan_blocks = sc.split > 1 ? sc.dense_chunks * sc.split + (sc.nchunks - sc.dense_chunks) : sc.nchunks;
  hipLaunchKernelGGL(scan_chunk_sums, dim3(scan_blocks, nviews), dim3(SCAN_THREADS), 0, s, sc);
  hipLaunchKernelGGL(scan_chunk_offsets, dim3(nviews), dim3(SCAN_THREADS), 0, s, sc);
  // (no per-Gaussian offsets: the emission scans inside its 256-Gaussian sub-block, see scan_chunk_sums)
}

void b3gs_launch_tile_lists_batch(int32_t P, int nviews, const BinJob* jobs, hipStream_t s) {
  if (nviews <= 0 || P <= 0) return;
  // views of different tile-sort depth cannot share the pass loop: run them one by one
  const int passes = tile_sort_passes(jobs[0].W, jobs[0].H);
  for (int v = 1; v < nviews; v++) {
    if (tile_sort_passes(jobs[v].W, jobs[v].H) != passes) {
      for (int k = 0; k < nviews; k++) {
        BinJob one = jobs[k];
        // a donor index refers to the batch; resolve it to the donor's buffers before splitting
        if (one.order_from >= 0) { one.order = jobs[one.order_from].g.sval[0]; one.order_from = -2; }
        b3gs_launch_tile_lists_batch(P, 1, &one, s);
      }
      return;
    }
  }


  // ---- 3. emit (tile, index) instances in depth order into the buffer from which `passes` ping-pongs end in [0];
  //         every kernel clamps to min(N, capacity)
  int tbits = 0;
  for (int v = 0; v < nviews; v++) tbits = max(tbits, b3gs_tile_bits(jobs[v].W, jobs[v].H));
  const int first = passes & 1;
  const int K1 = b3gs_seg1_count(jobs[0], P);
  EmitBatch eb;
  eb.n = nviews;
  eb.P = P;
  eb.K1 = K1;
  eb.first = 0;
  {   // the chunking of b3gs_launch_depth_order_batch's scan
    const int total_tiles = (P + SCAN_TILE - 1) / SCAN_TILE;
    eb.tiles_per_chunk = (total_tiles + SCAN_MAX_CHUNKS - 1) / SCAN_MAX_CHUNKS;
    eb.subs = eb.tiles_per_chunk * SCAN_ITEMS;
  }
  // segment 1 as 256-Gaussian sub-blocks, then one workgroup per 4096-Gaussian tile behind it (flagged lists)
  eb.dense_blocks = K1 < P ? K1 / 256 : (P + 255) / 256;
  eb.compact_blocks = K1 < P ? (P - K1 + SCAN_TILE - 1) / SCAN_TILE : 0;
  const int emit_blocks = eb.compact_blocks + eb.dense_blocks;
  SortBatch tb;
  tb.n = nviews;
  RangeBatch rb;
  rb.n = nviews;
  uint32_t max_cap = 0;
  for (int v = 0; v < nviews; v++) {
    const BinJob& jb = jobs[v];
    const uint32_t n_cap = (uint32_t)(jb.n_bound > 0 ? jb.n_bound : 0);
    max_cap = n_cap > max_cap ? n_cap : max_cap;
    const int gx = (jb.W + B3GS_TILE - 1) / B3GS_TILE;
    const int idx_bits = b3gs_packed_idx_bits(P, jb.W, jb.H);
    const OpenMap pm = open_map(jb.im.pred_rows, jb.W, jb.H);
    // the flagged lists of a pair live with the view whose workgroups scanned both
    const int dn = jb.order_from;
    const GeomView& fg = (dn >= 0 && scan_partner_of(jobs, nviews, dn) == v) ? jobs[dn].g : jb.g;
    if (idx_bits >= 0) {
      // (tile << idx_bits | index) fits 32 bits: ONE word per instance through emission, both passes and the
      // blend kernels (which mask the index out) -- half the tile-sort traffic.  The words ping-pong
      // between val[first] and val[first ^ 1] and end in val[0], where the point list is expected.
      eb.j[v] = EmitJob{depth_order_of(jobs, v), jb.g.soffs, jb.g.srect, jb.b.val[first], nullptr, n_cap, gx, idx_bits, jb.g.scount, pm, nullptr, nullptr, jb.g.scan_tmp,
                        fg.flist, fg.fcount, jb.g.tsum};
      tb.j[v] = SortJob{jb.b.val[first], nullptr, jb.b.val[first ^ 1], nullptr, jb.g.header, n_cap,
                        b3gs_sort_blocks((int64_t)n_cap), idx_bits, jb.b.hist, nullptr, nullptr};
      rb.j[v] = RangeJob{jb.b.val[0], jb.g.header, jb.im.ranges, n_cap, idx_bits, nullptr};
    } else {
      eb.j[v] = EmitJob{depth_order_of(jobs, v), jb.g.soffs, jb.g.srect, jb.b.key[first], jb.b.val[first], n_cap, gx, 0, jb.g.scount, pm, nullptr, nullptr, jb.g.scan_tmp,
                        fg.flist, fg.fcount, jb.g.tsum};
      tb.j[v] = SortJob{jb.b.key[first], jb.b.val[first], jb.b.key[first ^ 1], jb.b.val[first ^ 1], jb.g.header, n_cap,
                        b3gs_sort_blocks((int64_t)n_cap), 0, jb.b.hist, nullptr, nullptr};
      rb.j[v] = RangeJob{jb.b.key[0], jb.g.header, jb.im.ranges, n_cap, 0, nullptr};
    }
  }
  if (max_cap == 0) return;  // im.ranges was reset to "empty" by the preprocess launch
  hipLaunchKernelGGL(emit_instances<false>, dim3(emit_blocks * nviews), dim3(256), 0, s, eb);

  // ---- 4. stable split by tile id
  for (int p = 0; p < passes; p++) {
    if (p == passes - 1)
      for (int v = 0; v < nviews; v++) tb.j[v].ranges = jobs[v].im.ranges;
    radix_pass(tb, 8 * p, s, min(8, tbits - 8 * p));
    for (int v = 0; v < nviews; v++) {
      SortJob& j = tb.j[v];
      const uint32_t* k = j.kin; const uint32_t* vv = j.vin;
      j.kin = j.kout; j.vin = j.vout;
      j.kout = const_cast<uint32_t*>(k); j.vout = const_cast<uint32_t*>(vv);  // stays null for packed keys
    }
  }
  // ---- 5. per-tile [begin, end): produced by the last pass above; a single-tile image has no pass
  if (passes == 0) hipLaunchKernelGGL(tile_ranges, dim3((max_cap + 255) / 256, nviews), dim3(256), 0, s, rb);
}

// The persistent repair launch needs ALL its workgroups resident at once (software grid barrier).  Its size is what the
// device this process sees can hold -- the runtime's occupancy answer for the kernel times the CU count of the current
// device (a partitioned or CU-masked MI355X reports fewer CUs) -- capped at one workgroup per CU of a whole part; 0 = the
// barrier cannot be used here (the multi-launch round runs instead).  Co-resident kernels of OTHER streams or processes
// (RCCL's persistent kernels, a second rank sharing the GPU) are not visible to that query: the barrier's bounded spin and
// the sticky time-out word (bit 2 of the overflow word) cover them.  B3GS_REPAIR_GRID / B3GS_REPAIR_SPINS: test overrides.
static int repair_grid(bool has_val) {
  static int cached[2][16] = {};   // [has_val][device] : 0 = not asked yet, -1 = unusable
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return REPAIR_GRID;
  int& c = cached[has_val ? 1 : 0][dev];
  if (c == 0) {
    static const char* force = getenv("B3GS_REPAIR_GRID");
    int per_cu = 0, cus = 0;
    hipError_t e = has_val ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, repair_kernel<true>, 256, 0)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, repair_kernel<false>, 256, 0);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) { (void)hipGetLastError(); per_cu = 1; cus = REPAIR_GRID; }
    const long resident = (long)per_cu * (long)cus;
    c = resident >= 32 ? (int)(resident < REPAIR_GRID ? resident : REPAIR_GRID) : -1;
    if (force && atoi(force) > 0) c = atoi(force);
  }
  return c;
}
static uint32_t repair_spin_limit() {
  static const uint32_t v = getenv("B3GS_REPAIR_SPINS") ? (uint32_t)strtoul(getenv("B3GS_REPAIR_SPINS"), nullptr, 10) : (1u << 22);
  return v;
}

void b3gs_launch_round2_batch(int32_t P, int nviews, const BinJob* jobs, hipStream_t s) {
  if (nviews <= 0 || P <= 0) return;
  const int K1 = b3gs_seg1_count(jobs[0], P);
  if (K1 >= P) return;
  const int passes = tile_sort_passes(jobs[0].W, jobs[0].H);
  for (int v = 0; v < nviews; v++)   // (the caller only enables K1 for batches of equal tile-sort depth)
    if (tile_sort_passes(jobs[v].W, jobs[v].H) != passes) return;
  const int rest = P - K1;
  const int total_tiles = (rest + SCAN_TILE - 1) / SCAN_TILE;
  Scan2Batch sc;
  sc.n = nviews;
  sc.P = P;
  sc.K1 = K1;
  sc.tiles_per_chunk = (total_tiles + SCAN_MAX_CHUNKS - 1) / SCAN_MAX_CHUNKS;
  sc.nchunks = (total_tiles + sc.tiles_per_chunk - 1) / sc.tiles_per_chunk;
  for (int v = 0; v < nviews; v++) {
    const BinJob& jb = jobs[v];
    sc.j[v] = Scan2Job{depth_order_of(jobs, v), jb.rect ? jb.rect : jb.g.rect, jb.rect ? jb.rect_stride : 1,
                       jb.g.srect, jb.g.scount, jb.g.soffs, jb.g.scan_tmp, jb.g.header, jb.im.header, jb.n_out,
                       open_map(jb.im.open_rows, jb.W, jb.H), jb.high_water, jb.overflow_flag,
                       (uint32_t)(jb.n_bound > 0 ? (jb.n_bound < 0xFFFFFFFFll ? jb.n_bound : 0xFFFFFFFFll) : 0)};
  }
  static const bool legacy = getenv("B3GS_ROUND2_LEGACY") != nullptr;
  bool one_launch = !legacy && passes >= 1 && passes <= 2 && jobs[0].im.header != nullptr;
  if (one_launch) {   // (the value-carrying variant is the larger kernel: if it fits, the other one does)
    bool two_words = false;
    for (int v = 0; v < nviews; v++) two_words = two_words || b3gs_packed_idx_bits(P, jobs[v].W, jobs[v].H) < 0;
    one_launch = repair_grid(two_words) > 0;
  }
  if (!one_launch) {
    hipLaunchKernelGGL(scan2_chunk_sums, dim3(sc.nchunks, nviews), dim3(SCAN_THREADS), 0, s, sc);
    hipLaunchKernelGGL(scan2_chunk_offsets, dim3(nviews), dim3(SCAN_THREADS), 0, s, sc);
    hipLaunchKernelGGL(scan2_chunk_apply, dim3(sc.nchunks, nviews), dim3(SCAN_THREADS), 0, s, sc);
  }

  // emission + stable split by tile id BEHIND segment 1 in the same ping-pong arrays (element offset N1 = header[0],
  // read on the device): the lists end in val[0] / key[0] like segment 1's, the ranges hold absolute positions
  int tbits = 0;
  for (int v = 0; v < nviews; v++) tbits = max(tbits, b3gs_tile_bits(jobs[v].W, jobs[v].H));
  const int first = passes & 1;
  EmitBatch eb;
  eb.n = nviews;
  eb.P = P;
  eb.K1 = P;
  eb.first = K1;
  eb.subs = 1;
  eb.dense_blocks = 0x7FFFFFFF;
  eb.compact_blocks = 0;
  eb.tiles_per_chunk = 1;
  SortBatch tb;
  tb.n = nviews;
  RangeBatch rb;
  rb.n = nviews;
  uint32_t max_cap = 0;
  for (int v = 0; v < nviews; v++) {
    const BinJob& jb = jobs[v];
    const uint32_t n_cap = (uint32_t)(jb.n_bound > 0 ? jb.n_bound : 0);
    max_cap = n_cap > max_cap ? n_cap : max_cap;
    const int gx = (jb.W + B3GS_TILE - 1) / B3GS_TILE;
    const int idx_bits = b3gs_packed_idx_bits(P, jb.W, jb.H);
    const OpenMap om = open_map(jb.im.open_rows, jb.W, jb.H);
    if (idx_bits >= 0) {
      eb.j[v] = EmitJob{depth_order_of(jobs, v), jb.g.soffs, jb.g.srect, jb.b.val[first], nullptr, n_cap, gx, idx_bits,
                        jb.g.scount, om, jb.im.header + 3, jb.g.header, nullptr, nullptr, nullptr, nullptr};
      tb.j[v] = SortJob{jb.b.val[first], nullptr, jb.b.val[first ^ 1], nullptr, jb.g.header + 2, n_cap,
                        b3gs_sort_blocks((int64_t)n_cap), idx_bits, jb.b.hist, nullptr, jb.g.header};
      rb.j[v] = RangeJob{jb.b.val[0], jb.g.header + 2, jb.im.ranges2, n_cap, idx_bits, jb.g.header};
    } else {
      eb.j[v] = EmitJob{depth_order_of(jobs, v), jb.g.soffs, jb.g.srect, jb.b.key[first], jb.b.val[first], n_cap, gx, 0,
                        jb.g.scount, om, jb.im.header + 3, jb.g.header, nullptr, nullptr, nullptr, nullptr};
      tb.j[v] = SortJob{jb.b.key[first], jb.b.val[first], jb.b.key[first ^ 1], jb.b.val[first ^ 1], jb.g.header + 2, n_cap,
                        b3gs_sort_blocks((int64_t)n_cap), 0, jb.b.hist, nullptr, jb.g.header};
      rb.j[v] = RangeJob{jb.b.key[0], jb.g.header + 2, jb.im.ranges2, n_cap, 0, jb.g.header};
    }
  }
  if (max_cap == 0) return;
  if (one_launch) {
    // the whole round as one persistent launch (repair_kernel): exits at once when no tile is open
    RepairArgs ra;
    ra.sc = sc;
    ra.eb = eb;
    ra.passes = passes;
    ra.emit_blocks = (uint32_t)((rest + 255) / 256);
    ra.max_blk = 0;
    bool any_val = false;
    for (int v = 0; v < nviews; v++) {
      ra.max_blk = tb.j[v].nblk > ra.max_blk ? tb.j[v].nblk : ra.max_blk;
      any_val = any_val || tb.j[v].vout != nullptr;
    }
    for (int p = 0; p < 2; p++) {
      ra.bits[p] = min(8, max(0, tbits - 8 * p));
      if (p == passes - 1)
        for (int v = 0; v < nviews; v++) tb.j[v].ranges = jobs[v].im.ranges2;
      ra.tb[p] = tb;
      for (int v = 0; v < nviews; v++) {
        SortJob& j = tb.j[v];
        const uint32_t* k = j.kin; const uint32_t* vv = j.vin;
        j.kin = j.kout; j.vin = j.vout;
        j.kout = const_cast<uint32_t*>(k); j.vout = const_cast<uint32_t*>(vv);
      }
    }
    ra.barrier = jobs[0].im.header + B3GS_HDR_REPAIR_BARRIER;
    for (int v = 0; v < B3GS_MAX_FUSED_VIEWS; v++) ra.overflow[v] = v < nviews ? jobs[v].overflow_flag : nullptr;
    ra.spin_limit = repair_spin_limit();
    static_assert(sizeof(RepairArgs) <= 4000, "kernel arguments of the repair kernel");
    const int grid = repair_grid(any_val);
    if (any_val) hipLaunchKernelGGL(repair_kernel<true>, dim3(grid), dim3(256), 0, s, ra);
    else hipLaunchKernelGGL(repair_kernel<false>, dim3(grid), dim3(256), 0, s, ra);
    return;
  }
  hipLaunchKernelGGL(emit_instances<true>, dim3((rest + 255) / 256, nviews), dim3(256), 0, s, eb);
  for (int p = 0; p < passes; p++) {
    if (p == passes - 1)
      for (int v = 0; v < nviews; v++) tb.j[v].ranges = jobs[v].im.ranges2;
    radix_pass(tb, 8 * p, s, min(8, tbits - 8 * p));
    for (int v = 0; v < nviews; v++) {
      SortJob& j = tb.j[v];
      const uint32_t* k = j.kin; const uint32_t* vv = j.vin;
      j.kin = j.kout; j.vin = j.vout;
      j.kout = const_cast<uint32_t*>(k); j.vout = const_cast<uint32_t*>(vv);
    }
  }
  if (passes == 0) hipLaunchKernelGGL(tile_ranges, dim3((max_cap + 255) / 256, nviews), dim3(256), 0, s, rb);
}
