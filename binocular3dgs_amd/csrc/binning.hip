// Tile binning: depth order, instance emission, stable tile split, per-tile ranges.
//
// The published rasterizer (and oracle/tile_ref.c) sorts N (tile<<32 | depth-bits) 64-bit keys in
// one stable radix sort, 6 byte-passes over N at 800x600.  The same permutation is produced here
// with far less HBM traffic by sorting in two levels:
//   1. stable LSD radix sort of the P Gaussians by their 32 depth bits (4 passes over P;
//      culled Gaussians carry key 0xFFFFFFFF and sink to the end; ties keep index order),
//   2. emit the (tile, index) instances in that depth order (wave-cooperative expansion:
//      coalesced writes regardless of how many tiles one Gaussian covers),
//   3. stable LSD radix sort of the N instances by tile id only (ceil(log2(tiles)/8) passes,
//      2 at 800x600 and 1600x1600) -- stability keeps the depth/index order inside each tile.
// Resulting point_list is bit-identical to the 64-bit sort (tests/test_binning_parity.py).
//
// All kernels read the element count from device memory (N is produced on the device), so the
// same launches serve the sync-free forward; grids are sized from a host-side bound.
#include "b3gs_internal.h"
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

// ---------------------------------------------------------------------------------------------
// scan of tiles_touched in depth order: soffs[s] = sum_{s' <= s} tiles_touched[sval[s']]
// three launches: per-chunk sums, scan of chunk sums (+ N, V to the header), per-chunk rescan
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096
constexpr int SCAN_MAX_CHUNKS = 2048;

__global__ void __launch_bounds__(SCAN_THREADS)
    scan_chunk_sums(int P, int tiles_per_chunk, const uint32_t* __restrict__ order, const uint32_t* __restrict__ touched,
                    uint32_t* __restrict__ chunk_sums, uint32_t* __restrict__ chunk_vis) {
  __shared__ uint32_t tmp[8];
  const int64_t begin = (int64_t)blockIdx.x * tiles_per_chunk * SCAN_TILE;
  const int64_t end = min((int64_t)P, begin + (int64_t)tiles_per_chunk * SCAN_TILE);
  uint32_t sum = 0, vis = 0;
  for (int64_t i = begin + threadIdx.x; i < end; i += SCAN_THREADS) {
    uint32_t t = touched[order[i]];
    sum += t;
    vis += (t != 0);
  }
  uint32_t tot, tot2;
  block_excl_scan_256(sum, tmp, &tot);
  block_excl_scan_256(vis, tmp, &tot2);
  if (threadIdx.x == 0) {
    chunk_sums[blockIdx.x] = tot;
    chunk_vis[blockIdx.x] = tot2;
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
    scan_chunk_offsets(int nchunks, uint32_t* __restrict__ chunk_sums, const uint32_t* __restrict__ chunk_vis,
                       uint32_t* __restrict__ header, uint32_t* __restrict__ img_header, int32_t* __restrict__ n_out) {
  __shared__ uint32_t tmp[8];
  // nchunks <= 2048: 8 per thread, sequential
  uint32_t loc[8], s = 0, v = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int idx = threadIdx.x * 8 + k;
    loc[k] = idx < nchunks ? chunk_sums[idx] : 0u;
    s += loc[k];
    v += idx < nchunks ? chunk_vis[idx] : 0u;
  }
  uint32_t tot, totv;
  uint32_t base = block_excl_scan_256(s, tmp, &tot);
  block_excl_scan_256(v, tmp, &totv);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int idx = threadIdx.x * 8 + k;
    if (idx < nchunks) chunk_sums[idx] = base;
    base += loc[k];
  }
  if (threadIdx.x == 0) {
    header[0] = tot;   // N
    header[1] = totv;  // V
    if (img_header) { img_header[0] = tot; img_header[1] = totv; }
    if (n_out) *n_out = (int32_t)tot;
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
    scan_chunk_apply(int P, int tiles_per_chunk, const uint32_t* __restrict__ order, const uint32_t* __restrict__ touched,
                     const uint32_t* __restrict__ chunk_offs, uint32_t* __restrict__ soffs) {
  __shared__ uint32_t tmp[8];
  uint32_t carry = chunk_offs[blockIdx.x];
  const int64_t begin = (int64_t)blockIdx.x * tiles_per_chunk * SCAN_TILE;
  for (int t = 0; t < tiles_per_chunk; t++) {
    const int64_t tb = begin + (int64_t)t * SCAN_TILE;
    if (tb >= P) break;
    // blocked arrangement: thread owns SCAN_ITEMS consecutive elements
    uint32_t loc[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
      int64_t i = tb + (int64_t)threadIdx.x * SCAN_ITEMS + k;
      loc[k] = i < P ? touched[order[i]] : 0u;
      s += loc[k];
    }
    uint32_t tot;
    uint32_t run = carry + block_excl_scan_256(s, tmp, &tot);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
      int64_t i = tb + (int64_t)threadIdx.x * SCAN_ITEMS + k;
      run += loc[k];
      if (i < P) soffs[i] = run;
    }
    carry += tot;
  }
}

// ---------------------------------------------------------------------------------------------
// stable LSD radix pass on (u32 key, u32 value) pairs, 8-bit digit at `shift`
//   hist:    per-workgroup digit counts, stored digit-major  hist[d * nblk + blk]
//   rowscan: one workgroup per digit: exclusive prefix over workgroups, digit total -> totals[d]
//   scatter: wave-striped stable ranking (ballot match), LDS reorder, coalesced run writes
// n is read from *n_ptr and clamped to n_cap.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(B3GS_SORT_THREADS)
    radix_hist(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_ptr, uint32_t n_cap, int shift,
               uint32_t nblk, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  const uint32_t n = n_ptr ? min(*n_ptr, n_cap) : n_cap;
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * B3GS_SORT_TILE;
  if (base < n) {
#pragma unroll
    for (int k = 0; k < B3GS_SORT_ITEMS; k++) {
      uint32_t i = base + k * B3GS_SORT_THREADS + threadIdx.x;
      if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xFF], 1u);
    }
  }
  __syncthreads();
  hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(256) radix_rowscan(uint32_t nblk, uint32_t* __restrict__ hist, uint32_t* __restrict__ totals) {
  __shared__ uint32_t tmp[8];
  uint32_t* row = hist + (size_t)blockIdx.x * nblk;
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nblk; b0 += 256) {
    uint32_t i = b0 + threadIdx.x;
    uint32_t v = i < nblk ? row[i] : 0u;
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(v, tmp, &tot);
    if (i < nblk) row[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(B3GS_SORT_THREADS)
    radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                  uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr,
                  uint32_t n_cap, int shift, uint32_t nblk, const uint32_t* __restrict__ hist,
                  const uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_cnt[4][256];
  __shared__ uint32_t blk_start[256];  // first slot of digit d inside this workgroup's reorder buffer
  __shared__ uint32_t gbase[256];      // global destination of that first slot
  __shared__ uint32_t tmp[8];
  __shared__ uint32_t s_key[B3GS_SORT_TILE];
  __shared__ uint32_t s_val[B3GS_SORT_TILE];

  const uint32_t n = n_ptr ? min(*n_ptr, n_cap) : n_cap;
  const uint32_t tile_base = blockIdx.x * B3GS_SORT_TILE;
  if (tile_base >= n) return;  // uniform per workgroup
  const uint32_t tile_n = min((uint32_t)B3GS_SORT_TILE, n - tile_base);
  const unsigned lane = lane_id(), w = threadIdx.x >> 6;
  const u64 lt = lanemask_lt();

#pragma unroll
  for (int k = 0; k < 4; k++) wave_cnt[k][threadIdx.x] = 0;
  __syncthreads();

  // wave w owns the contiguous slab [w*1024, (w+1)*1024) of the tile, 16 rounds of 64
  uint32_t key[B3GS_SORT_ITEMS], val[B3GS_SORT_ITEMS], rank[B3GS_SORT_ITEMS];
#pragma unroll
  for (int r = 0; r < B3GS_SORT_ITEMS; r++) {
    const uint32_t li = w * (B3GS_SORT_ITEMS * 64) + r * 64 + lane;
    const bool valid = li < tile_n;
    const uint32_t gi = tile_base + li;
    key[r] = valid ? keys_in[gi] : 0xFFFFFFFFu;
    val[r] = valid ? (vals_in ? vals_in[gi] : gi) : 0u;
  }
#pragma unroll
  for (int r = 0; r < B3GS_SORT_ITEMS; r++) {
    const uint32_t li = w * (B3GS_SORT_ITEMS * 64) + r * 64 + lane;
    const bool valid = li < tile_n;
    const uint32_t d = (key[r] >> shift) & 0xFF;
    u64 m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
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
    gbase[d] = dig_base + hist[(size_t)d * nblk + blockIdx.x];
  }
  __syncthreads();

#pragma unroll
  for (int r = 0; r < B3GS_SORT_ITEMS; r++) {
    const uint32_t li = w * (B3GS_SORT_ITEMS * 64) + r * 64 + lane;
    if (li < tile_n) {
      const uint32_t d = (key[r] >> shift) & 0xFF;
      const uint32_t p = wave_cnt[w][d] + rank[r];
      s_key[p] = key[r];
      s_val[p] = val[r];
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
      vals_out[dst] = s_val[p];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Single-launch radix pass ("onesweep"): histogram, cross-workgroup prefix and scatter in ONE
// kernel per 8-bit digit.  The global digit histograms of all passes are produced up front (for
// the depth sort by radix_global_hist, for the tile sort inside emit_instances), so a pass only
// needs, per digit, the number of keys in EARLIER workgroups: chained scan with decoupled
// look-back.  Inter-workgroup protocol (placement independent, MI355X guide G16 "granule" form):
// one 32-bit word per (workgroup, digit) = flag[31:30] | count[29:0], written with ONE agent-scope
// relaxed atomic store and polled with agent-scope relaxed atomic loads -- value and flag travel
// in the same word, so no separate payload needs release/acquire ordering.  Workgroup ids come
// from an atomic ticket, so every predecessor a workgroup waits for has already started (no
// dependence on dispatch order); every spin is bounded.
// Scratch layout (words): ghist[4][256] | tickets[16] | pad | status[pass][nblk][256]
// ---------------------------------------------------------------------------------------------
constexpr uint32_t OS_GHIST = 0, OS_TICKET = 1024, OS_STATUS = 1280;
constexpr uint32_t OS_FLAG_AGG = 1u << 30, OS_FLAG_INC = 2u << 30, OS_VAL_MASK = (1u << 30) - 1u;

__global__ void __launch_bounds__(B3GS_SORT_THREADS)
    radix_global_hist(const uint32_t* __restrict__ keys, uint32_t n, int passes, uint32_t* __restrict__ scratch) {
  __shared__ uint32_t h[4][256];
#pragma unroll
  for (int p = 0; p < 4; p++) h[p][threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * B3GS_SORT_TILE;
#pragma unroll
  for (int k = 0; k < B3GS_SORT_ITEMS; k++) {
    const uint32_t i = base + k * B3GS_SORT_THREADS + threadIdx.x;
    if (i < n) {
      const uint32_t key = keys[i];
      for (int p = 0; p < passes; p++) atomicAdd(&h[p][(key >> (8 * p)) & 0xFF], 1u);
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; p++) {
    const uint32_t c = h[p][threadIdx.x];
    if (c) atomicAdd(&scratch[OS_GHIST + p * 256 + threadIdx.x], c);
  }
}

__global__ void __launch_bounds__(B3GS_SORT_THREADS)
    radix_onesweep(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                   uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr,
                   uint32_t n_cap, int pass, uint32_t nblk, uint32_t* __restrict__ scratch) {
  __shared__ uint32_t wave_cnt[4][256];
  __shared__ uint32_t blk_start[256];
  __shared__ uint32_t gbase[256];
  __shared__ uint32_t tmp[8];
  __shared__ uint32_t s_blk;
  __shared__ uint32_t s_key[B3GS_SORT_TILE];
  __shared__ uint32_t s_val[B3GS_SORT_TILE];

  const int shift = 8 * pass;
  const uint32_t n = n_ptr ? min(*n_ptr, n_cap) : n_cap;
  if (threadIdx.x == 0) s_blk = atomicAdd(&scratch[OS_TICKET + pass], 1u);
#pragma unroll
  for (int k = 0; k < 4; k++) wave_cnt[k][threadIdx.x] = 0;
  __syncthreads();
  const uint32_t blk = s_blk;
  uint32_t* status = scratch + OS_STATUS + ((size_t)pass * nblk + blk) * 256;
  const uint32_t tile_base = blk * B3GS_SORT_TILE;
  if (tile_base >= n) {  // past the end: publish zeros so that nobody ever waits on this workgroup
    __hip_atomic_store(&status[threadIdx.x], OS_FLAG_INC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const uint32_t tile_n = min((uint32_t)B3GS_SORT_TILE, n - tile_base);
  const unsigned lane = lane_id(), w = threadIdx.x >> 6;
  const u64 lt = lanemask_lt();

  uint32_t key[B3GS_SORT_ITEMS], val[B3GS_SORT_ITEMS], rank[B3GS_SORT_ITEMS];
#pragma unroll
  for (int r = 0; r < B3GS_SORT_ITEMS; r++) {
    const uint32_t li = w * (B3GS_SORT_ITEMS * 64) + r * 64 + lane;
    const bool valid = li < tile_n;
    const uint32_t gi = tile_base + li;
    key[r] = valid ? keys_in[gi] : 0xFFFFFFFFu;
    val[r] = valid ? (vals_in ? vals_in[gi] : gi) : 0u;
  }
#pragma unroll
  for (int r = 0; r < B3GS_SORT_ITEMS; r++) {
    const uint32_t li = w * (B3GS_SORT_ITEMS * 64) + r * 64 + lane;
    const bool valid = li < tile_n;
    const uint32_t d = (key[r] >> shift) & 0xFF;
    u64 m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
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

  {
    const uint32_t d = threadIdx.x;
    const uint32_t c0 = wave_cnt[0][d], c1 = wave_cnt[1][d], c2 = wave_cnt[2][d], c3 = wave_cnt[3][d];
    const uint32_t tot = c0 + c1 + c2 + c3;
    // publish this workgroup's count of digit d, then look back for the exclusive prefix
    __hip_atomic_store(&status[d], OS_FLAG_AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0;
    for (int pb = (int)blk - 1; pb >= 0; pb--) {
      const uint32_t* ps = scratch + OS_STATUS + ((size_t)pass * nblk + (uint32_t)pb) * 256 + d;
      uint32_t v = 0;
      for (int spin = 0; spin < (1 << 24); spin++) {
        v = __hip_atomic_load(ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >> 30) break;
        __builtin_amdgcn_s_sleep(1);
      }
      excl += v & OS_VAL_MASK;
      if ((v >> 30) != 1u) break;  // inclusive prefix found (or spin bound hit: fail soft, never hang)
    }
    __hip_atomic_store(&status[d], OS_FLAG_INC | ((excl + tot) & OS_VAL_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    uint32_t dummy, dummy2;
    const uint32_t start = block_excl_scan_256(tot, tmp, &dummy);
    const uint32_t dig_base = block_excl_scan_256(scratch[OS_GHIST + pass * 256 + d], tmp, &dummy2);
    wave_cnt[0][d] = start;
    wave_cnt[1][d] = start + c0;
    wave_cnt[2][d] = start + c0 + c1;
    wave_cnt[3][d] = start + c0 + c1 + c2;
    blk_start[d] = start;
    gbase[d] = dig_base + excl;
  }
  __syncthreads();

#pragma unroll
  for (int r = 0; r < B3GS_SORT_ITEMS; r++) {
    const uint32_t li = w * (B3GS_SORT_ITEMS * 64) + r * 64 + lane;
    if (li < tile_n) {
      const uint32_t d = (key[r] >> shift) & 0xFF;
      const uint32_t p = wave_cnt[w][d] + rank[r];
      s_key[p] = key[r];
      s_val[p] = val[r];
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
      vals_out[dst] = s_val[p];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// instance emission in depth order (wave-cooperative expansion)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    emit_instances(int P, int grid_x, const uint32_t* __restrict__ order, const uint32_t* __restrict__ soffs,
                   const uint32_t* __restrict__ touched, const uint2* __restrict__ rect, uint32_t n_cap,
                   uint32_t* __restrict__ tile_out, uint32_t* __restrict__ idx_out, int hist_passes,
                   uint32_t* __restrict__ os_scratch) {
  __shared__ uint32_t s_end[4][64];
  __shared__ uint32_t s_hist[4][256];  // digit histograms of the tile ids this workgroup emits (onesweep input)
  if (hist_passes > 0) {
#pragma unroll
    for (int p = 0; p < 4; p++) s_hist[p][threadIdx.x] = 0;
    __syncthreads();
  }
  const unsigned lane = lane_id(), w = threadIdx.x >> 6;
  const int s = blockIdx.x * 256 + threadIdx.x;
  uint32_t gid = 0, cnt = 0, end = 0;
  uint2 rc = make_uint2(0, 0);
  if (s < P) {
    gid = order[s];
    cnt = touched[gid];
    end = soffs[s];
    rc = rect[gid];
  }
  // lanes past P inherit the last valid end so the search array stays monotone
  const uint32_t wave_end = __shfl(end, 63 - (int)__builtin_clzll(__ballot(s < P) | 1ull), 64);
  if (s >= P) end = wave_end;
  s_end[w][lane] = end;
  const uint32_t wave_begin = __shfl(end - cnt, 0, 64);
  __builtin_amdgcn_wave_barrier();
  const bool wave_active = __ballot(cnt != 0) != 0;  // waves of culled Gaussians (sorted to the end) emit nothing

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
    if (j < wave_end && j < n_cap) {
      const uint32_t k = j - (src_end - src_cnt);
      const uint32_t ry = k / src_rw, rx = k - ry * src_rw;
      const uint32_t tile = (src_y0 + ry) * (uint32_t)grid_x + (src_x0 + rx);
      tile_out[j] = tile;
      idx_out[j] = src_gid;
      for (int p = 0; p < hist_passes; p++) atomicAdd(&s_hist[p][(tile >> (8 * p)) & 0xFF], 1u);
    }
  }
  if (hist_passes > 0) {
    __syncthreads();
    for (int p = 0; p < hist_passes; p++) {
      const uint32_t c = s_hist[p][threadIdx.x];
      if (c) atomicAdd(&os_scratch[OS_GHIST + p * 256 + threadIdx.x], c);
    }
  }
}

__global__ void __launch_bounds__(256)
    tile_ranges(const uint32_t* __restrict__ tile_sorted, const uint32_t* __restrict__ n_ptr, uint32_t n_cap,
                uint2* __restrict__ ranges) {
  const uint32_t n = min(*n_ptr, n_cap);
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t t = tile_sorted[j];
  if (j == 0 || tile_sorted[j - 1] != t) ranges[t].x = j;
  if (j == n - 1 || tile_sorted[j + 1] != t) ranges[t].y = j + 1;
}

void radix_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, const uint32_t* n_ptr,
                uint32_t n_cap, int shift, uint32_t* hist, hipStream_t s) {
  const uint32_t nblk = b3gs_sort_blocks((int64_t)n_cap);
  uint32_t* totals = hist + (size_t)256 * nblk;  // 256 spare words at the tail of the scratch
  hipLaunchKernelGGL(radix_hist, dim3(nblk), dim3(B3GS_SORT_THREADS), 0, s, kin, n_ptr, n_cap, shift, nblk, hist);
  hipLaunchKernelGGL(radix_rowscan, dim3(256), dim3(256), 0, s, nblk, hist, totals);
  hipLaunchKernelGGL(radix_scatter, dim3(nblk), dim3(B3GS_SORT_THREADS), 0, s, kin, vin, kout, vout, n_ptr, n_cap, shift,
                     nblk, hist, totals);
}

// Measured on MI355X (P = 1M, N = 4.8M, 6 passes per view): the chained-scan pass is correct but
// SLOWER than histogram / row scan / scatter as three launches (sort stage 352 us vs 274 us per view):
// a dependent kernel boundary costs ~1.5 us here, a cross-workgroup hand-off through agent-scope
// atomics ~1-2 us PER look-back hop.  The three-launch pass is therefore the default;
// B3GS_SORT=onesweep selects the single-launch pass for A/B runs.
bool use_onesweep() {
  static const bool v = getenv("B3GS_SORT") && getenv("B3GS_SORT")[0] == 'o';
  return v;
}
size_t onesweep_scratch_words(uint32_t nblk, int passes) { return OS_STATUS + (size_t)passes * nblk * 256; }
void onesweep_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, const uint32_t* n_ptr,
                   uint32_t n_cap, int pass, uint32_t* scratch, hipStream_t s) {
  const uint32_t nblk = b3gs_sort_blocks((int64_t)n_cap);
  hipLaunchKernelGGL(radix_onesweep, dim3(nblk), dim3(B3GS_SORT_THREADS), 0, s, kin, vin, kout, vout, n_ptr, n_cap, pass,
                     nblk, scratch);
}

}  // namespace

void b3gs_launch_depth_sort_and_scan(int32_t P, const GeomView& g, uint32_t* img_header, int32_t* n_out,
                                     hipStream_t s) {
  if (P <= 0) {
    (void)hipMemsetAsync(g.header, 0, 8, s);
    if (img_header) (void)hipMemsetAsync(img_header, 0, 8, s);
    if (n_out) (void)hipMemsetAsync(n_out, 0, 4, s);
    return;
  }
  // 4 passes: depth_key -> skey[1] -> skey[0] -> skey[1] -> skey[0]
  if (use_onesweep()) {
    const uint32_t nblk = b3gs_sort_blocks((int64_t)P);
    (void)hipMemsetAsync(g.hist, 0, onesweep_scratch_words(nblk, 4) * sizeof(uint32_t), s);
    hipLaunchKernelGGL(radix_global_hist, dim3(nblk), dim3(B3GS_SORT_THREADS), 0, s, g.depth_key, (uint32_t)P, 4, g.hist);
    onesweep_pass(g.depth_key, nullptr, g.skey[1], g.sval[1], nullptr, (uint32_t)P, 0, g.hist, s);
    onesweep_pass(g.skey[1], g.sval[1], g.skey[0], g.sval[0], nullptr, (uint32_t)P, 1, g.hist, s);
    onesweep_pass(g.skey[0], g.sval[0], g.skey[1], g.sval[1], nullptr, (uint32_t)P, 2, g.hist, s);
    onesweep_pass(g.skey[1], g.sval[1], g.skey[0], g.sval[0], nullptr, (uint32_t)P, 3, g.hist, s);
  } else {
    const uint32_t* n_ptr = nullptr;  // P is known on the host
    radix_pass(g.depth_key, nullptr, g.skey[1], g.sval[1], n_ptr, (uint32_t)P, 0, g.hist, s);
    radix_pass(g.skey[1], g.sval[1], g.skey[0], g.sval[0], n_ptr, (uint32_t)P, 8, g.hist, s);
    radix_pass(g.skey[0], g.sval[0], g.skey[1], g.sval[1], n_ptr, (uint32_t)P, 16, g.hist, s);
    radix_pass(g.skey[1], g.sval[1], g.skey[0], g.sval[0], n_ptr, (uint32_t)P, 24, g.hist, s);
  }

  const int total_tiles = (P + SCAN_TILE - 1) / SCAN_TILE;
  const int tiles_per_chunk = (total_tiles + SCAN_MAX_CHUNKS - 1) / SCAN_MAX_CHUNKS;
  const int nchunks = (total_tiles + tiles_per_chunk - 1) / tiles_per_chunk;
  uint32_t* chunk_sums = g.scan_tmp;
  uint32_t* chunk_vis = g.scan_tmp + SCAN_MAX_CHUNKS;
  hipLaunchKernelGGL(scan_chunk_sums, dim3(nchunks), dim3(SCAN_THREADS), 0, s, P, tiles_per_chunk, g.sval[0],
                     g.tiles_touched, chunk_sums, chunk_vis);
  hipLaunchKernelGGL(scan_chunk_offsets, dim3(1), dim3(SCAN_THREADS), 0, s, nchunks, chunk_sums, chunk_vis, g.header,
                     img_header, n_out);
  hipLaunchKernelGGL(scan_chunk_apply, dim3(nchunks), dim3(SCAN_THREADS), 0, s, P, tiles_per_chunk, g.sval[0],
                     g.tiles_touched, chunk_sums, g.soffs);
}

void b3gs_launch_binning(int32_t P, int32_t W, int32_t H, int64_t n_bound, const GeomView& g, const BinView& b,
                         const ImgView& im, hipStream_t s) {
  const int gx = (W + B3GS_TILE - 1) / B3GS_TILE, gy = (H + B3GS_TILE - 1) / B3GS_TILE;
  const size_t tiles = (size_t)gx * gy;
  // im.ranges was zeroed by the preprocess launch (empty tiles keep [0,0))
  if (P <= 0 || n_bound <= 0) return;
  const uint32_t n_cap = (uint32_t)n_bound;
  const uint32_t* n_ptr = g.header;  // N
  int tbits = 0;
  while (((size_t)1 << tbits) < tiles) tbits++;
  const int passes = tbits == 0 ? 0 : (tbits + 7) / 8;
  // emit into the buffer from which `passes` ping-pongs end in [0]
  const int first = passes & 1;
  const bool os = use_onesweep() && passes <= 4;
  if (os && passes > 0)
    (void)hipMemsetAsync(b.hist, 0, onesweep_scratch_words(b3gs_sort_blocks((int64_t)n_cap), passes) * sizeof(uint32_t), s);
  hipLaunchKernelGGL(emit_instances, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, g.sval[0], g.soffs,
                     g.tiles_touched, g.rect, n_cap, b.key[first], b.val[first], os ? passes : 0, b.hist);
  int cur = first;
  for (int p = 0; p < passes; p++) {
    if (os) onesweep_pass(b.key[cur], b.val[cur], b.key[cur ^ 1], b.val[cur ^ 1], n_ptr, n_cap, p, b.hist, s);
    else radix_pass(b.key[cur], b.val[cur], b.key[cur ^ 1], b.val[cur ^ 1], n_ptr, n_cap, 8 * p, b.hist, s);
    cur ^= 1;
  }
  hipLaunchKernelGGL(tile_ranges, dim3((n_cap + 255) / 256), dim3(256), 0, s, b.key[0], n_ptr, n_cap, im.ranges);
}
