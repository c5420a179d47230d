// Micro-benchmark for VERDICT r4 item 4 ("bin first, order later"): what would an UNORDERED per-tile emission followed by a
// per-tile sort in LDS cost, against the chain it would replace (depth sort of the P Gaussians + in-order emission + two
// tile-split radix passes)?  Kernels, each the cheapest honest form of its stage:
//   count     one atomic per instance into its tile's counter (instances arrive in Gaussian-index order, a Gaussian's tiles
//             consecutive -- the order an unsorted emission produces)
//   scatter   one returning atomic per instance on the tile's cursor, then two 4-byte stores (depth key, Gaussian index)
//   tilesort  one workgroup per tile: the tile's (key, index) pairs through LDS, bitonic sort by (key, index), written back
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/binfirst.hip -o tools/ubench/binfirst.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__global__ void __launch_bounds__(256) k_count(const uint32_t* __restrict__ tile, int64_t n, uint32_t* __restrict__ cnt) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) atomicAdd(&cnt[tile[i]], 1u);
}

__global__ void __launch_bounds__(256) k_scatter(const uint32_t* __restrict__ tile, const uint32_t* __restrict__ key,
                                                 const uint32_t* __restrict__ idx, int64_t n, uint32_t* __restrict__ cursor,
                                                 uint32_t* __restrict__ okey, uint32_t* __restrict__ oidx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t p = atomicAdd(&cursor[tile[i]], 1u);
    okey[p] = key[i];
    oidx[p] = idx[i];
  }
}

// the same with the pair written as ONE 8-byte store
__global__ void __launch_bounds__(256) k_scatter64(const uint32_t* __restrict__ tile, const uint32_t* __restrict__ key,
                                                   const uint32_t* __restrict__ idx, int64_t n, uint32_t* __restrict__ cursor,
                                                   uint2* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t p = atomicAdd(&cursor[tile[i]], 1u);
    out[p] = make_uint2(key[i], idx[i]);
  }
}

// one workgroup per tile (blockIdx.x -> tile through `order`: longest first); lists longer than CAP are sorted in CAP-sized
// pieces only (a real implementation would merge them: counted separately by the caller)
template <int CAP>
__device__ __forceinline__ void tile_sort(const uint32_t* __restrict__ begin, const uint32_t* __restrict__ order, uint2* __restrict__ data) {
  __shared__ unsigned long long s[CAP];
  const uint32_t t = order[blockIdx.x];
  const uint32_t lo = begin[t], hi = begin[t + 1];
  for (uint32_t base = lo; base < hi; base += CAP) {
    const uint32_t n = min((uint32_t)CAP, hi - base);
    uint32_t m = 64;
    while (m < n) m <<= 1;
    for (uint32_t i = threadIdx.x; i < m; i += 256) {
      unsigned long long v = ~0ull;
      if (i < n) { const uint2 e = data[base + i]; v = ((unsigned long long)e.x << 32) | e.y; }
      s[i] = v;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t i = threadIdx.x; i < m; i += 256) {
          const uint32_t l = i ^ j;
          if (l > i) {
            const unsigned long long a = s[i], b = s[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { s[i] = b; s[l] = a; }
          }
        }
        __syncthreads();
      }
    for (uint32_t i = threadIdx.x; i < n; i += 256) { const unsigned long long v = s[i]; data[base + i] = make_uint2((uint32_t)(v >> 32), (uint32_t)v); }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_tilesort_2048(const uint32_t* begin, const uint32_t* order, uint2* data) { tile_sort<2048>(begin, order, data); }
__global__ void __launch_bounds__(256) k_tilesort_4096(const uint32_t* begin, const uint32_t* order, uint2* data) { tile_sort<4096>(begin, order, data); }

static float timed(hipStream_t s, int reps, void (*fn)(void*, hipStream_t), void* ctx) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  fn(ctx, s);
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int r = 0; r < reps; r++) fn(ctx, s);
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms * 1000.f / reps;
}

struct Ctx { const uint32_t *tile, *key, *idx; int64_t n; uint32_t *cnt, *cursor, *okey, *oidx; uint2* out; const uint32_t *begin, *order;
             uint32_t tiles; int cap; const uint32_t* begin_src; };

static void f_count(void* c_, hipStream_t s) {
  Ctx* c = (Ctx*)c_;
  hipMemsetAsync(c->cnt, 0, 4 * (size_t)c->tiles, s);
  hipLaunchKernelGGL(k_count, dim3(4096), dim3(256), 0, s, c->tile, c->n, c->cnt);
}
static void f_scatter(void* c_, hipStream_t s) {
  Ctx* c = (Ctx*)c_;
  hipMemcpyAsync(c->cursor, c->begin_src, 4 * (size_t)c->tiles, hipMemcpyDeviceToDevice, s);
  hipLaunchKernelGGL(k_scatter, dim3(4096), dim3(256), 0, s, c->tile, c->key, c->idx, c->n, c->cursor, c->okey, c->oidx);
}
static void f_scatter64(void* c_, hipStream_t s) {
  Ctx* c = (Ctx*)c_;
  hipMemcpyAsync(c->cursor, c->begin_src, 4 * (size_t)c->tiles, hipMemcpyDeviceToDevice, s);
  hipLaunchKernelGGL(k_scatter64, dim3(4096), dim3(256), 0, s, c->tile, c->key, c->idx, c->n, c->cursor, c->out);
}
static void f_sort(void* c_, hipStream_t s) {
  Ctx* c = (Ctx*)c_;
  if (c->cap == 2048) hipLaunchKernelGGL(k_tilesort_2048, dim3(c->tiles), dim3(256), 0, s, c->begin, c->order, c->out);
  else hipLaunchKernelGGL(k_tilesort_4096, dim3(c->tiles), dim3(256), 0, s, c->begin, c->order, c->out);
}

}  // namespace

// All pointers device.  tile/key/idx: n instances in emission (Gaussian-index) order, tile ids already offset per view
// (tiles = views x tiles per view).  begin: exclusive prefix of the per-tile counts [tiles + 1]; order: tiles by decreasing
// length.  times_us[0..4] <- count, scatter (2 x 4 B), scatter (8 B), tile sort (CAP 2048), tile sort (CAP 4096)
extern "C" int binfirst_run(const uint32_t* tile, const uint32_t* key, const uint32_t* idx, int64_t n, uint32_t tiles, const uint32_t* begin,
                 const uint32_t* order, uint32_t* cnt, uint32_t* cursor, uint32_t* okey, uint32_t* oidx, uint2* out, int reps,
                 float* times_us) {
  Ctx c{tile, key, idx, n, cnt, cursor, okey, oidx, out, begin, order, tiles, 2048, begin};
  hipStream_t s = 0;
  times_us[0] = timed(s, reps, f_count, &c);
  times_us[1] = timed(s, reps, f_scatter, &c);
  times_us[2] = timed(s, reps, f_scatter64, &c);
  // (the sort runs on the scattered data; sorting sorted data again costs the same compare-exchange network)
  times_us[3] = timed(s, reps, f_sort, &c);
  c.cap = 4096;
  times_us[4] = timed(s, reps, f_sort, &c);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
