// Mean squared distance to the 3 nearest neighbours of every point: what the reference gets from
// simple_knn._C.distCUDA2 to initialise the Gaussian scales (scene/gaussian_model.py:134-135,
// submodules/simple-knn/simple_knn.cu:185-221); SURVEY 8f-4.  Exact k-NN, same published scheme: Morton order
// (10 bits per axis), boxes of 1024 consecutive points, per point a bound from its Morton neighbours and a sweep
// over the boxes that can still improve it.  The Morton sort is this library's own radix sort
// (b3gs_launch_sort_u32_index).  A point is excluded by INDEX, not by distance (duplicates give 0), and fewer
// than 4 points leave FLT_MAX terms in the mean, as upstream.  One-off initialisation cost, O(P * P / 1024) box tests.
#include "b3gs_internal.h"
#include <cfloat>

namespace {

constexpr int KNN_BOX = 1024;
struct Box { float lo[3], hi[3]; };

__global__ void __launch_bounds__(256) bbox_partial_kernel(int P, const float* __restrict__ pts, float* __restrict__ part) {
  __shared__ float red[4][6];
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
    for (int a = 0; a < 3; a++) { const float v = pts[3 * (size_t)i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64)); }
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; a++) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
    part[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

__global__ void __launch_bounds__(256) morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ part,
                                                     int nparts, uint32_t* __restrict__ codes) {
  __shared__ float bb[6];
  if (threadIdx.x < 6) {
    // upstream seeds both reductions with 0, i.e. the box always contains the origin
    float v = 0.0f;
    for (int k = 0; k < nparts; k++) v = threadIdx.x < 3 ? fminf(v, part[k * 6 + threadIdx.x]) : fmaxf(v, part[k * 6 + threadIdx.x]);
    bb[threadIdx.x] = v;
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  uint32_t c[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float ext = bb[3 + a] - bb[a];
    const float t = ext > 0.0f ? (pts[3 * (size_t)i + a] - bb[a]) / ext : 0.0f;
    c[a] = spread10((uint32_t)fminf(fmaxf(t * 1023.0f, 0.0f), 1023.0f));
  }
  codes[i] = c[0] | (c[1] << 1) | (c[2] << 2);
}

// points in Morton order (coalesced reads in the sweeps) + per-box bounds
__global__ void __launch_bounds__(KNN_BOX) box_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                       float* __restrict__ sorted, Box* __restrict__ boxes) {
  __shared__ float red[KNN_BOX / 64][6];
  const int i = blockIdx.x * KNN_BOX + threadIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < P) {
    const uint32_t src = order[i];
#pragma unroll
    for (int a = 0; a < 3; a++) { const float v = pts[3 * (size_t)src + a]; sorted[3 * (size_t)i + a] = v; lo[a] = hi[a] = v; }
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64)); }
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; a++) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[0][threadIdx.x];
    for (int w = 1; w < KNN_BOX / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
    if (threadIdx.x < 3) boxes[blockIdx.x].lo[threadIdx.x] = v; else boxes[blockIdx.x].hi[threadIdx.x - 3] = v;
  }
}

__device__ __forceinline__ void insert3(float d, float (&best)[3]) {
#pragma unroll
  for (int j = 0; j < 3; j++)
    if (best[j] > d) { const float t = best[j]; best[j] = d; d = t; }
}
__device__ __forceinline__ float dist2(const float* a, float x, float y, float z) {
  const float dx = a[0] - x, dy = a[1] - y, dz = a[2] - z;
  return dx * dx + dy * dy + dz * dz;
}

__global__ void __launch_bounds__(256) knn3_kernel(int P, const float* __restrict__ sorted, const uint32_t* __restrict__ order,
                                                   const Box* __restrict__ boxes, int nboxes, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float x = sorted[3 * (size_t)i], y = sorted[3 * (size_t)i + 1], z = sorted[3 * (size_t)i + 2];
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  for (int j = max(0, i - 3); j <= min(P - 1, i + 3); j++)
    if (j != i) insert3(dist2(sorted + 3 * (size_t)j, x, y, z), best);
  const float reject = best[2];
  best[0] = best[1] = best[2] = FLT_MAX;
  for (int b = 0; b < nboxes; b++) {
    const Box bx = boxes[b];
    float d = 0.0f;
    const float p[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; a++)
      if (p[a] < bx.lo[a] || p[a] > bx.hi[a]) { const float t = fminf(fabsf(p[a] - bx.lo[a]), fabsf(p[a] - bx.hi[a])); d += t * t; }
    if (d > reject || d > best[2]) continue;
    const int end = min(P, (b + 1) * KNN_BOX);
    for (int j = b * KNN_BOX; j < end; j++)
      if (j != i) insert3(dist2(sorted + 3 * (size_t)j, x, y, z), best);
  }
  out[order[i]] = (best[0] + best[1] + best[2]) / 3.0f;
}

struct KnnWs {
  float* part; uint32_t* codes; uint32_t* skey[2]; uint32_t* sval[2]; uint32_t* hist; float* sorted; Box* boxes;
};
size_t carve_ws(char* base, int32_t P, KnnWs* w) {
  char* cur = base;
  const size_t p = (size_t)(P > 0 ? P : 1);
  KnnWs t;
  t.part = b3gs_carve<float>(cur, 6 * 256);
  t.codes = b3gs_carve<uint32_t>(cur, p);
  for (int k = 0; k < 2; k++) t.skey[k] = b3gs_carve<uint32_t>(cur, p);
  for (int k = 0; k < 2; k++) t.sval[k] = b3gs_carve<uint32_t>(cur, p);
  t.hist = b3gs_carve<uint32_t>(cur, b3gs_sort_scratch_words((int64_t)p));
  t.sorted = b3gs_carve<float>(cur, 3 * p);
  t.boxes = b3gs_carve<Box>(cur, (p + KNN_BOX - 1) / KNN_BOX);
  if (w) *w = t;
  return (size_t)(cur - base);
}

}  // namespace

extern "C" size_t b3gs_knn_workspace_bytes(int32_t P) { return carve_ws(nullptr, P, nullptr); }

extern "C" int b3gs_knn_mean_dist2(int32_t P, const float* points, float* mean_dist2, char* workspace, b3gs_stream_t stream) {
  if (P < 0 || (P > 0 && (!points || !mean_dist2 || !workspace)))
    return b3gs_fail(B3GS_ERR_ARG, "b3gs_knn_mean_dist2", "negative P or NULL points / output / workspace");
  if (P == 0) return B3GS_OK;
  hipStream_t s = (hipStream_t)stream;
  KnnWs w;
  carve_ws(workspace, P, &w);
  const int nparts = (P + 255) / 256 < 256 ? (P + 255) / 256 : 256;
  hipLaunchKernelGGL(bbox_partial_kernel, dim3(nparts), dim3(256), 0, s, P, points, w.part);
  hipLaunchKernelGGL(morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, w.part, nparts, w.codes);
  b3gs_launch_sort_u32_index(w.codes, w.skey, w.sval, (uint32_t)P, w.hist, s);
  const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
  hipLaunchKernelGGL(box_kernel, dim3(nboxes), dim3(KNN_BOX), 0, s, P, points, w.sval[0], w.sorted, w.boxes);
  hipLaunchKernelGGL(knn3_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, w.sorted, w.sval[0], w.boxes, nboxes, mean_dist2);
  return b3gs_launch_status("b3gs_knn_mean_dist2");
}
