// Micro-benchmark: how a stream of 64-byte slots should be written on gfx950.  One lane owns one slot (the projection's render
// record).  Build on the build box: hipcc --offload-arch=gfx950 -O3 slot_store.hip -o slot_store.bin; run on the GPU box.
//   kind 0  lane-owned: four 16-byte stores per lane, 64 bytes apart between lanes -- every instruction touches 64 lines,
//           a quarter of each (what preprocess_fwd_kernel does)
//   kind 1  quad-cooperative: instruction r writes the four quarters of the slot of lane 4q + r from the four lanes of quad q
//           -- every instruction writes 16 whole lines (the data would have to be transposed inside the quad first)
//   kind 2  lane-owned, 48 of 64 bytes (the record before round 5's padding store)
//   kind 3  lane-owned with ~half of the lanes masked off at random (culled Gaussians)
//   kind 4  quad-cooperative with the same slots masked off
//   kind 5  dense: lane l writes 16 bytes at 16 l of four 1 KB rows (the ideal stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void __launch_bounds__(256) k(float4* out, size_t nslots, float seed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nslots) return;
  const float4 v = make_float4(seed, seed + 1.f, seed + 2.f, (float)i);
  const unsigned lane = threadIdx.x & 63u;
  const bool keep = ((uint32_t)(i * 2654435761u) >> 16) & 1u;
  if (KIND == 0 || KIND == 2 || KIND == 3) {
    if (KIND == 3 && !keep) return;
    float4* rec = out + 4 * i;
    rec[0] = v; rec[1] = v; rec[2] = v;
    if (KIND != 2) rec[3] = v;
  } else if (KIND == 1 || KIND == 4) {
    const size_t quad0 = i & ~(size_t)3;
    const unsigned j = lane & 3u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const size_t slot = quad0 + r;
      bool on = true;
      if (KIND == 4) on = ((uint32_t)(slot * 2654435761u) >> 16) & 1u;
      if (on) out[4 * slot + j] = v;
    }
  } else {
    const size_t wave0 = i & ~(size_t)63;
#pragma unroll
    for (int r = 0; r < 4; r++) out[4 * wave0 + 64 * r + lane] = v;
  }
}
template <int KIND>
void run(const char* name, float4* buf, size_t nslots) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (int)((nslots + 255) / 256);
  float best = 1e9f;
  for (int rep = 0; rep < 6; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, buf, nslots, (float)rep);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("%-44s %8.1f us  %6.2f TB/s of slot bytes\n", name, best * 1e3, nslots * 64.0 / (best * 1e-3) / 1e12);
}
int main() {
  const size_t nslots = 6u * 1000000u;   // six views x 1M Gaussians
  float4* buf; hipMalloc(&buf, nslots * 64 + 4096);
  hipMemset(buf, 0, nslots * 64);
  run<5>("dense rows (ideal)", buf, nslots);
  run<0>("lane-owned 4 x 16 B, stride 64", buf, nslots);
  run<1>("quad-cooperative whole lines", buf, nslots);
  run<2>("lane-owned 48 of 64 B", buf, nslots);
  run<3>("lane-owned, half of the slots", buf, nslots);
  run<4>("quad-cooperative, half of the slots", buf, nslots);
  return 0;
}
