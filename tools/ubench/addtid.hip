// What is "TID" in ds_write_addtid_b32 on gfx950: the lane (0..63) or the workgroup thread id?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  __shared__ unsigned s[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) s[i] = 0xFFFFFFFFu;
  __syncthreads();
  const unsigned w = threadIdx.x >> 6;
  const unsigned m0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)s + w * 2048u);
  unsigned v = threadIdx.x;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:0" ::"v"(v), "s"(m0) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) out[i] = s[i];
}
int main() {
  unsigned* d; unsigned h[2048];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int w = 0; w < 4; w++) {
    printf("wave %d region (dwords %d..): first written at", w, w * 512);
    for (int i = 0; i < 2048; i++) if (h[i] == (unsigned)(w * 64)) printf(" dword %d", i);
    printf("\n");
  }
  return 0;
}
