// Micro-benchmark: VALU issue cost per wave64 instruction on gfx950 for the instruction kinds the
// blend kernels are made of.  Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (KIND == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
      if (KIND == 1) a[i] = a[i] * 1.0001f;
      if (KIND == 2) { int t = __builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0xB1, 0xF, 0xF, false); a[i] = a[i] + __int_as_float(t); }
      if (KIND == 3) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if (KIND == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (KIND == 5) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[(i + 1) & 15]), false, false); a[i] = __uint_as_float(r[0]); a[(i + 1) & 15] = __uint_as_float(r[1]); }
      if (KIND == 6) a[i] = fminf(a[i], 0.99f);
      if (KIND == 7) a[i] = (a[i] > 0.5f) ? a[i] : 0.25f;   // v_cmp + v_cndmask
      if (KIND == 9 && (i & 1) == 0) {  // one v_pk_fma_f32 per PAIR of elements: 8 instructions per 16 elements
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f x = {a[i], a[i + 1]};
        x = __builtin_elementwise_fma(x, (v2f){1.0001f, 1.0002f}, (v2f){0.5f, 0.25f});
        a[i] = x.x; a[i + 1] = x.y;
      }
      if (KIND == 10 && (i & 1) == 0) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f x = {a[i], a[i + 1]};
        x = x * (v2f){1.0001f, 1.0002f};
        a[i] = x.x; a[i + 1] = x.y;
      }
      if (KIND == 8) { int t = __builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x140, 0xF, 0xF, false); a[i] = a[i] + __int_as_float(t); }
      // a full-wave rotate by one lane (DPP wave_ror:1, gfx9 only): what a systolic walk of pixel states across candidate
      // lanes would pay per state register and step (round 4, DESIGN Appendix A)
      if (KIND == 11) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x13C, 0xF, 0xF, false));
      if (KIND == 12) { int t = __builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x13C, 0xF, 0xF, false); a[i] = __builtin_fmaf(__int_as_float(t), 1.0001f, a[i]); }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, int blocks_per_cu, float* out) {
  int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: blocks_per_cu blocks * 4 waves / 4 SIMDs = blocks_per_cu waves per SIMD
  double instr_per_simd = (double)blocks_per_cu * ITERS * 16;
  double ns_per_instr = ms * 1e6 / instr_per_simd;
  printf("%-22s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz, %.2f @2.0GHz)\n", name, blocks_per_cu, ms, ns_per_instr, ns_per_instr * 2.4, ns_per_instr * 2.0);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int occ : {1, 2, 4, 8}) {
    if (occ == 1 || occ == 8) {
      run<0>("v_fma_f32", occ, out); run<1>("v_mul_f32", occ, out); run<2>("v_add_f32_dpp quad", occ, out);
      run<8>("v_add_f32_dpp rowmirror", occ, out);
      run<3>("v_exp_f32", occ, out); run<4>("v_rcp_f32", occ, out); run<5>("v_permlane32_swap", occ, out);
      run<6>("v_min_f32", occ, out); run<7>("v_cmp+v_cndmask", occ, out);
      run<9>("v_pk_fma_f32 (x0.5 instr)", occ, out); run<10>("v_pk_mul_f32 (x0.5 instr)", occ, out);
      run<11>("v_mov_b32_dpp wave_ror:1", occ, out); run<12>("v_fma_f32_dpp wave_ror:1", occ, out);
    }
  }
  return 0;
}
