// Per-Gaussian stages of the rasterizer: projection / culling / EWA splat / SH colour (forward)
// and the chain rule back to means, scales, rotations and SH coefficients (backward).
//
// Replaces the `preprocess` stages of the un-vendored `diff_gaussian_rasterization` extension
// the reference calls at gaussian_renderer/__init__.py:85-93; the in-tree Python duplicates of
// two sub-steps pin the arithmetic: SH basis utils/sh_utils.py:57-112, covariance
// utils/general_utils.py:78-110 + scene/gaussian_model.py:27-31.
//
// HBM-bound streaming kernels: one lane per Gaussian, consecutive lanes read consecutive
// Gaussians (the [P,3]/[P,4]/[P,M,3] rows of a wave are one contiguous span), every visible
// Gaussian leaves exactly one 64-byte render record.  Compiled with -ffp-contract=off: every
// expression that feeds an integer decision (cull, radius, tile rect) is evaluated with the
// same IEEE operations, in the same order, as oracle/tile_ref.c, so those integers are bit-exact.
#include "b3gs_internal.h"

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

struct Mat16 {
  float m[16];
};

__device__ __forceinline__ Mat16 load_mat(const float* __restrict__ p) {
  Mat16 r;
#pragma unroll
  for (int i = 0; i < 16; i++) r.m[i] = p[i];
  return r;
}
// The same for data that no kernel of the launch writes (camera matrices, camera position) at a wave-uniform address:
// through the constant address space the loads become scalar (s_load_dwordx16: ONE round trip into SGPRs).  As plain
// loads they are vector loads -- the compiler cannot prove that the kernel's own stores leave them alone -- in several
// dependent groups per view of the projection's view loop, holding 35 VGPRs.
typedef const float __attribute__((address_space(4))) cfloat_k;
__device__ __forceinline__ cfloat_k* uniform_ptr(const float* p) { return (cfloat_k*)(uintptr_t)p; }
__device__ __forceinline__ Mat16 load_mat_uniform(const float* p) {
  cfloat_k* q = uniform_ptr(p);
  Mat16 r;
#pragma unroll
  for (int i = 0; i < 16; i++) r.m[i] = q[i];
  return r;
}

__device__ __forceinline__ void quat_to_R(float r, float x, float y, float z, float R[9]) {
  R[0] = 1.f - 2.f * (y * y + z * z);
  R[1] = 2.f * (x * y - r * z);
  R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z);
  R[4] = 1.f - 2.f * (x * x + z * z);
  R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y);
  R[7] = 2.f * (y * z + r * x);
  R[8] = 1.f - 2.f * (x * x + y * y);
}

// RAW mode = the parameter accessors of scene/gaussian_model.py:95-115 fused into the kernels:
// scales = exp(_scaling), rotations = normalize(_rotation) (F.normalize, eps 1e-12),
// opacity = sigmoid(_opacity), features = cat(_features_dc, _features_rest).
template <bool RAW>
__device__ __forceinline__ void load_scale_rot(const SceneX& sx_, int i, float s[3], float q[4], float* inv_norm) {
  if (RAW) {
    const float* ls = sx_.raw.scaling + 3 * (size_t)i;
    s[0] = expf(ls[0]); s[1] = expf(ls[1]); s[2] = expf(ls[2]);
    const float4 r = reinterpret_cast<const float4*>(sx_.raw.rotation)[i];
    // Bit for bit what torch-ROCm's F.normalize(_rotation) hands the reference's rasterizer (scene/gaussian_model.py:99-101):
    // ATen reduces the four squares of a row as a tree, (x0^2 + x1^2) + (x2^2 + x3^2) (four lanes, shuffle-down by 1 then 2),
    // takes the correctly rounded square root, clamps at eps and DIVIDES every element -- multiplying by one reciprocal, or
    // summing left to right, lands one ulp beside it for ~1 quaternion in 4, which moved 0-3 integer radii per view at 1M
    // Gaussians (tests/test_gpu_round5.py::test_in_kernel_activations_are_torch_bits pins all three accessors)
    const float n = sqrtf((r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w));
    const float den = fmaxf(n, 1e-12f);
    q[0] = r.x / den; q[1] = r.y / den; q[2] = r.z / den; q[3] = r.w / den;
    *inv_norm = 1.0f / den;
  } else {
    const float* ps = sx_.sc.scales + 3 * (size_t)i;
    s[0] = ps[0]; s[1] = ps[1]; s[2] = ps[2];
    const float4 r = reinterpret_cast<const float4*>(sx_.sc.rotations)[i];
    q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
    *inv_norm = 1.0f;
  }
}
template <bool RAW>
__device__ __forceinline__ float load_opacity(const SceneX& sx_, int i) {
  if (RAW) return 1.0f / (1.0f + expf(-sx_.raw.opacity[i]));
  return sx_.sc.opacities[i];
}
// SH coefficient k, channel ch: [M][3] rows; in RAW mode row 0 lives in features_dc, rows 1.. in features_rest
struct ShView {
  const float* dc;
  const float* rest;
  __device__ __forceinline__ float operator()(int k, int ch) const { return k == 0 ? dc[ch] : rest[3 * (k - 1) + ch]; }
};
// The same with rows 0..3 (degree <= 1) in this thread's column of an LDS table: the forward projects one Gaussian into
// every view of the batch, and a load inside the view loop is one exposed L2 round trip per view (the compiler cannot hoist
// it past the stores); held in registers across the loop the 12 values push the kernel past 96 VGPRs (measured: spills, or
// one wave per SIMD less: no gain).  Measured on MI355X: 176-187 -> 167-171 us.
struct ShLds {
  const float* lo;     // &table[0][tid], row stride 256 floats
  const float* rest;
  __device__ __forceinline__ float operator()(int k, int ch) const { return k < 4 ? lo[(3 * k + ch) * 256] : rest[3 * (k - 1) + ch]; }
};
template <bool RAW>
__device__ __forceinline__ ShView sh_view(const SceneX& sx_, int i) {
  ShView v;
  if (RAW) {
    v.dc = sx_.raw.features_dc + 3 * (size_t)i;
    v.rest = sx_.raw.features_rest + (size_t)3 * (sx_.sc.M - 1) * i;
  } else {
    v.dc = sx_.sc.shs + (size_t)3 * sx_.sc.M * i;
    v.rest = v.dc + 3;
  }
  return v;
}

template <bool RAW>
__device__ __forceinline__ void cov3d_of(const SceneX& sx_, int i, float c6[6]) {
  const B3gsScene& sc = sx_.sc;
  if (!RAW && sc.cov3D_precomp) {
#pragma unroll
    for (int k = 0; k < 6; k++) c6[k] = sc.cov3D_precomp[6 * (size_t)i + k];
    return;
  }
  float s[3], q[4], inv_norm;
  load_scale_rot<RAW>(sx_, i, s, q, &inv_norm);
  float R[9], L[9];
  quat_to_R(q[0], q[1], q[2], q[3], R);
  float sx = sc.scale_modifier * s[0], sy = sc.scale_modifier * s[1], sz = sc.scale_modifier * s[2];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    L[3 * r + 0] = R[3 * r + 0] * sx;
    L[3 * r + 1] = R[3 * r + 1] * sy;
    L[3 * r + 2] = R[3 * r + 2] * sz;
  }
#define LL(a, b) ((L[3 * a] * L[3 * b] + L[3 * a + 1] * L[3 * b + 1]) + L[3 * a + 2] * L[3 * b + 2])
  c6[0] = LL(0, 0);
  c6[1] = LL(0, 1);
  c6[2] = LL(0, 2);
  c6[3] = LL(1, 1);
  c6[4] = LL(1, 2);
  c6[5] = LL(2, 2);
#undef LL
}

// EWA: the 2x3 matrix T = J * Wr (third row of J is zero) and cov2D = T Sigma T^T
struct Ewa {
  float T0[3], T1[3];
  float a, b, c;        // cov2D without the low-pass term
  float tx, ty, tz;     // clamped view-space position
  float xmul, ymul;     // 0 when the corresponding axis was clamped
};

__device__ __forceinline__ Ewa ewa_project(const float pv[3], float fx, float fy, float tanfovx, float tanfovy,
                                           const float c6[6], const Mat16& vm) {
  Ewa e;
  float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
  float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
  e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  e.tx = fminf(limx, fmaxf(-limx, txtz)) * pv[2];
  e.ty = fminf(limy, fmaxf(-limy, tytz)) * pv[2];
  e.tz = pv[2];
  float J00 = fx / e.tz, J02 = -(fx * e.tx) / (e.tz * e.tz);
  float J11 = fy / e.tz, J12 = -(fy * e.ty) / (e.tz * e.tz);
#pragma unroll
  for (int j = 0; j < 3; j++) {
    e.T0[j] = J00 * vm.m[4 * j + 0] + J02 * vm.m[4 * j + 2];
    e.T1[j] = J11 * vm.m[4 * j + 1] + J12 * vm.m[4 * j + 2];
  }
  float S00 = c6[0], S01 = c6[1], S02 = c6[2], S11 = c6[3], S12 = c6[4], S22 = c6[5];
  float v0x = (S00 * e.T0[0] + S01 * e.T0[1]) + S02 * e.T0[2];
  float v0y = (S01 * e.T0[0] + S11 * e.T0[1]) + S12 * e.T0[2];
  float v0z = (S02 * e.T0[0] + S12 * e.T0[1]) + S22 * e.T0[2];
  float v1x = (S00 * e.T1[0] + S01 * e.T1[1]) + S02 * e.T1[2];
  float v1y = (S01 * e.T1[0] + S11 * e.T1[1]) + S12 * e.T1[2];
  float v1z = (S02 * e.T1[0] + S12 * e.T1[1]) + S22 * e.T1[2];
  e.a = (e.T0[0] * v0x + e.T0[1] * v0y) + e.T0[2] * v0z;
  e.b = (e.T0[0] * v1x + e.T0[1] * v1y) + e.T0[2] * v1z;
  e.c = (e.T1[0] * v1x + e.T1[1] * v1y) + e.T1[2] * v1z;
  return e;
}

__device__ __forceinline__ int clampi_from_float(float v, int hi) {
  return (int)fminf(fmaxf(v, 0.0f), (float)hi);
}

// SH -> RGB for one channel; sh points at coefficient 0 of this Gaussian, stride 3 floats
template <typename SHV>
__device__ __forceinline__ float sh_eval(int deg, const SHV& sh, int ch, float x, float y, float z) {
#define SH(k) sh(k, ch)
  float r = SH_C0 * SH(0);
  if (deg > 0) {
    r = ((r - SH_C1 * y * SH(1)) + SH_C1 * z * SH(2)) - SH_C1 * x * SH(3);
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      r = ((((r + SH_C2[0] * xy * SH(4)) + SH_C2[1] * yz * SH(5)) + SH_C2[2] * (2.f * zz - xx - yy) * SH(6)) +
           SH_C2[3] * xz * SH(7)) +
          SH_C2[4] * (xx - yy) * SH(8);
      if (deg > 2) {
        r = ((((((r + SH_C3[0] * y * (3.f * xx - yy) * SH(9)) + SH_C3[1] * xy * z * SH(10)) +
                SH_C3[2] * y * (4.f * zz - xx - yy) * SH(11)) +
               SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * SH(12)) +
              SH_C3[4] * x * (4.f * zz - xx - yy) * SH(13)) +
             SH_C3[5] * z * (xx - yy) * SH(14)) +
            SH_C3[6] * x * (xx - 3.f * yy) * SH(15);
      }
    }
  }
#undef SH
  return r;
}

#ifdef B3GS_PRE_TRACE   // (tools/pre_trace.py: per-workgroup wall-clock stamps of the projection; never in the product build)
__device__ unsigned long long g_pre_trace[8192][12];
#define PRE_TRACE(slot, val) do { if (threadIdx.x == 0 && blockIdx.x < 8192u) g_pre_trace[blockIdx.x][slot] = (val); } while (0)
extern "C" size_t b3gs_debug_pre_trace(unsigned long long* host) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pre_trace), sizeof(g_pre_trace));
  return sizeof(g_pre_trace) / 8;
}
#else
#define PRE_TRACE(slot, val) do { } while (0)
#endif
constexpr int PRE_PRED_WORDS = 128;   // LDS copy of one view's predicted-open bitmap (8 KB for 8 views)
template <bool RAW>
#ifndef B3GS_PRE_WAVES
#define B3GS_PRE_WAVES 5   /* waves per SIMD the projection must leave room for (<= 96 VGPRs) */
#endif
__global__ void __launch_bounds__(256, B3GS_PRE_WAVES) preprocess_fwd_kernel(PreBatch pb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  PRE_TRACE(0, wall_clock64());
  PRE_TRACE(10, (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));   // HW_ID
  PRE_TRACE(11, (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)));   // XCC_ID
  for (int v = 0; v < pb.n; v++)
    if (i < pb.out[v].ntiles) {   // empty = (max, 0): the tile sort's last pass min/maxes into it
      pb.out[v].ranges[i] = make_uint2(0xFFFFFFFFu, 0u);
      pb.out[v].ranges2[i] = make_uint2(0xFFFFFFFFu, 0u);
    }
  for (int v = 0; v < pb.n; v++)
    if (i < pb.out[v].nrowwords) pb.out[v].open_rows[i] = 0ull;
  // two-round forward: the bitmaps of the tiles predicted open (304 bytes per view at 800x600) are read once per
  // Gaussian and view below -- staged in LDS when they fit
  __shared__ unsigned long long s_pred[B3GS_MAX_FUSED_VIEWS][PRE_PRED_WORDS];
  // ... and, when a tile row is one word (<= 64 x 64 tiles), which rows / columns hold a predicted tile at all: most
  // Gaussians miss both masks and never walk the rows of their rect
  __shared__ unsigned long long s_any[B3GS_MAX_FUSED_VIEWS][2];
  if (pb.out[0].pred_rows) {
    for (int v = 0; v < pb.n; v++) {
      const int nw = pb.out[v].nrowwords;
      unsigned long long wd = 0ull;
      if (nw <= PRE_PRED_WORDS && (int)threadIdx.x < nw) {
        wd = pb.out[v].pred_rows[threadIdx.x];
        s_pred[v][threadIdx.x] = wd;
      }
      if (threadIdx.x < 64u) {
        const int gx = (pb.sc[v].W + B3GS_TILE - 1) / B3GS_TILE, gy = (pb.sc[v].H + B3GS_TILE - 1) / B3GS_TILE;
        unsigned long long rows = ~0ull, cols = ~0ull;
        if (gx <= 64 && gy <= 64 && nw == gy) {   // word y = row y
          rows = __ballot(wd != 0ull);
          uint32_t lo = (uint32_t)wd, hi = (uint32_t)(wd >> 32);
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) { lo |= (uint32_t)__shfl_xor((int)lo, d, 64); hi |= (uint32_t)__shfl_xor((int)hi, d, 64); }
          cols = ((unsigned long long)hi << 32) | lo;
        }
        if (threadIdx.x == 0) { s_any[v][0] = rows; s_any[v][1] = cols; }
      }
    }
    __syncthreads();
  }
  __shared__ float s_sh[12][256];
  if (i >= pb.sc[0].P) return;
  // view-independent part, once per Gaussian: position, 3D covariance (all views of a batch share the
  // scale modifier), activated opacity
  SceneX sx_;
  sx_.sc = pb.sc[0];
  sx_.raw = pb.raw;
  sx_.raw_mode = pb.raw_mode;
  sx_.tight = pb.tight;
  const float* __restrict__ means3D = RAW ? sx_.raw.xyz : sx_.sc.means3D;
  const float px3 = means3D[3 * (size_t)i], py3 = means3D[3 * (size_t)i + 1], pz3 = means3D[3 * (size_t)i + 2];
  float c6[6];
  cov3d_of<RAW>(sx_, i, c6);
  const float op = load_opacity<RAW>(sx_, i);
  const ShView shm = sh_view<RAW>(sx_, i);
  ShLds sh;
  sh.rest = shm.rest;
  sh.lo = &s_sh[0][threadIdx.x];
  if (RAW || !sx_.sc.colors_precomp) {
    s_sh[0][threadIdx.x] = shm.dc[0]; s_sh[1][threadIdx.x] = shm.dc[1]; s_sh[2][threadIdx.x] = shm.dc[2];
    if (sx_.sc.M >= 4) {
#pragma unroll
      for (int k = 0; k < 9; k++) s_sh[3 + k][threadIdx.x] = shm.rest[k];
    }
  }

  uint2 held_rect = make_uint2(0u, 0u);
  uint32_t held_key = 0u;
  PRE_TRACE(1, wall_clock64());
#pragma unroll 1
  for (int v = 0; v < pb.n; v++) {
  PRE_TRACE(2 + v, wall_clock64());
  // the view's descriptors by VALUE: most of their fields are then read from the kernel-argument segment in a few wide scalar
  // loads at the top of the iteration; by reference every use was a scalar load of its own with an `s_waitcnt lgkmcnt(0)` behind
  // it (53 -> 39 such waits in the kernel; 183.7 -> 180.4 us same box, round 5)
  const B3gsScene sc = pb.sc[v];
  const PreOut g = pb.out[v];
  const Mat16 vm = load_mat_uniform(sc.viewmatrix);
  const Mat16 pm = load_mat_uniform(sc.projmatrix);
  cfloat_k* cp = uniform_ptr(sc.campos);
  const float campos[3] = {cp[0], cp[1], cp[2]};

  int32_t radius_out = 0;
  uint32_t touched = 0, dkey = 0xFFFFFFFFu, clamp_bits = 0;
  uint2 rect = make_uint2(0, 0);

  float pv[3];
  pv[0] = ((vm.m[0] * px3 + vm.m[4] * py3) + vm.m[8] * pz3) + vm.m[12];
  pv[1] = ((vm.m[1] * px3 + vm.m[5] * py3) + vm.m[9] * pz3) + vm.m[13];
  pv[2] = ((vm.m[2] * px3 + vm.m[6] * py3) + vm.m[10] * pz3) + vm.m[14];

  if (sc.prefiltered || !(pv[2] <= B3GS_NEAR)) {  // same predicate form as the oracle (NaN passes)
    // the key depends on view-space z alone (not on the screen-space culls below, whose Gaussians emit no
    // instances anyway), so two views with the same z row of the view matrix produce the same depth order
    dkey = __float_as_uint(pv[2]);
    float hx = ((pm.m[0] * px3 + pm.m[4] * py3) + pm.m[8] * pz3) + pm.m[12];
    float hy = ((pm.m[1] * px3 + pm.m[5] * py3) + pm.m[9] * pz3) + pm.m[13];
    float hw = ((pm.m[3] * px3 + pm.m[7] * py3) + pm.m[11] * pz3) + pm.m[15];
    float pw = 1.0f / (hw + 0.0000001f);
    float ppx = hx * pw, ppy = hy * pw;

    const float fx = (float)sc.W / (2.0f * sc.tan_fovx), fy = (float)sc.H / (2.0f * sc.tan_fovy);
    Ewa e = ewa_project(pv, fx, fy, sc.tan_fovx, sc.tan_fovy, c6, vm);
    float a = e.a + 0.3f, b = e.b, c = e.c + 0.3f;
    float det = a * c - b * b;
    if (det != 0.0f) {
      float det_inv = 1.0f / det;
      float cxx = c * det_inv, cxy = -b * det_inv, cyy = a * det_inv;
      float mid = 0.5f * (a + c);
      float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
      float l1 = mid + disc, l2 = mid - disc;
      float rad_f = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
      float mx = ((ppx + 1.0f) * (float)sc.W - 1.0f) * 0.5f;
      float my = ((ppy + 1.0f) * (float)sc.H - 1.0f) * 0.5f;
      const int gx = (sc.W + B3GS_TILE - 1) / B3GS_TILE, gy = (sc.H + B3GS_TILE - 1) / B3GS_TILE;
      int x0 = clampi_from_float((mx - rad_f) / (float)B3GS_TILE, gx);
      int y0 = clampi_from_float((my - rad_f) / (float)B3GS_TILE, gy);
      int x1 = clampi_from_float((mx + rad_f + (float)(B3GS_TILE - 1)) / (float)B3GS_TILE, gx);
      int y1 = clampi_from_float((my + rad_f + (float)(B3GS_TILE - 1)) / (float)B3GS_TILE, gy);
      int area = (x1 - x0) * (y1 - y0);
      if (area != 0) {
        float rgb[3];
        if (!RAW && sc.colors_precomp) {
          rgb[0] = sc.colors_precomp[3 * (size_t)i];
          rgb[1] = sc.colors_precomp[3 * (size_t)i + 1];
          rgb[2] = sc.colors_precomp[3 * (size_t)i + 2];
        } else {
          float dx = px3 - campos[0], dy = py3 - campos[1], dz = pz3 - campos[2];
          float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
          dx = dx * inv; dy = dy * inv; dz = dz * inv;
#pragma unroll
          for (int ch = 0; ch < 3; ch++) {
            float v = sh_eval(sc.D, sh, ch, dx, dy, dz) + 0.5f;
            if (v < 0.0f) clamp_bits |= (1u << ch);
            rgb[ch] = fmaxf(v, 0.0f);
          }
        }
        // conservative half-extents of the region where op*G >= 1/255 (G <= 1): used by the blend
        // kernels to skip whole 8x8 pixel quadrants; never changes which pixels contribute
        float ext_x = -1.0e30f, ext_y = -1.0e30f;
        if (op >= 0.0039f) {
          float tau2 = 2.0f * logf(fmaxf(255.0f * op, 1.0f));
          ext_x = sqrtf(tau2 * a) * 1.001f + 0.05f;
          ext_y = sqrtf(tau2 * c) * 1.001f + 0.05f;
        }
        if (sx_.tight) {
          // Tight binning (fused path): drop the tiles of the reference's 3-sigma square that the
          // alpha >= 1/255 footprint cannot reach.  Tile t holds pixel centres [16t, 16t+15]; a
          // dropped tile has no pixel inside the (conservative) footprint, so every one of its
          // instances would have been skipped by the alpha < 1/255 rule: images and gradients are
          // unchanged, only N and the lists shrink (order preserved).
          const float T = (float)B3GS_TILE;
          const int tx0 = clampi_from_float(ceilf((mx - ext_x - (T - 1.0f)) / T), gx);
          const int tx1 = clampi_from_float(floorf((mx + ext_x) / T) + 1.0f, gx);
          const int ty0 = clampi_from_float(ceilf((my - ext_y - (T - 1.0f)) / T), gy);
          const int ty1 = clampi_from_float(floorf((my + ext_y) / T) + 1.0f, gy);
          x0 = max(x0, tx0); x1 = max(x0, min(x1, tx1));
          y0 = max(y0, ty0); y1 = max(y0, min(y1, ty1));
          if (ext_x < 0.0f) x1 = x0;  // opacity below 1/255: no tiles (rect area == tiles_touched, binning relies on it)
          area = (x1 - x0) * (y1 - y0);
        }
        float4* rec = g.rec + 4 * (size_t)i;
        rec[0] = make_float4(mx, my, cxx, cxy);
        rec[1] = make_float4(cyy, op, rgb[0], rgb[1]);
        rec[2] = make_float4(rgb[2], pv[2], ext_x, ext_y);
        // The fourth 16 bytes of the 64-byte record are padding nobody reads -- and they are written all the same (round 5):
        // a slot written 48 bytes out of 64 is a masked (read-modify-write) access at the memory side, a whole 64-byte slot
        // is not.  Same-box A/B at the headline: projection 240 -> 191 us per iteration (-20 %), 594-599 -> 610-613 iters/s,
        // for 16 more bytes per visible Gaussian and view.  (A 48-byte store was `-DB3GS_PRE_PARTIAL_RECORD` until the fourth part got a use.)
        // tools/ubench/slot_store.hip, 6M slots: 48 of 64 bytes 156 us, these four 16-byte stores per lane 112 us, the
        // quarters of a quad's four records transposed across its lanes (DPP) so that every instruction writes whole lines
        // 64 us (a dense stream: 63).  The last step was built here too (64 VALU instructions per view, bit-identical
        // records) and changes nothing -- 183.4 against 184.4 us: spread over this kernel's 180 us the stores are far from
        // either rate; its time is the dependent chain per view at 5 waves per SIMD.  Not kept.
        // Since round 6 its first word carries the SH clamp bits (bit c: colour channel c clamped at 0), which only the chain
        // rule of the ~2 % touched (Gaussian, view) pairs reads: they were a 4-byte store stream of their own before.
        rec[3] = make_float4(__uint_as_float(clamp_bits), 0.f, 0.f, 0.f);
        radius_out = (int32_t)fminf(rad_f, 2147483520.0f);
        touched = (uint32_t)area;
        rect = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16));
      }
    }
  }
  g.radii[i] = radius_out;
  if (g.visible) g.visible[i] = radius_out > 0 ? 1 : 0;
  g.tiles_touched[i] = touched;
  g.depth_key[i] = dkey;
  // the depth order of an earlier forward is adopted only while every key equals the key that forward sorted
  if (g.hint_key && g.hint_key[i] != dkey) {
    *g.hint_word = 1;
    if (g.hint_fatal) atomicOr(g.hint_fatal, 8);   // trusted hint: no sort was launched for this view
  }
  // the three-pass depth sort assumes every visible key within 2^27 of the float bits of the near plane (binning.hip):
  // say so when one is not (z > ~13107, or a prefiltered Gaussian in front of the near plane)
  if (g.span_flag && dkey != 0xFFFFFFFFu && (dkey <= 0x3E4CCCCDu || dkey - 0x3E4CCCCDu >= (1u << 27) - 2u))
    atomicOr(g.span_flag, 2);
  if (g.pred_rows) {
    // two-round forward: does this Gaussian reach a tile that is predicted open?  One bit per Gaussian (a wave's 64
    // consecutive Gaussians = one word): behind segment 1 the scan gathers a rect only where the bit is set
    bool hit = false;
    if (touched != 0u) {
      const uint32_t rx0 = rect.x & 0xFFFFu, ry0 = rect.x >> 16, rx1 = rect.y & 0xFFFFu, ry1 = rect.y >> 16;
      // rows / columns of the rect as bit masks (exact when the grid fits 64 x 64; otherwise the "any" masks are all ones)
      const unsigned long long rm = (ry1 - ry0 >= 64u ? ~0ull : ((1ull << (ry1 - ry0)) - 1ull)) << (ry0 & 63u);
      const unsigned long long cm = (rx1 - rx0 >= 64u ? ~0ull : ((1ull << (rx1 - rx0)) - 1ull)) << (rx0 & 63u);
      const bool exact = s_any[v][0] != ~0ull || s_any[v][1] != ~0ull;
      if (!exact || ((s_any[v][0] & rm) != 0ull && (s_any[v][1] & cm) != 0ull)) {
        // two copies of the walk, one per address space: through ONE generic pointer the loads are `flat_load` +
        // `s_waitcnt vmcnt(0)`, and on gfx9 that counter also holds this wave's record stores of the view (measured: no
        // difference at the headline's sizes -- few lanes get here -- but the wait had no reason to exist)
        if (g.nrowwords <= PRE_PRED_WORDS) hit = open_tiles(rect, open_map(&s_pred[v][0], sc.W, sc.H)) != 0u;
        else hit = open_tiles(rect, open_map(g.pred_rows, sc.W, sc.H)) != 0u;
      }
    }
    const unsigned long long word = __ballot(hit);
    if ((threadIdx.x & 63u) == 0u) g.pflag[(size_t)i >> 6] = word;
  }
  // a pair that shares ONE depth order must have equal keys (same z row of the view matrix): checked when the caller asks
  if (g.rect_role == 2 && g.pair_fatal && dkey != held_key) atomicOr(g.pair_fatal, 8);
  held_key = dkey;
  if (g.rect_role == 1) held_rect = rect;
  else if (g.rect_role == 2) reinterpret_cast<uint4*>(g.rect - 1)[i] = make_uint4(held_rect.x, held_rect.y, rect.x, rect.y);
  else g.rect[(size_t)i * g.rect_stride] = rect;
  }
  PRE_TRACE(9, wall_clock64());
}

// ------------------------------------------------------------------------------------------
// backward: one lane per Gaussian.
// ------------------------------------------------------------------------------------------
// the fp32 sums the blend backward accumulated for one Gaussian in one view
struct PixSums {
  float g2x, g2y;       // d/d(pixel position), already scaled by 0.5*W, 0.5*H
  float gxx, gxy, gyy;  // d/d(conic); gxy is HALF the true xy gradient
  float gdepth, gcol[3], gop;
};

// Raw-mode scratch row (B3GS_SCRATCH_ROW = 10 floats, 40 bytes, no padding: 8-byte aligned, five float2): conic xx,
// xy, yy, depth | mean2D x, y | colour r, g, b | opacity.  Returns the sums and leaves the row zero (the scratch is persistent: no per-view memset).
__device__ __forceinline__ PixSums load_scratch_row(float* scratch, int i, bool* touched = nullptr) {
  static_assert(B3GS_SCRATCH_ROW == 10, "five float2 per row");
  float2* row = reinterpret_cast<float2*>(scratch) + 5 * (size_t)i;
  const float2 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3], r4 = row[4];
  // sums are never -0.0 (atomic adds onto +0.0), so the bit pattern tells "nothing arrived"
  const uint32_t any = ((__float_as_uint(r0.x) | __float_as_uint(r0.y)) | (__float_as_uint(r1.x) | __float_as_uint(r1.y))) |
                       ((__float_as_uint(r2.x) | __float_as_uint(r2.y)) | (__float_as_uint(r3.x) | __float_as_uint(r3.y))) |
                       (__float_as_uint(r4.x) | __float_as_uint(r4.y));
  if (touched) *touched = any != 0u;
  if (!touched || any != 0u) {
    const float2 z = make_float2(0.f, 0.f);
    row[0] = z; row[1] = z; row[2] = z; row[3] = z; row[4] = z;
  }
  PixSums in;
  in.gxx = r0.x; in.gxy = r0.y; in.gyy = r1.x; in.gdepth = r1.y;
  in.g2x = r2.x; in.g2y = r2.y;
  in.gcol[0] = r3.x; in.gcol[1] = r3.y; in.gcol[2] = r4.x;
  in.gop = r4.y;
  return in;
}
// gradients w.r.t. the (activated) rasterizer inputs of one Gaussian
struct GaussGrad {
  float dmean[3], dS[6], dscale[3];
  float4 dq;
  float sraw[3], q[4], inv_norm;  // activated scale / rotation and 1/|raw quaternion| (RAW chain rule)
};
// where SH-coefficient gradients go: plain store (standard API), += in memory (one view, RAW),
// or += in registers for the first REG coefficients (several views fused into one kernel)
struct ShStore {
  float* dc; float* rest;
  __device__ __forceinline__ void put(int k, int ch, float v) { (k == 0 ? dc + ch : rest + 3 * (k - 1) + ch)[0] = v; }
};
struct ShAddMem {
  float* dc; float* rest;
  __device__ __forceinline__ void put(int k, int ch, float v) { (k == 0 ? dc + ch : rest + 3 * (k - 1) + ch)[0] += v; }
};
template <int REG>
struct ShAddReg {
  float acc[3 * REG];
  float* dc; float* rest;
  __device__ __forceinline__ void put(int k, int ch, float v) {
    if (k < REG) acc[3 * k + ch] += v;   // k, ch are compile-time constants at every call site
    else rest[3 * (k - 1) + ch] += v;
  }
};

// Chain rule from the per-view sums to mean / covariance / scale / rotation / SH of Gaussian i.
// Shared by the standard backward, the RAW single-view backward and the fused multi-view backward.
template <bool RAW, class ShSink>
__device__ __forceinline__ void gaussian_backward(const SceneX& sx_, const Mat16& vm, const Mat16& pm, int i,
                                                  uint32_t clamp_bits, const PixSums& in, bool do_sh, bool do_sr,
                                                  GaussGrad& o, ShSink& sink) {
  const B3gsScene& sc = sx_.sc;
  const size_t i3 = 3 * (size_t)i;
  const float* __restrict__ means3D = RAW ? sx_.raw.xyz : sc.means3D;
  const float mx3 = means3D[i3], my3 = means3D[i3 + 1], mz3 = means3D[i3 + 2];
  const float g2x = in.g2x, g2y = in.g2y, gxx = in.gxx, gxy = in.gxy, gyy = in.gyy, gdepth = in.gdepth;

  float pv[3];
  pv[0] = ((vm.m[0] * mx3 + vm.m[4] * my3) + vm.m[8] * mz3) + vm.m[12];
  pv[1] = ((vm.m[1] * mx3 + vm.m[5] * my3) + vm.m[9] * mz3) + vm.m[13];
  pv[2] = ((vm.m[2] * mx3 + vm.m[6] * my3) + vm.m[10] * mz3) + vm.m[14];

  float c6[6];
  cov3d_of<RAW>(sx_, i, c6);
  const float fx = (float)sc.W / (2.0f * sc.tan_fovx), fy = (float)sc.H / (2.0f * sc.tan_fovy);
  const Ewa e = ewa_project(pv, fx, fy, sc.tan_fovx, sc.tan_fovy, c6, vm);
  const float a = e.a + 0.3f, b = e.b, c = e.c + 0.3f;

  // conic = (c, -b, a)/den  ->  gradient w.r.t. (a, b, c)
  const float den = a * c - b * b;
  const float k2 = 1.0f / (den * den + 0.0000001f);
  const float dL_da = k2 * (-c * c * gxx + 2.f * b * c * gxy + (den - a * c) * gyy);
  const float dL_dc = k2 * (-a * a * gyy + 2.f * a * b * gxy + (den - a * c) * gxx);
  const float dL_db = k2 * 2.f * (b * c * gxx - (den + 2.f * b * b) * gxy + a * b * gyy);

  const float* T0 = e.T0;
  const float* T1 = e.T1;
  float* dS = o.dS;
  dS[0] = T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc;
  dS[3] = T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc;
  dS[5] = T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc;
  dS[1] = 2.f * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2.f * T1[0] * T1[1] * dL_dc;
  dS[2] = 2.f * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2.f * T1[0] * T1[2] * dL_dc;
  dS[4] = 2.f * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2.f * T1[1] * T1[2] * dL_dc;

  // dL/dT, then through T = J Wr to the view-space position
  const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
  float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    float st0 = S[3 * r] * T0[0] + S[3 * r + 1] * T0[1] + S[3 * r + 2] * T0[2];
    float st1 = S[3 * r] * T1[0] + S[3 * r + 1] * T1[1] + S[3 * r + 2] * T1[2];
    float dT0 = 2.f * st0 * dL_da + st1 * dL_db;
    float dT1 = 2.f * st1 * dL_dc + st0 * dL_db;
    // Wr[row][r] = vm[4*r + row]
    dJ00 += vm.m[4 * r + 0] * dT0;
    dJ02 += vm.m[4 * r + 2] * dT0;
    dJ11 += vm.m[4 * r + 1] * dT1;
    dJ12 += vm.m[4 * r + 2] * dT1;
  }
  const float tz = 1.f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
  const float dtx = e.xmul * -fx * tz2 * dJ02;
  const float dty = e.ymul * -fy * tz2 * dJ12;
  const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * e.tx) * tz3 * dJ02 + (2.f * fy * e.ty) * tz3 * dJ12;
  float* dmean = o.dmean;
#pragma unroll
  for (int k = 0; k < 3; k++) dmean[k] = vm.m[4 * k + 0] * dtx + vm.m[4 * k + 1] * dty + vm.m[4 * k + 2] * dtz;

  // pixel position -> mean (perspective divide); g2x/g2y already carry the 0.5*W / 0.5*H factor
  {
    float hx = ((pm.m[0] * mx3 + pm.m[4] * my3) + pm.m[8] * mz3) + pm.m[12];
    float hy = ((pm.m[1] * mx3 + pm.m[5] * my3) + pm.m[9] * mz3) + pm.m[13];
    float hw = ((pm.m[3] * mx3 + pm.m[7] * my3) + pm.m[11] * mz3) + pm.m[15];
    float mw = 1.0f / (hw + 0.0000001f);
    float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
#pragma unroll
    for (int k = 0; k < 3; k++)
      dmean[k] += (pm.m[4 * k + 0] * mw - pm.m[4 * k + 3] * mul1) * g2x + (pm.m[4 * k + 1] * mw - pm.m[4 * k + 3] * mul2) * g2y;
  }
  // depth = view z
#pragma unroll
  for (int k = 0; k < 3; k++) dmean[k] += vm.m[4 * k + 2] * gdepth;

  // colour: SH coefficients and view direction
  if (do_sh) {
    const ShView sh = sh_view<RAW>(sx_, i);
    cfloat_k* cp = uniform_ptr(sc.campos);   // (wave-uniform, read-only: scalar loads)
    float ddx = mx3 - cp[0], ddy = my3 - cp[1], ddz = mz3 - cp[2];
    float inv = 1.0f / sqrtf((ddx * ddx + ddy * ddy) + ddz * ddz);
    const float x = ddx * inv, y = ddy * inv, z = ddz * inv;
    float gdir[3] = {0.f, 0.f, 0.f};
    const int deg = sc.D;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      const float gl = ((clamp_bits >> ch) & 1u) ? 0.f : in.gcol[ch];
#define SH(k) sh(k, ch)
#define DSH(k, v) sink.put(k, ch, v)
      DSH(0, SH_C0 * gl);
      float rx = 0.f, ry = 0.f, rz = 0.f;
      if (deg > 0) {
        DSH(1, -SH_C1 * y * gl);
        DSH(2, SH_C1 * z * gl);
        DSH(3, -SH_C1 * x * gl);
        rx = -SH_C1 * SH(3);
        ry = -SH_C1 * SH(1);
        rz = SH_C1 * SH(2);
        if (deg > 1) {
          float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          DSH(4, SH_C2[0] * xy * gl);
          DSH(5, SH_C2[1] * yz * gl);
          DSH(6, SH_C2[2] * (2.f * zz - xx - yy) * gl);
          DSH(7, SH_C2[3] * xz * gl);
          DSH(8, SH_C2[4] * (xx - yy) * gl);
          rx += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
          ry += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
          rz += SH_C2[1] * y * SH(5) + SH_C2[2] * 4.f * z * SH(6) + SH_C2[3] * x * SH(7);
          if (deg > 2) {
            DSH(9, SH_C3[0] * y * (3.f * xx - yy) * gl);
            DSH(10, SH_C3[1] * xy * z * gl);
            DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy) * gl);
            DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * gl);
            DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy) * gl);
            DSH(14, SH_C3[5] * z * (xx - yy) * gl);
            DSH(15, SH_C3[6] * x * (xx - 3.f * yy) * gl);
            rx += SH_C3[0] * SH(9) * 6.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                  SH_C3[3] * SH(12) * -6.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                  SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy);
            ry += SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                  SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -6.f * yz +
                  SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz + SH_C3[6] * SH(15) * -6.f * xy;
            rz += SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 8.f * yz + SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                  SH_C3[4] * SH(13) * 8.f * xz + SH_C3[5] * SH(14) * (xx - yy);
          }
        }
      }
#undef SH
#undef DSH
      gdir[0] += rx * gl;
      gdir[1] += ry * gl;
      gdir[2] += rz * gl;
    }
    // through n = d/|d|:  (I - n n^T)/|d|
    float dot = gdir[0] * x + gdir[1] * y + gdir[2] * z;
    dmean[0] += (gdir[0] - x * dot) * inv;
    dmean[1] += (gdir[1] - y * dot) * inv;
    dmean[2] += (gdir[2] - z * dot) * inv;
  }

  // Sigma = L L^T, L = R diag(mod*s)
  if (do_sr) {
    load_scale_rot<RAW>(sx_, i, o.sraw, o.q, &o.inv_norm);
    const float r = o.q[0], x = o.q[1], y = o.q[2], z = o.q[3];
    float R[9];
    quat_to_R(r, x, y, z, R);
    const float sv[3] = {sc.scale_modifier * o.sraw[0], sc.scale_modifier * o.sraw[1], sc.scale_modifier * o.sraw[2]};
    const float G[9] = {dS[0], 0.5f * dS[1], 0.5f * dS[2], 0.5f * dS[1], dS[3], 0.5f * dS[4], 0.5f * dS[2], 0.5f * dS[4], dS[5]};
    float dR[9];
#pragma unroll
    for (int bcol = 0; bcol < 3; bcol++) {
      float ds = 0.f;
#pragma unroll
      for (int arow = 0; arow < 3; arow++) {
        // dL/dL[a][b] = 2 * sum_k G[a][k] L[k][b],  L[k][b] = R[k][b] sv[b]
        float dLab = 2.f * sv[bcol] * (G[3 * arow] * R[bcol] + G[3 * arow + 1] * R[3 + bcol] + G[3 * arow + 2] * R[6 + bcol]);
        ds += dLab * R[3 * arow + bcol];
        dR[3 * arow + bcol] = dLab * sv[bcol];
      }
      o.dscale[bcol] = ds * sc.scale_modifier;
    }
    o.dq.x = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    o.dq.y = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
    o.dq.z = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
    o.dq.w = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
  }
}

// RAW chain rule of the fused activations: scales = exp(s) -> d/ds = scale; q = v/|v| ->
// d/dv = (dq - q (q.dq))/|v|; opacity = sigmoid(o) -> d/do = op (1 - op)
__device__ __forceinline__ void raw_chain(const GaussGrad& gg, float dscaling[3], float4& drot) {
  dscaling[0] = gg.dscale[0] * gg.sraw[0];
  dscaling[1] = gg.dscale[1] * gg.sraw[1];
  dscaling[2] = gg.dscale[2] * gg.sraw[2];
  const float qd = ((gg.q[0] * gg.dq.x + gg.q[1] * gg.dq.y) + gg.q[2] * gg.dq.z) + gg.q[3] * gg.dq.w;
  drot.x = (gg.dq.x - gg.q[0] * qd) * gg.inv_norm;
  drot.y = (gg.dq.y - gg.q[1] * qd) * gg.inv_norm;
  drot.z = (gg.dq.z - gg.q[2] * qd) * gg.inv_norm;
  drot.w = (gg.dq.w - gg.q[3] * qd) * gg.inv_norm;
}

// Per-view backward.  Inputs are the fp32 sums the blend backward accumulated:
//   dL_dmeans2D[i] = (gx, gy, 0)   dL_dcolors[i]   dL_dopacity[i]
//   dL_dcov3D[i]   = (g_conic_xx, g_conic_xy(half), g_conic_yy, g_depth, -, -)   (scratch use)
// Standard mode: every output row is overwritten with its final value (culled rows = 0).
// RAW mode: the four arrays above are a caller-owned scratch that is read and reset to zero here
// (so it is clean for the next view without a memset), the chain rule continues through the
// fused activations and the results are ACCUMULATED (+=) into the parameter-shaped gradient
// buffers `rg` -- one owner thread per Gaussian, so plain read-modify-write; culled Gaussians
// touch nothing.  `m2d_out` (optional, [P,3]) receives the screen-space mean gradient.
template <bool RAW>
__global__ void __launch_bounds__(256)
    preprocess_bwd_kernel(SceneX sx_, GeomView g, const int32_t* __restrict__ radii, float* __restrict__ dL_dmeans2D,
                          float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
                          float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
                          float* __restrict__ dL_dscales, float* __restrict__ dL_drots, B3gsRawGrads rg,
                          float* __restrict__ m2d_out) {
  const B3gsScene& sc = sx_.sc;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sc.P) return;
  const size_t i3 = 3 * (size_t)i;
  if (radii[i] <= 0) {
    if (RAW) {
      if (m2d_out) { m2d_out[i3] = 0.f; m2d_out[i3 + 1] = 0.f; m2d_out[i3 + 2] = 0.f; }
      return;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { dL_dmeans2D[i3 + k] = 0.f; dL_dcolors[i3 + k] = 0.f; dL_dmeans3D[i3 + k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = 0.f;
    dL_dopacity[i] = 0.f;
    if (dL_dsh) for (int k = 0; k < 3 * sc.M; k++) dL_dsh[(size_t)3 * sc.M * i + k] = 0.f;
    if (dL_dscales) { dL_dscales[i3] = 0.f; dL_dscales[i3 + 1] = 0.f; dL_dscales[i3 + 2] = 0.f; }
    if (dL_drots) reinterpret_cast<float4*>(dL_drots)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const Mat16 vm = load_mat_uniform(sc.viewmatrix);
  const Mat16 pm = load_mat_uniform(sc.projmatrix);

  PixSums in;
  if (RAW) {
    // dL_dcov3D is the raw-mode scratch: one row per Gaussian (read, and left clean for the next view)
    bool touched;
    in = load_scratch_row(dL_dcov3D, i, &touched);      // (an untouched row is not written back either)
    if (m2d_out) { m2d_out[i3] = in.g2x; m2d_out[i3 + 1] = in.g2y; m2d_out[i3 + 2] = 0.f; }
  } else {
    // the blend backward left this Gaussian's ten sums in one row of the geometry buffer's scratch region (zeroed
    // by b3gs_backward before the launch); the three accumulated outputs of the reference API are written here
    const float2* row = reinterpret_cast<const float2*>(g.bwd_rows) + 5 * (size_t)i;
    const float2 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3], r4 = row[4];
    in.gxx = r0.x; in.gxy = r0.y; in.gyy = r1.x; in.gdepth = r1.y;
    in.g2x = r2.x; in.g2y = r2.y;
    in.gcol[0] = r3.x; in.gcol[1] = r3.y; in.gcol[2] = r4.x;
    in.gop = r4.y;
    dL_dmeans2D[i3] = in.g2x; dL_dmeans2D[i3 + 1] = in.g2y; dL_dmeans2D[i3 + 2] = 0.f;
    dL_dcolors[i3] = in.gcol[0]; dL_dcolors[i3 + 1] = in.gcol[1]; dL_dcolors[i3 + 2] = in.gcol[2];
    dL_dopacity[i] = in.gop;
  }
  // Round 6: ~92 % of the VISIBLE Gaussians of a view lie behind the saturation point of every pixel they cover and receive
  // nothing from the blend backward (the multi-view pass above has always skipped them).  Sums are never -0.0 (atomic adds onto
  // +0.0): an all-zero bit pattern means "nothing arrived", and the chain rule of zero sums is zero -- stored, not computed
  // (tools/module_surface_profile.py: this kernel was 100 us per view at 1M Gaussians, 0.6 of the module surface's 5.6 ms).
  {
    const uint32_t any = ((__float_as_uint(in.gxx) | __float_as_uint(in.gxy)) | (__float_as_uint(in.gyy) | __float_as_uint(in.gdepth))) |
                         ((__float_as_uint(in.g2x) | __float_as_uint(in.g2y)) | (__float_as_uint(in.gcol[0]) | __float_as_uint(in.gcol[1]))) |
                         (__float_as_uint(in.gcol[2]) | __float_as_uint(in.gop));
    if (any == 0u) {
      if (RAW) return;      // accumulate mode: nothing to add (the row is zero already: nothing to reset either)
#pragma unroll
      for (int k = 0; k < 3; k++) dL_dmeans3D[i3 + k] = 0.f;
#pragma unroll
      for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = 0.f;
      if (dL_dsh) for (int k = 0; k < 3 * sc.M; k++) dL_dsh[(size_t)3 * sc.M * i + k] = 0.f;
      if (dL_dscales) { dL_dscales[i3] = 0.f; dL_dscales[i3 + 1] = 0.f; dL_dscales[i3 + 2] = 0.f; }
      if (dL_drots) reinterpret_cast<float4*>(dL_drots)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
  }

  GaussGrad gg;
  const uint32_t cb = __float_as_uint(reinterpret_cast<const float*>(g.rec + 4 * (size_t)i + 3)[0]);   // clamp bits: record word 12
  if (RAW) {
    ShAddMem sink{rg.features_dc + i3, rg.features_rest + (size_t)3 * (sc.M - 1) * i};
    gaussian_backward<true>(sx_, vm, pm, i, cb, in, true, true, gg, sink);
    rg.xyz[i3] += gg.dmean[0];
    rg.xyz[i3 + 1] += gg.dmean[1];
    rg.xyz[i3 + 2] += gg.dmean[2];
    const float op = load_opacity<true>(sx_, i);
    rg.opacity[i] += in.gop * op * (1.0f - op);
    float dsc[3];
    float4 drot;
    raw_chain(gg, dsc, drot);
    rg.scaling[i3] += dsc[0];
    rg.scaling[i3 + 1] += dsc[1];
    rg.scaling[i3 + 2] += dsc[2];
    float4* dst = reinterpret_cast<float4*>(rg.rotation) + i;
    float4 cur = *dst;
    cur.x += drot.x; cur.y += drot.y; cur.z += drot.z; cur.w += drot.w;
    *dst = cur;
  } else {
    const bool do_sh = !sc.colors_precomp && dL_dsh;
    const bool do_sr = !sc.cov3D_precomp && dL_dscales && dL_drots;
    float* dsh = do_sh ? dL_dsh + (size_t)3 * sc.M * i : nullptr;
    if (do_sh) {
      const int nb = (sc.D + 1) * (sc.D + 1);
      for (int k = nb; k < sc.M; k++) { dsh[3 * k] = 0.f; dsh[3 * k + 1] = 0.f; dsh[3 * k + 2] = 0.f; }
    }
    ShStore sink{dsh, dsh ? dsh + 3 : nullptr};
    gaussian_backward<false>(sx_, vm, pm, i, cb, in, do_sh, do_sr, gg, sink);
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = gg.dS[k];
    dL_dmeans3D[i3] = gg.dmean[0];
    dL_dmeans3D[i3 + 1] = gg.dmean[1];
    dL_dmeans3D[i3 + 2] = gg.dmean[2];
    if (do_sr) {
      dL_dscales[i3] = gg.dscale[0];
      dL_dscales[i3 + 1] = gg.dscale[1];
      dL_dscales[i3 + 2] = gg.dscale[2];
      reinterpret_cast<float4*>(dL_drots)[i] = gg.dq;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fused multi-view backward (RAW): ONE pass over the Gaussians for all views of an iteration.
// The per-view kernel above moves ~400 B per visible Gaussian per view (parameters read, gradient
// read-modify-write); here the parameters are read once, every view contributes its 11 sums
// (read + reset), the gradients live in registers across the view loop and are written once:
// ~130 B per Gaussian per view at 6 views, no ordering between views, no zero-fill of the
// gradient buffers (`overwrite`).  One lane per Gaussian, the view loop is wave-uniform.
// ------------------------------------------------------------------------------------------
struct MultiViews {
  int n;
  B3gsViewRef v[B3GS_MAX_FUSED_VIEWS];
};

#ifndef B3GS_ACC_WAVES
#define B3GS_ACC_WAVES 3   /* the pair-chunk kernel below needs 128 VGPRs (no spills): the hardware runs it at 4 waves per SIMD */
#endif
// Most visible Gaussians of a view lie behind the saturation point of every pixel they cover: only ~8 % of the
// (Gaussian, view) rows receive anything from the blend backward, ~20 % of the Gaussians have at least one such view.
// With lane = Gaussian and a wave-uniform view loop the chain rule ran with ~5 of 64 lanes active (PMC:
// SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = 4.9) while every wave still executed it for every view.  So the pass is
// split inside one workgroup of 256 threads / ACC_BLOCK = 1024 Gaussians:
//   phase 1 (streaming, all lanes): read every view's radius + scratch row (coalesced), update the densification
//           statistics, write the optional screen-space gradients, store zero gradients for untouched Gaussians
//           (overwrite mode), and append (local index | touched-view mask) of the others to an LDS list -- wave ballot +
//           one LDS atomic per wave;
//   phase 2 (compute, compacted): lane = list entry; the view loop runs only over the entry's touched views
//           (row read + reset, chain rule, 23 gradients in registers), gradients written once.
// (Round 5, measured and dropped: phase 1 at 131 MB in 69 us is neither a latency chain nor bound by cache-line look-ups --
// all views' radii, then all views' rows in flight together (no control flow around the loads): 68.5 against 70.9 us; the
// rows of a wave read as 160 consecutive 16-byte pieces and handed out through LDS, 40 line look-ups per (wave, view)
// instead of 200: 78.9 us.  Probes: without the row reads 29.9 us (radii, statistics, list); without the statistics 62.7;
// with ONE 8-byte read per row instead of five 69.7 -- the rows cost what their LINES cost, and with half of the Gaussians
// visible at random nearly every line of the 6 x 40 MB is fetched: the pass already streams.  What would shrink it is
// knowing the ~8 % touched rows without reading the others -- a bit per row set by the blend backward, one more atomic per
// flush in the kernel that is half of the iteration; not built.)
// ACC_PER_THREAD Gaussians per thread = 1024 per workgroup for big calls (1M Gaussians: 977 workgroups); calls over fewer
// Gaussians -- a 100k scene, one range of the pipelined data-parallel tail -- take ONE per thread: with 1024 per workgroup
// they were a quarter of a workgroup per CU and pure latency (100k Gaussians, 2 views: 76 us in the chain-rule kernel;
// a 250k-Gaussian range of the headline scene: 86 us per call whatever the range held)
template <int ACC_PER_THREAD>
__global__ void __launch_bounds__(256, ACC_PER_THREAD >= 4 ? 4 : 8)
    accumulate_scan_kernel(B3gsScene base, MultiViews mv, B3gsRawGrads rg, int overwrite, int first, int count,
                           B3gsDensifyStats ds, uint32_t* __restrict__ g_list, uint32_t* __restrict__ g_count) {
  constexpr int ACC_BLOCK = 256 * ACC_PER_THREAD;
  __shared__ uint32_t s_list[ACC_BLOCK];
  __shared__ uint32_t s_count;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  // Gaussians [first, first + count): every pointer is indexed by the GLOBAL Gaussian index
  const int blk_first = first + (int)blockIdx.x * ACC_BLOCK;
  const int end = first + count;
  const unsigned lane = threadIdx.x & 63u;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int nrest = 3 * (base.M - 1);
  // a step rendered from truncated tile lists leaves the densification statistics alone (B3gsDensifyStats)
  const bool skip_stats = ds.skip_if_nonzero && *ds.skip_if_nonzero != 0;
  // the epoch byte of every view's forward (GeomView::staged), read once: eight bytes in one scalar pair
  unsigned long long epochs = 0ull;
#pragma unroll
  for (int v = 0; v < B3GS_MAX_FUSED_VIEWS; v++)
    if (v < mv.n && mv.v[v].staged) epochs |= (unsigned long long)(*mv.v[v].epoch & 0xFFu) << (8 * v);

  // ---- phase 1 ------------------------------------------------------------------------------------------
  // Load stage: the radii and marks of every view and the three statistics words of ALL of this thread's Gaussians are
  // requested before anything is consumed (ACC_PER_THREAD x 15 independent loads: one round trip instead of two or three per
  // Gaussian -- the 977 workgroups of a 1M-Gaussian call fill less than half of the chip's wave slots, so the pass is paid in
  // round trips).  The blend forward marks the Gaussians that sit below the deepest used position of some tile's list
  // (GeomView::staged): only their rows can hold anything -- ~5 % of the visible ones -- and the others are known to be zero
  // without being read.
  uint32_t vis_j[ACC_PER_THREAD], stg_j[ACC_PER_THREAD];
  int rad_j[ACC_PER_THREAD];
  float cnt_j[ACC_PER_THREAD], acc_j[ACC_PER_THREAD], den_j[ACC_PER_THREAD], mxr_j[ACC_PER_THREAD];
  const bool do_stats = ds.denom && !skip_stats;
#pragma unroll
  for (int j = 0; j < ACC_PER_THREAD; j++) {
    const int i = blk_first + j * 256 + (int)threadIdx.x;
    const bool valid = i < end;
    const int ic = valid ? i : end - 1;   // (end > first >= 0: the load stage has no control flow around its loads)
    uint32_t vis = 0u, stg = 0u;
    int st_rad = 0;
    float st_cnt = 0.f;
#pragma unroll
    for (int v = 0; v < B3GS_MAX_FUSED_VIEWS; v++) {
      if (v < mv.n) {
        const B3gsViewRef& vr = mv.v[v];
        const int rad = vr.radii[ic];
        const uint8_t ep = (uint8_t)(epochs >> (8 * v));
        const uint8_t mk = vr.staged ? vr.staged[ic] : ep;
        if (rad > 0) {
          vis |= 1u << v;
          if (vr.densify_stats) {   // visibility_filter = radii > 0 (train.py:178-179), contribution or not
            st_cnt += 1.0f;
            st_rad = max(st_rad, rad);
          }
        }
        if (mk == ep) stg |= 1u << v;
      }
    }
    vis_j[j] = valid ? vis : 0u, stg_j[j] = stg, rad_j[j] = st_rad, cnt_j[j] = valid ? st_cnt : 0.f;
    acc_j[j] = do_stats ? ds.xyz_gradient_accum[ic] : 0.f;
    den_j[j] = do_stats ? ds.denom[ic] : 0.f;
    mxr_j[j] = do_stats ? ds.max_radii2D[ic] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < ACC_PER_THREAD; j++) {
    const int li = j * 256 + (int)threadIdx.x;
    const int i = blk_first + li;
    uint32_t mask = 0;
    if (i < end) {
      const size_t i3 = 3 * (size_t)i;
      float st_norm = 0.f;
      const float st_cnt = cnt_j[j];
      const int st_rad = rad_j[j];
      const uint32_t vis = vis_j[j], stg = stg_j[j];
      for (int v = 0; v < mv.n; v++) {
        const B3gsViewRef& vr = mv.v[v];
        float* m2d = vr.dL_dmeans2D;
        if (!((vis >> v) & 1u)) {
          if (m2d) { m2d[i3] = 0.f; m2d[i3 + 1] = 0.f; m2d[i3 + 2] = 0.f; }
          continue;
        }
        float2 r0 = make_float2(0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0;
        if ((stg >> v) & 1u) {
          const float2* row = reinterpret_cast<const float2*>(vr.scratch) + 5 * (size_t)i;
          r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3], r4 = row[4];
        }
        // sums are never -0.0 (atomic adds onto +0.0), so the bit pattern tells "nothing arrived"
        const uint32_t any = ((__float_as_uint(r0.x) | __float_as_uint(r0.y)) | (__float_as_uint(r1.x) | __float_as_uint(r1.y))) |
                             ((__float_as_uint(r2.x) | __float_as_uint(r2.y)) | (__float_as_uint(r3.x) | __float_as_uint(r3.y))) |
                             (__float_as_uint(r4.x) | __float_as_uint(r4.y));
        if (any) mask |= 1u << v;
        if (m2d) { m2d[i3] = r2.x; m2d[i3 + 1] = r2.y; m2d[i3 + 2] = 0.f; }
        if (vr.densify_stats) st_norm += sqrtf(r2.x * r2.x + r2.y * r2.y);
      }
      if (do_stats && st_cnt > 0.f) {
        ds.xyz_gradient_accum[i] = acc_j[j] + st_norm;
        ds.denom[i] = den_j[j] + st_cnt;
        ds.max_radii2D[i] = fmaxf(mxr_j[j], (float)st_rad);
      }
      if (mask == 0u && overwrite && !rg.touched_rows) {   // no view has a gradient for this Gaussian: its rows of the slab are zero
        rg.xyz[i3] = 0.f; rg.xyz[i3 + 1] = 0.f; rg.xyz[i3 + 2] = 0.f;
        rg.scaling[i3] = 0.f; rg.scaling[i3 + 1] = 0.f; rg.scaling[i3 + 2] = 0.f;
        reinterpret_cast<float4*>(rg.rotation)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        rg.opacity[i] = 0.f;
        rg.features_dc[i3] = 0.f; rg.features_dc[i3 + 1] = 0.f; rg.features_dc[i3 + 2] = 0.f;
        float* rest = rg.features_rest + (size_t)nrest * i;
        for (int k = 0; k < nrest; k++) rest[k] = 0.f;
      }
    }
    const unsigned long long b = __ballot(mask != 0u);
    // sparse-row slab: the bitmap word of these 64 Gaussians replaces their zero rows (first % 64 == 0: word aligned)
    if (overwrite && rg.touched_rows && lane == 0 && blk_first + j * 256 + (int)(threadIdx.x & ~63u) < end)
      rg.touched_rows[(size_t)(blk_first + j * 256 + (int)(threadIdx.x & ~63u)) >> 6] = b;
    if (b) {
      const int leader = __builtin_ctzll(b);
      uint32_t at = 0;
      if ((int)lane == leader) at = atomicAdd(&s_count, (uint32_t)__builtin_popcountll(b));
      at = (uint32_t)__shfl((int)at, leader, 64);
      if (mask != 0u) s_list[at + (uint32_t)__builtin_popcountll(b & lt)] = (uint32_t)li | (mask << 16);
    }
  }
  __syncthreads();
  // the list of this block's touched Gaussians travels through global memory to the chain-rule kernel (which needs 168
  // VGPRs: 3 waves per SIMD; this streaming pass runs at 8)
  const uint32_t n_list = s_count;
  for (uint32_t e = threadIdx.x; e < n_list; e += 256) g_list[(size_t)blockIdx.x * ACC_BLOCK + e] = s_list[e];
  if (threadIdx.x == 0) g_count[blockIdx.x] = n_list;
}

// SH sink of the chain-rule kernel below: coefficients 0..REG-1 of ONE (Gaussian, view) pair in registers (they are added
// to the Gaussian's LDS row afterwards); higher ones go to memory with an atomic add (several lanes may serve one Gaussian)
template <int REG>
struct ShPairReg {
  float acc[3 * REG];
  float* rest;
  __device__ __forceinline__ void put(int k, int ch, float v) {
    if (k < REG) acc[3 * k + ch] += v;   // k, ch are compile-time constants at every call site
    else unsafeAtomicAdd(rest + 3 * (k - 1) + ch, v);
  }
};

// Phase 2: the chain rule of the touched (Gaussian, view) pairs.  A touched Gaussian has a gradient in 2.4 of the 6
// views on average; with lane = Gaussian and the view loop around it every wave ran all 6 view iterations with 40 % of its
// lanes (measured 60 us, latency-bound at 3 waves per SIMD).  Now, per group of 256 list entries: the entries of every
// view are compacted (ballots) into dense position lists, cut into chunks of <= 64 positions of ONE view, and the chunks
// are dealt to the four waves: a wave runs ~3 dense chunks instead of 6 sparse view iterations, the view of a chunk is
// wave-uniform (matrices through scalar loads).  The 23 gradients of a pair are added to the Gaussian's row of an LDS
// table (ds_add_f32: the pairs of one Gaussian sit in different chunks); the entry's own thread then finishes the row
// (sigmoid chain of the opacity) and stores it.
constexpr int CH_GROUP = 256;
constexpr int CH_ROW = 25;      // 23 gradients per Gaussian, odd row stride: conflict-free column access
template <int ACC_PER_THREAD>
__global__ void __launch_bounds__(256, B3GS_ACC_WAVES)
    accumulate_chain_kernel(B3gsScene base, B3gsRawParams raw, MultiViews mv, B3gsRawGrads rg, int overwrite, int first,
                            const uint32_t* __restrict__ g_list, const uint32_t* __restrict__ g_count) {
  constexpr int ACC_BLOCK = 256 * ACC_PER_THREAD;
  __shared__ float s_acc[CH_GROUP][CH_ROW];
  __shared__ uint8_t s_idx[B3GS_MAX_FUSED_VIEWS][CH_GROUP];   // (view, position) -> entry of the group
  __shared__ uint32_t s_wcnt[B3GS_MAX_FUSED_VIEWS][4];
  __shared__ uint32_t s_chunk[B3GS_MAX_FUSED_VIEWS * 4];      // view | first position << 8 | positions << 20
  __shared__ uint32_t s_nchunk;
  // One workgroup per GROUP of CH_GROUP list entries of a scan block (round 6; until then one workgroup per scan block looped
  // over its groups): the lists of neighbouring blocks differ wildly when the storage order of the Gaussians is spatially
  // coherent -- a model stored in Morton order put 1024 entries into some blocks and none into most, and the chain rule took
  // 53 us per view instead of 20 (tools/spatial_order_probe.py) -- while the scan wants its 1024-Gaussian blocks for streaming.
  // Empty groups exit on their first load.
  // (part-major: workgroups go round-robin over the eight XCDs by index, and with block-major numbering the first groups --
  // the only ones with work when the lists are short -- all landed on XCDs 0 and 4: measured, 20 -> 37 us per view)
  const uint32_t nblk = gridDim.x / (uint32_t)ACC_PER_THREAD;
  const uint32_t blk = blockIdx.x % nblk, part = blockIdx.x / nblk;
  const int blk_first = first + (int)blk * ACC_BLOCK;
  const int nrest = 3 * (base.M - 1);
  const uint32_t* __restrict__ s_list = g_list + (size_t)blk * ACC_BLOCK;
  const uint32_t n_list = min(g_count[blk], (part + 1u) * (uint32_t)CH_GROUP);
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  SceneX sx_;
  sx_.sc = base;
  sx_.raw = raw;
  sx_.raw_mode = 1;
  sx_.tight = 0;
#pragma unroll 1
  for (uint32_t e0 = part * (uint32_t)CH_GROUP; e0 < n_list; e0 += CH_GROUP) {   // (at most one trip)
    const uint32_t e = e0 + threadIdx.x;
    const uint32_t ent = e < n_list ? s_list[e] : 0u;
    const uint32_t mask = ent >> 16;
    const int i = blk_first + (int)(ent & 0xFFFFu);
    const size_t i3 = 3 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 23; k++) s_acc[threadIdx.x][k] = 0.f;
    if (mask != 0u && overwrite && base.M > 4) {   // coefficients beyond the register window accumulate in memory
      float* rest = rg.features_rest + (size_t)nrest * i;
      for (int k = 9; k < nrest; k++) rest[k] = 0.f;
    }
    // ---- positions of this group's entries inside every view's list
    uint32_t rank[B3GS_MAX_FUSED_VIEWS];
#pragma unroll
    for (int v = 0; v < B3GS_MAX_FUSED_VIEWS; v++) {
      rank[v] = 0;
      if (v < mv.n) {
        const unsigned long long b = __ballot((mask >> v) & 1u);
        rank[v] = (uint32_t)__builtin_popcountll(b & lt);
        if (lane == 0) s_wcnt[v][w] = (uint32_t)__builtin_popcountll(b);
      }
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < B3GS_MAX_FUSED_VIEWS; v++)
      if (v < mv.n && ((mask >> v) & 1u)) {
        uint32_t base_w = 0;
        for (unsigned k = 0; k < w; k++) base_w += s_wcnt[v][k];
        s_idx[v][base_w + rank[v]] = (uint8_t)threadIdx.x;
      }
    if (threadIdx.x == 0) {   // chunks of <= 64 positions of one view
      uint32_t nc = 0;
      for (int v = 0; v < mv.n; v++) {
        const uint32_t tot = (s_wcnt[v][0] + s_wcnt[v][1]) + (s_wcnt[v][2] + s_wcnt[v][3]);
        for (uint32_t p0 = 0; p0 < tot; p0 += 64u) s_chunk[nc++] = (uint32_t)v | (p0 << 8) | (min(64u, tot - p0) << 20);
      }
      s_nchunk = nc;
    }
    __syncthreads();
    // ---- the pairs, chunk by chunk
    const uint32_t nchunk = s_nchunk;
#pragma unroll 1
    for (uint32_t c = w; c < nchunk; c += 4u) {
      const uint32_t cd = __builtin_amdgcn_readfirstlane(s_chunk[c]);
      const int v = (int)(cd & 0xFFu);
      const uint32_t p0 = (cd >> 8) & 0xFFFu, np = cd >> 20;
      if (lane >= np) continue;
      const uint32_t t2 = s_idx[v][p0 + lane];
      const int i2 = blk_first + (int)(s_list[e0 + t2] & 0xFFFFu);
      const B3gsViewRef& vr = mv.v[v];
      sx_.sc.W = vr.W; sx_.sc.H = vr.H;
      sx_.sc.tan_fovx = vr.tan_fovx; sx_.sc.tan_fovy = vr.tan_fovy;
      sx_.sc.viewmatrix = vr.viewmatrix; sx_.sc.projmatrix = vr.projmatrix; sx_.sc.campos = vr.campos;
      const Mat16 vm = load_mat_uniform(vr.viewmatrix);
      const Mat16 pm = load_mat_uniform(vr.projmatrix);
      const PixSums in = load_scratch_row(vr.scratch, i2);   // read the sums, leave the row zero for the next iteration
      ShPairReg<4> sink;
#pragma unroll
      for (int k = 0; k < 12; k++) sink.acc[k] = 0.f;
      sink.rest = rg.features_rest + (size_t)nrest * i2;
      GaussGrad gg;
      gaussian_backward<true>(sx_, vm, pm, i2, __float_as_uint(vr.rec_words[16 * (size_t)i2 + 12]), in, true, true, gg, sink);
      float dsc[3];
      float4 dr;
      raw_chain(gg, dsc, dr);
      float* row = s_acc[t2];
      atomicAdd(row + 0, gg.dmean[0]); atomicAdd(row + 1, gg.dmean[1]); atomicAdd(row + 2, gg.dmean[2]);
      atomicAdd(row + 3, dsc[0]); atomicAdd(row + 4, dsc[1]); atomicAdd(row + 5, dsc[2]);
      atomicAdd(row + 6, dr.x); atomicAdd(row + 7, dr.y); atomicAdd(row + 8, dr.z); atomicAdd(row + 9, dr.w);
      atomicAdd(row + 10, in.gop);
#pragma unroll
      for (int k = 0; k < 12; k++) atomicAdd(row + 11 + k, sink.acc[k]);
    }
    __syncthreads();
    // ---- every entry's own thread finishes and stores its row
    if (mask != 0u) {
      const float* row = s_acc[threadIdx.x];
      const float op = load_opacity<true>(sx_, i);
      const float dop = row[10] * op * (1.0f - op);
      const int nreg = base.M < 4 ? base.M : 4;
      float4* rot_dst = reinterpret_cast<float4*>(rg.rotation) + i;
      float* dc = rg.features_dc + i3;
      float* rest = rg.features_rest + (size_t)nrest * i;
      if (overwrite) {
        rg.xyz[i3] = row[0]; rg.xyz[i3 + 1] = row[1]; rg.xyz[i3 + 2] = row[2];
        rg.scaling[i3] = row[3]; rg.scaling[i3 + 1] = row[4]; rg.scaling[i3 + 2] = row[5];
        *rot_dst = make_float4(row[6], row[7], row[8], row[9]);
        rg.opacity[i] = dop;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) dc[ch] = row[11 + ch];
#pragma unroll
        for (int k = 1; k < 4; k++)
          if (k < nreg) { rest[3 * (k - 1)] = row[11 + 3 * k]; rest[3 * (k - 1) + 1] = row[12 + 3 * k]; rest[3 * (k - 1) + 2] = row[13 + 3 * k]; }
      } else {
        rg.xyz[i3] += row[0]; rg.xyz[i3 + 1] += row[1]; rg.xyz[i3 + 2] += row[2];
        rg.scaling[i3] += row[3]; rg.scaling[i3 + 1] += row[4]; rg.scaling[i3 + 2] += row[5];
        float4 cur = *rot_dst;
        cur.x += row[6]; cur.y += row[7]; cur.z += row[8]; cur.w += row[9];
        *rot_dst = cur;
        rg.opacity[i] += dop;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) dc[ch] += row[11 + ch];
#pragma unroll
        for (int k = 1; k < 4; k++)
          if (k < nreg) { rest[3 * (k - 1)] += row[11 + 3 * k]; rest[3 * (k - 1) + 1] += row[12 + 3 * k]; rest[3 * (k - 1) + 2] += row[13 + 3 * k]; }
      }
    }
    // (the next group's zero fill of row t is thread t's own; its chunks start behind the barriers above)
  }
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ viewmatrix,
                                    uint8_t* __restrict__ present) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float x = means3D[3 * (size_t)i], y = means3D[3 * (size_t)i + 1], z = means3D[3 * (size_t)i + 2];
  float vz = ((viewmatrix[2] * x + viewmatrix[6] * y) + viewmatrix[10] * z) + viewmatrix[14];
  present[i] = vz > B3GS_NEAR ? 1 : 0;
}

}  // namespace

PreOut b3gs_pre_out(const B3gsScene& sc, const GeomView& g, const ImgView& im, int32_t* radii) {
  PreOut o;
  o.rec = g.rec;
  o.depth_key = g.depth_key;
  o.tiles_touched = g.tiles_touched;
  o.rect = g.rect;
  o.rect_stride = 1;
  o.rect_role = 0;
  o.radii = radii;
  o.ranges = im.ranges;
  o.ranges2 = im.ranges2;
  o.open_rows = im.open_rows;
  o.span_flag = nullptr;   // set by b3gs_forward_raw_batch when the 27-bit depth sort is requested
  o.pred_rows = nullptr;   // set by the two-round forward (b3gs_forward_raw_batch)
  o.pflag = g.pflag;
  o.nrowwords = ((sc.H + B3GS_TILE - 1) / B3GS_TILE) * (((sc.W + B3GS_TILE - 1) / B3GS_TILE + 63) / 64);
  o.ntiles = ((sc.W + B3GS_TILE - 1) / B3GS_TILE) * ((sc.H + B3GS_TILE - 1) / B3GS_TILE);
  return o;
}

void b3gs_launch_preprocess(const PreBatch& pb, hipStream_t s) {
  int work = pb.sc[0].P;
  for (int v = 0; v < pb.n; v++) work = pb.out[v].ntiles > work ? pb.out[v].ntiles : work;
  if (pb.n <= 0 || work <= 0) return;
  const dim3 grid((work + 255) / 256);
  if (pb.raw_mode) hipLaunchKernelGGL(preprocess_fwd_kernel<true>, grid, dim3(256), 0, s, pb);
  else hipLaunchKernelGGL(preprocess_fwd_kernel<false>, grid, dim3(256), 0, s, pb);
}

void b3gs_launch_preprocess_backward(const SceneX& sx, const GeomView& g, const int32_t* radii, float* dL_dmeans2D,
                                     float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D,
                                     float* dL_dsh, float* dL_dscales, float* dL_drotations, const B3gsRawGrads* rg,
                                     float* m2d_out, hipStream_t s) {
  if (sx.sc.P <= 0) return;
  const dim3 grid((sx.sc.P + 255) / 256);
  B3gsRawGrads none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (sx.raw_mode)
    hipLaunchKernelGGL(preprocess_bwd_kernel<true>, grid, dim3(256), 0, s, sx, g, radii, dL_dmeans2D, dL_dcolors,
                       dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, *rg, m2d_out);
  else
    hipLaunchKernelGGL(preprocess_bwd_kernel<false>, grid, dim3(256), 0, s, sx, g, radii, dL_dmeans2D, dL_dcolors,
                       dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, none, m2d_out);
}

void b3gs_launch_accumulate_views(const B3gsScene& base, const B3gsRawParams& raw, int nviews, const B3gsViewRef* views,
                                  const B3gsRawGrads& rg, int overwrite, const B3gsDensifyStats* stats, int first, int count,
                                  uint32_t* list, uint32_t* counts, hipStream_t s) {
  if (base.P <= 0 || nviews <= 0 || count <= 0) return;
  MultiViews mv;
  mv.n = nviews;
  for (int v = 0; v < nviews; v++) mv.v[v] = views[v];
  const B3gsDensifyStats ds = stats ? *stats : B3gsDensifyStats{nullptr, nullptr, nullptr, nullptr};
  // scratch of the two-kernel pass: the depth-sort ping-pong arrays of view 0's geometry buffer (P words each) are idle
  // once the forward has built its tile lists
  static const int force = getenv("B3GS_ACC_PER_THREAD") ? atoi(getenv("B3GS_ACC_PER_THREAD")) : 0;   // (A/B switch)
  // from 384k Gaussians: 1024 per scan workgroup (round 6: 768k until the chain rule ran one workgroup per list group -- the
  // 500k-Gaussian ranges of the pipelined data-parallel tail at 1M then gain 1 %: 602-607 -> 610-613 iters/s on the 1-rank group)
  const bool big = force ? force >= 4 : count >= (1 << 18) + (1 << 17);
  if (force == 2) {   // (experiment: 512 Gaussians per scan workgroup)
    const dim3 grid((count + 511) / 512);
    hipLaunchKernelGGL(accumulate_scan_kernel<2>, grid, dim3(256), 0, s, base, mv, rg, overwrite, first, count, ds, list, counts);
    hipLaunchKernelGGL(accumulate_chain_kernel<2>, dim3(grid.x * 2), dim3(256), 0, s, base, raw, mv, rg, overwrite, first, list, counts);
  } else if (big) {
    const dim3 grid((count + 1023) / 1024);
    hipLaunchKernelGGL(accumulate_scan_kernel<4>, grid, dim3(256), 0, s, base, mv, rg, overwrite, first, count, ds, list, counts);
    hipLaunchKernelGGL(accumulate_chain_kernel<4>, dim3(grid.x * 4), dim3(256), 0, s, base, raw, mv, rg, overwrite, first, list, counts);
  } else {
    const dim3 grid((count + 255) / 256);
    hipLaunchKernelGGL(accumulate_scan_kernel<1>, grid, dim3(256), 0, s, base, mv, rg, overwrite, first, count, ds, list, counts);
    hipLaunchKernelGGL(accumulate_chain_kernel<1>, grid, dim3(256), 0, s, base, raw, mv, rg, overwrite, first, list, counts);
  }
}

void b3gs_launch_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, uint8_t* present,
                              hipStream_t s) {
  if (P <= 0) return;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}


// ---- parity hook: the activations exactly as the kernels above evaluate them ------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) debug_activations_kernel(SceneX sx_, float* __restrict__ scales, float* __restrict__ rot,
                                                                float* __restrict__ opacity) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= sx_.sc.P) return;
  float s[3], q[4], inv;
  load_scale_rot<true>(sx_, i, s, q, &inv);
  for (int k = 0; k < 3; k++) scales[3 * (size_t)i + k] = s[k];
  for (int k = 0; k < 4; k++) rot[4 * (size_t)i + k] = q[k];
  opacity[i] = load_opacity<true>(sx_, i);
}
}  // namespace

extern "C" int b3gs_debug_activations(int32_t P, const B3gsRawParams* raw, float* scales, float* rotations, float* opacity,
                                      b3gs_stream_t stream) {
  if (P < 0 || !raw || (P > 0 && (!raw->scaling || !raw->rotation || !raw->opacity || !scales || !rotations || !opacity)))
    return b3gs_fail(B3GS_ERR_ARG, "b3gs_debug_activations", "NULL pointer or negative count");
  if (P == 0) return B3GS_OK;
  SceneX sx_{};
  sx_.sc.P = P;
  sx_.raw = *raw;
  sx_.raw_mode = 1;
  hipLaunchKernelGGL(debug_activations_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, sx_, scales, rotations, opacity);
  return b3gs_launch_status("b3gs_debug_activations");
}
