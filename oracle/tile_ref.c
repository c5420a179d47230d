/*
 * oracle/tile_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Tile-faithful CPU restatement of the differentiable Gaussian rasterizer that the
 * reference calls through `diff_gaussian_rasterization`
 * (reference call sites: gaussian_renderer/__init__.py:14,36-49,51,85-93).
 *
 * PARITY UNPINNED: the rasterizer's own source (ashawkey/diff-gaussian-rasterization,
 * .gitmodules:1-3 of the reference) is NOT vendored in /root/reference (empty directory,
 * no recoverable gitlink SHA) and the reference ships no tests, so no golden vector of the
 * rasterizer itself exists.  This file restates the published 3DGS tile-rasterisation
 * algorithm (Kerbl et al., SIGGRAPH 2023) extended with the depth and
 * alpha outputs/gradients the reference consumes (train.py:105-106,131,141-143).  The pieces
 * of the arithmetic that DO exist in the reference tree pin the corresponding stages here and
 * are checked against golden vectors generated from the reference's Python
 * (tests/golden/, tests/golden/make_golden.py):
 *   - SH basis, constants, +0.5 / clamp   utils/sh_utils.py:26-112, gaussian_renderer/__init__.py:74-78
 *   - quaternion -> R, Sigma = (R S)(R S)^T, 6-vector order   utils/general_utils.py:64-110,
 *                                                            scene/gaussian_model.py:27-31
 *   - matrix conventions (row-vector, transposed)  scene/cameras.py:55-58, utils/graphics_utils.py:38-71
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Arithmetic contract shared with the HIP path (so that integer decisions are bit-exact):
 *   - compiled with -ffp-contract=off; every fused multiply-add is an explicit fmaf()
 *   - division and sqrtf are IEEE correctly rounded
 *   - float -> int conversions are preceded by a clamp in float
 *   - exp() in the blend uses expf() here and the GPU's native exp2 there: colour/depth/alpha are
 *     therefore compared with a tolerance, and n_contrib may differ at borderline pixels.
 * Per-Gaussian gradient sums are accumulated in double here (the GPU sums fp32 partials in a
 * non-deterministic order), so gradients are compared with a relative tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_TILE 16
#define ORC_NEAR 0.2f
#define ORC_ALPHA_MIN (1.0f / 255.0f)
#define ORC_ALPHA_MAX 0.99f
#define ORC_T_EPS 0.0001f

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct {
  int P;        /* Gaussians */
  int D;        /* active SH degree (0..3) */
  int M;        /* SH coefficients per channel stored in `shs` (K); 0 if shs absent */
  int W, H;
  float tanfovx, tanfovy;
  float scale_modifier;
  int prefiltered;
  const float* bg;             /* [3] */
  const float* means3D;        /* [P,3] */
  const float* shs;            /* [P,M,3] or NULL */
  const float* colors_precomp; /* [P,3] or NULL */
  const float* opacities;      /* [P] */
  const float* scales;         /* [P,3] or NULL */
  const float* rotations;      /* [P,4] (w,x,y,z) or NULL */
  const float* cov3D_precomp;  /* [P,6] or NULL */
  const float* viewmatrix;     /* [16] row-vector convention: [x y z 1] @ V */
  const float* projmatrix;     /* [16] */
  const float* campos;         /* [3] */
} OrcScene;

/* Per-Gaussian forward state (all arrays caller-allocated, length P * width). */
typedef struct {
  float* depths;        /* [P]   view-space z */
  int32_t* radii;       /* [P]   */
  float* means2D;       /* [P,2] pixel coordinates */
  float* cov3D;         /* [P,6] */
  float* conic_opacity; /* [P,4] */
  float* rgb;           /* [P,3] */
  uint8_t* clamped;     /* [P,3] */
  int32_t* rect;        /* [P,4] xmin,ymin,xmax,ymax in tiles */
  uint32_t* tiles_touched; /* [P] */
  uint32_t* point_offsets; /* [P] inclusive scan */
} OrcGeom;

/* ---- small helpers; the op order in each is part of the contract ---- */

static inline void xform_point_4x3(const float* m, const float* p, float* o) {
  o[0] = ((m[0] * p[0] + m[4] * p[1]) + m[8] * p[2]) + m[12];
  o[1] = ((m[1] * p[0] + m[5] * p[1]) + m[9] * p[2]) + m[13];
  o[2] = ((m[2] * p[0] + m[6] * p[1]) + m[10] * p[2]) + m[14];
}
static inline void xform_point_4x4(const float* m, const float* p, float* o) {
  o[0] = ((m[0] * p[0] + m[4] * p[1]) + m[8] * p[2]) + m[12];
  o[1] = ((m[1] * p[0] + m[5] * p[1]) + m[9] * p[2]) + m[13];
  o[2] = ((m[2] * p[0] + m[6] * p[1]) + m[10] * p[2]) + m[14];
  o[3] = ((m[3] * p[0] + m[7] * p[1]) + m[11] * p[2]) + m[15];
}

/* quaternion (w,x,y,z), used as given (no normalisation; the reference normalises outside,
 * scene/gaussian_model.py:100-101) -> R, same entries as utils/general_utils.py:90-98 */
static inline void quat_to_R(const float* q, float R[9]) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
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

/* Sigma = L L^T with L = R diag(mod*s); 6-vector (00,01,02,11,12,22) as utils/general_utils.py:64-73 */
static inline void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* c6) {
  float R[9], L[9];
  quat_to_R(q, R);
  float sx = mod * s[0], sy = mod * s[1], sz = mod * s[2];
  for (int i = 0; i < 3; i++) {
    L[3 * i + 0] = R[3 * i + 0] * sx;
    L[3 * i + 1] = R[3 * i + 1] * sy;
    L[3 * i + 2] = R[3 * i + 2] * sz;
  }
#define LL(i, k) ((L[3 * i] * L[3 * k] + L[3 * i + 1] * L[3 * k + 1]) + L[3 * i + 2] * L[3 * k + 2])
  c6[0] = LL(0, 0);
  c6[1] = LL(0, 1);
  c6[2] = LL(0, 2);
  c6[3] = LL(1, 1);
  c6[4] = LL(1, 2);
  c6[5] = LL(2, 2);
#undef LL
}

/* EWA projection of the 3D covariance; returns (a,b,c) WITHOUT the low-pass term.
 * Also hands back the intermediate 2x3 matrix T = J * Wr and clamp flags for the backward. */
static inline void cov2d_project(const float* pv, float fx, float fy, float tanfovx, float tanfovy,
                                 const float* c6, const float* vm, float* a, float* b, float* c) {
  float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
  float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
  float tx = fminf(limx, fmaxf(-limx, txtz)) * pv[2];
  float ty = fminf(limy, fmaxf(-limy, tytz)) * pv[2];
  float tz = pv[2];
  float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
  float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
  /* Wr[i][j] = vm[4*j + i]  (p_view = Wr p + t) */
  float T0[3], T1[3];
  for (int j = 0; j < 3; j++) {
    T0[j] = J00 * vm[4 * j + 0] + J02 * vm[4 * j + 2];
    T1[j] = J11 * vm[4 * j + 1] + J12 * vm[4 * j + 2];
  }
  /* S = Sigma (symmetric) ; v0 = S T0^T ; v1 = S T1^T */
  float S00 = c6[0], S01 = c6[1], S02 = c6[2], S11 = c6[3], S12 = c6[4], S22 = c6[5];
  float v0[3] = {(S00 * T0[0] + S01 * T0[1]) + S02 * T0[2], (S01 * T0[0] + S11 * T0[1]) + S12 * T0[2],
                 (S02 * T0[0] + S12 * T0[1]) + S22 * T0[2]};
  float v1[3] = {(S00 * T1[0] + S01 * T1[1]) + S02 * T1[2], (S01 * T1[0] + S11 * T1[1]) + S12 * T1[2],
                 (S02 * T1[0] + S12 * T1[1]) + S22 * T1[2]};
  *a = (T0[0] * v0[0] + T0[1] * v0[1]) + T0[2] * v0[2];
  *b = (T0[0] * v1[0] + T0[1] * v1[1]) + T0[2] * v1[2];
  *c = (T1[0] * v1[0] + T1[1] * v1[1]) + T1[2] * v1[2];
}

/* SH -> RGB (before +0.5 / clamp).  sh is [M][3] for this Gaussian.  Sign pattern and constants
 * follow utils/sh_utils.py:70-100. */
static inline void sh_to_rgb(int deg, const float* sh, const float* dir, float* out) {
  float x = dir[0], y = dir[1], z = dir[2];
  for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[3 * (k) + ch]
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
    out[ch] = r;
  }
}

static inline int clampi_from_float(float v, int hi) {
  /* clamp in float first: well defined for any finite/inf/NaN input, identical to
   * min(hi, max(0, (int)v)) wherever the latter is defined */
  float c = fminf(fmaxf(v, 0.0f), (float)hi);
  return (int)c;
}

/* ------------------------------------------------------------------ */
/* A.1 preprocess                                                      */
/* ------------------------------------------------------------------ */
void orc_preprocess(const OrcScene* s, OrcGeom* g) {
  const int P = s->P;
  const float fx = (float)s->W / (2.0f * s->tanfovx);
  const float fy = (float)s->H / (2.0f * s->tanfovy);
  const int gx = (s->W + ORC_TILE - 1) / ORC_TILE, gy = (s->H + ORC_TILE - 1) / ORC_TILE;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    g->radii[i] = 0;
    g->tiles_touched[i] = 0;
    g->depths[i] = 0.f;
    g->means2D[2 * i] = g->means2D[2 * i + 1] = 0.f;
    for (int k = 0; k < 6; k++) g->cov3D[6 * i + k] = 0.f;
    for (int k = 0; k < 4; k++) g->conic_opacity[4 * i + k] = 0.f;
    for (int k = 0; k < 3; k++) { g->rgb[3 * i + k] = 0.f; g->clamped[3 * i + k] = 0; }
    for (int k = 0; k < 4; k++) g->rect[4 * i + k] = 0;

    const float* p = s->means3D + 3 * i;
    float pv[3];
    xform_point_4x3(s->viewmatrix, p, pv);
    if (!s->prefiltered && pv[2] <= ORC_NEAR) continue;

    float ph[4];
    xform_point_4x4(s->projmatrix, p, ph);
    float pw = 1.0f / (ph[3] + 0.0000001f);
    float pp[2] = {ph[0] * pw, ph[1] * pw};

    float c6[6];
    if (s->cov3D_precomp) {
      for (int k = 0; k < 6; k++) c6[k] = s->cov3D_precomp[6 * i + k];
    } else {
      cov3d_from_scale_rot(s->scales + 3 * i, s->scale_modifier, s->rotations + 4 * i, c6);
    }
    for (int k = 0; k < 6; k++) g->cov3D[6 * i + k] = c6[k];

    float a, b, c;
    cov2d_project(pv, fx, fy, s->tanfovx, s->tanfovy, c6, s->viewmatrix, &a, &b, &c);
    a += 0.3f;
    c += 0.3f;
    float det = a * c - b * b;
    if (det == 0.0f) continue;
    float det_inv = 1.0f / det;
    float conic[3] = {c * det_inv, -b * det_inv, a * det_inv};

    float mid = 0.5f * (a + c);
    float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    float l1 = mid + disc, l2 = mid - disc;
    float rad_f = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
    float px = ((pp[0] + 1.0f) * (float)s->W - 1.0f) * 0.5f;
    float py = ((pp[1] + 1.0f) * (float)s->H - 1.0f) * 0.5f;
    int x0 = clampi_from_float((px - rad_f) / (float)ORC_TILE, gx);
    int y0 = clampi_from_float((py - rad_f) / (float)ORC_TILE, gy);
    int x1 = clampi_from_float((px + rad_f + (float)(ORC_TILE - 1)) / (float)ORC_TILE, gx);
    int y1 = clampi_from_float((py + rad_f + (float)(ORC_TILE - 1)) / (float)ORC_TILE, gy);
    if ((x1 - x0) * (y1 - y0) == 0) continue;

    if (s->colors_precomp) {
      for (int k = 0; k < 3; k++) g->rgb[3 * i + k] = s->colors_precomp[3 * i + k];
    } else {
      float d[3] = {p[0] - s->campos[0], p[1] - s->campos[1], p[2] - s->campos[2]};
      float inv = 1.0f / sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
      float dir[3] = {d[0] * inv, d[1] * inv, d[2] * inv};
      float rgb[3];
      sh_to_rgb(s->D, s->shs + (size_t)3 * s->M * i, dir, rgb);
      for (int k = 0; k < 3; k++) {
        float v = rgb[k] + 0.5f;
        g->clamped[3 * i + k] = (v < 0.0f);
        g->rgb[3 * i + k] = fmaxf(v, 0.0f);
      }
    }
    g->depths[i] = pv[2];
    /* radius is an integer-valued float; clamp before the cast (same on the GPU) */
    g->radii[i] = (int32_t)fminf(rad_f, 2147483520.0f);
    g->means2D[2 * i] = px;
    g->means2D[2 * i + 1] = py;
    g->conic_opacity[4 * i + 0] = conic[0];
    g->conic_opacity[4 * i + 1] = conic[1];
    g->conic_opacity[4 * i + 2] = conic[2];
    g->conic_opacity[4 * i + 3] = s->opacities[i];
    g->rect[4 * i + 0] = x0;
    g->rect[4 * i + 1] = y0;
    g->rect[4 * i + 2] = x1;
    g->rect[4 * i + 3] = y1;
    g->tiles_touched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
  }
  /* inclusive scan */
  uint32_t acc = 0;
  for (int i = 0; i < P; i++) {
    acc += g->tiles_touched[i];
    g->point_offsets[i] = acc;
  }
}

/* ------------------------------------------------------------------ */
/* A.2 binning: emit (tile<<32 | depth bits, idx), stable sort, ranges  */
/* ------------------------------------------------------------------ */
static void stable_sort_u64(uint64_t* keys, uint32_t* vals, size_t n, int bits) {
  /* LSD radix sort, 8-bit digits: stable by construction */
  uint64_t* k2 = (uint64_t*)malloc(n * sizeof(uint64_t));
  uint32_t* v2 = (uint32_t*)malloc(n * sizeof(uint32_t));
  for (int shift = 0; shift < bits; shift += 8) {
    size_t cnt[257];
    memset(cnt, 0, sizeof(cnt));
    for (size_t i = 0; i < n; i++) cnt[((keys[i] >> shift) & 0xFF) + 1]++;
    for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
    for (size_t i = 0; i < n; i++) {
      size_t dst = cnt[(keys[i] >> shift) & 0xFF]++;
      k2[dst] = keys[i];
      v2[dst] = vals[i];
    }
    memcpy(keys, k2, n * sizeof(uint64_t));
    memcpy(vals, v2, n * sizeof(uint32_t));
  }
  free(k2);
  free(v2);
}

/* keys/point_list: caller-allocated, length N = point_offsets[P-1]; ranges: [tiles,2] */
void orc_bin(const OrcScene* s, const OrcGeom* g, uint64_t* keys, uint32_t* point_list, uint32_t* ranges) {
  const int P = s->P;
  const int gx = (s->W + ORC_TILE - 1) / ORC_TILE, gy = (s->H + ORC_TILE - 1) / ORC_TILE;
  const size_t N = P ? g->point_offsets[P - 1] : 0;
  for (int i = 0; i < P; i++) {
    if (g->radii[i] <= 0) continue;
    size_t off = (i == 0) ? 0 : g->point_offsets[i - 1];
    uint32_t dbits;
    memcpy(&dbits, &g->depths[i], 4);
    for (int y = g->rect[4 * i + 1]; y < g->rect[4 * i + 3]; y++)
      for (int x = g->rect[4 * i + 0]; x < g->rect[4 * i + 2]; x++) {
        uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
        key = (key << 32) | dbits;
        keys[off] = key;
        point_list[off] = (uint32_t)i;
        off++;
      }
  }
  int tiles = gx * gy, tbits = 0;
  while ((1 << tbits) < tiles) tbits++; /* enough bits to hold every tile id */
  stable_sort_u64(keys, point_list, N, 32 + ((tbits + 7) / 8) * 8);
  for (int t = 0; t < tiles; t++) ranges[2 * t] = ranges[2 * t + 1] = 0;
  for (size_t i = 0; i < N; i++) {
    uint32_t t = (uint32_t)(keys[i] >> 32);
    if (i == 0 || t != (uint32_t)(keys[i - 1] >> 32)) ranges[2 * t] = (uint32_t)i;
    if (i + 1 == N || t != (uint32_t)(keys[i + 1] >> 32)) ranges[2 * t + 1] = (uint32_t)(i + 1);
  }
}

/* ------------------------------------------------------------------ */
/* A.3 blend forward                                                   */
/* ------------------------------------------------------------------ */
static inline float blend_power(const float* co, float dx, float dy) {
  /* -0.5 (cxx dx^2 + cyy dy^2) - cxy dx dy, fixed op order with explicit fma */
  float q = fmaf(co[2] * dy, dy, (co[0] * dx) * dx);
  return fmaf(-co[1] * dx, dy, -0.5f * q);
}

void orc_render_forward(const OrcScene* s, const OrcGeom* g, const uint32_t* point_list, const uint32_t* ranges,
                        float* out_color, float* out_depth, float* out_alpha, float* final_T, uint32_t* n_contrib) {
  const int W = s->W, H = s->H;
  const int gx = (W + ORC_TILE - 1) / ORC_TILE, gy = (H + ORC_TILE - 1) / ORC_TILE;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < gy; ty++)
    for (int tx = 0; tx < gx; tx++) {
      uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
      for (int ly = 0; ly < ORC_TILE; ly++)
        for (int lx = 0; lx < ORC_TILE; lx++) {
          int px = tx * ORC_TILE + lx, py = ty * ORC_TILE + ly;
          if (px >= W || py >= H) continue;
          float T = 1.0f, C[3] = {0, 0, 0}, Dp = 0.f, A = 0.f;
          uint32_t contributor = 0, last = 0;
          for (uint32_t j = r0; j < r1; j++) {
            contributor++;
            uint32_t id = point_list[j];
            const float* co = g->conic_opacity + 4 * id;
            float dx = g->means2D[2 * id] - (float)px, dy = g->means2D[2 * id + 1] - (float)py;
            float power = blend_power(co, dx, dy);
            if (power > 0.0f) continue;
            float alpha = fminf(ORC_ALPHA_MAX, co[3] * expf(power));
            if (alpha < ORC_ALPHA_MIN) continue;
            float test_T = T * (1.0f - alpha);
            if (test_T < ORC_T_EPS) break;
            float w = alpha * T;
            for (int ch = 0; ch < 3; ch++) C[ch] = fmaf(g->rgb[3 * id + ch], w, C[ch]);
            Dp = fmaf(g->depths[id], w, Dp);
            A += w;
            T = test_T;
            last = contributor;
          }
          size_t pix = (size_t)py * W + px;
          final_T[pix] = T;
          n_contrib[pix] = last;
          for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix] = fmaf(T, s->bg[ch], C[ch]);
          out_depth[pix] = Dp;
          out_alpha[pix] = A;
        }
    }
}

/* ------------------------------------------------------------------ */
/* A.4 blend backward: per-pixel reverse replay, double accumulators   */
/* ------------------------------------------------------------------ */
typedef struct {
  double* dL_dmean2D;  /* [P,2] (already scaled by 0.5*W, 0.5*H: NDC units, train.py:179 threshold) */
  double* dL_dconic;   /* [P,3] xx, xy, yy */
  double* dL_dopacity; /* [P] */
  double* dL_dcolor;   /* [P,3] */
  double* dL_ddepth;   /* [P] */
} OrcPixGrads;

void orc_render_backward(const OrcScene* s, const OrcGeom* g, const uint32_t* point_list, const uint32_t* ranges,
                         const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor_img,
                         const float* dL_ddepth_img, const float* dL_dalpha_img, OrcPixGrads* o) {
  const int W = s->W, H = s->H;
  const int gx = (W + ORC_TILE - 1) / ORC_TILE, gy = (H + ORC_TILE - 1) / ORC_TILE;
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  /* tiles in parallel; the double accumulators are updated atomically (summation order then
   * varies run to run at the 1e-16 level, far below the fp32 comparisons made against them) */
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < gy; ty++)
    for (int tx = 0; tx < gx; tx++) {
      uint32_t r0 = ranges[2 * (ty * gx + tx)];
      for (int ly = 0; ly < ORC_TILE; ly++)
        for (int lx = 0; lx < ORC_TILE; lx++) {
          int px = tx * ORC_TILE + lx, py = ty * ORC_TILE + ly;
          if (px >= W || py >= H) continue;
          size_t pix = (size_t)py * W + px;
          const float T_final = final_T[pix];
          float T = T_final;
          uint32_t last = n_contrib[pix];
          float dC[3] = {dL_dcolor_img[pix], dL_dcolor_img[(size_t)H * W + pix], dL_dcolor_img[(size_t)2 * H * W + pix]};
          float dD = dL_ddepth_img ? dL_ddepth_img[pix] : 0.f;
          float dA = dL_dalpha_img ? dL_dalpha_img[pix] : 0.f;
          float bg_dot = (s->bg[0] * dC[0] + s->bg[1] * dC[1]) + s->bg[2] * dC[2];
          float accum_c[3] = {0, 0, 0}, last_c[3] = {0, 0, 0};
          float accum_d = 0.f, last_d = 0.f, accum_a = 0.f, last_alpha = 0.f;
          for (uint32_t k = last; k-- > 0;) {
            uint32_t id = point_list[r0 + k];
            const float* co = g->conic_opacity + 4 * id;
            float dx = g->means2D[2 * id] - (float)px, dy = g->means2D[2 * id + 1] - (float)py;
            float power = blend_power(co, dx, dy);
            if (power > 0.0f) continue;
            float G = expf(power);
            float alpha = fminf(ORC_ALPHA_MAX, co[3] * G);
            if (alpha < ORC_ALPHA_MIN) continue;
            T = T / (1.0f - alpha);
            float w = alpha * T;
            float dL_dalpha = 0.f;
            for (int ch = 0; ch < 3; ch++) {
              float c = g->rgb[3 * id + ch];
              accum_c[ch] = fmaf(last_alpha, last_c[ch], (1.f - last_alpha) * accum_c[ch]);
              last_c[ch] = c;
              dL_dalpha = fmaf(c - accum_c[ch], dC[ch], dL_dalpha);
              #pragma omp atomic
              o->dL_dcolor[3 * id + ch] += (double)(w * dC[ch]);
            }
            float dep = g->depths[id];
            accum_d = fmaf(last_alpha, last_d, (1.f - last_alpha) * accum_d);
            last_d = dep;
            dL_dalpha = fmaf(dep - accum_d, dD, dL_dalpha);
            #pragma omp atomic
            o->dL_ddepth[id] += (double)(w * dD);
            accum_a = fmaf(last_alpha, 1.0f, (1.f - last_alpha) * accum_a);
            dL_dalpha = fmaf(1.0f - accum_a, dA, dL_dalpha);
            dL_dalpha *= T;
            last_alpha = alpha;
            dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
            /* alpha = min(0.99, op * G): the published algorithm passes the gradient straight
             * through the cap (no zeroing at 0.99); kept. */
            float dL_dG = co[3] * dL_dalpha;
            float gdx = G * dx, gdy = G * dy;
            float dG_ddx = -gdx * co[0] - gdy * co[1];
            float dG_ddy = -gdy * co[2] - gdx * co[1];
            #pragma omp atomic
            o->dL_dmean2D[2 * id + 0] += (double)(dL_dG * dG_ddx * ddelx_dx);
            #pragma omp atomic
            o->dL_dmean2D[2 * id + 1] += (double)(dL_dG * dG_ddy * ddely_dy);
            #pragma omp atomic
            o->dL_dconic[3 * id + 0] += (double)(-0.5f * gdx * dx * dL_dG);
            #pragma omp atomic
            o->dL_dconic[3 * id + 1] += (double)(-0.5f * gdx * dy * dL_dG);
            #pragma omp atomic
            o->dL_dconic[3 * id + 2] += (double)(-0.5f * gdy * dy * dL_dG);
            #pragma omp atomic
            o->dL_dopacity[id] += (double)(G * dL_dalpha);
          }
        }
    }
}

/* ------------------------------------------------------------------ */
/* A.5 per-Gaussian backward (double arithmetic: this is the reference  */
/* the fp32 HIP path is compared with at rtol)                          */
/* ------------------------------------------------------------------ */
typedef struct {
  float* dL_dmeans3D; /* [P,3] */
  float* dL_dcov3D;   /* [P,6] */
  float* dL_dsh;      /* [P,M,3] or NULL */
  float* dL_dcolors;  /* [P,3]  (gradient w.r.t. colors_precomp / rgb) */
  float* dL_dscales;  /* [P,3] or NULL */
  float* dL_drots;    /* [P,4] or NULL */
  float* dL_dopacity; /* [P] */
  float* dL_dmeans2D; /* [P,3] (z = 0) */
} OrcGrads;

static void sh_backward(int deg, int M, const float* sh, const double* dirn, const double* d_unnorm, double inv_len,
                        const uint8_t* clamped, const double* dL_drgb_in, float* dL_dsh, double* dL_dmean_add) {
  double x = dirn[0], y = dirn[1], z = dirn[2];
  double dL_drgb[3];
  for (int c = 0; c < 3; c++) dL_drgb[c] = clamped[c] ? 0.0 : dL_drgb_in[c];
  double dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
  double basis[16];
  for (int k = 0; k < 16; k++) basis[k] = 0.0;
  basis[0] = SH_C0;
  if (deg > 0) {
    basis[1] = -SH_C1 * y;
    basis[2] = SH_C1 * z;
    basis[3] = -SH_C1 * x;
    for (int c = 0; c < 3; c++) {
      dRGBdx[c] = -SH_C1 * sh[3 * 3 + c];
      dRGBdy[c] = -SH_C1 * sh[3 * 1 + c];
      dRGBdz[c] = SH_C1 * sh[3 * 2 + c];
    }
    if (deg > 1) {
      double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      basis[4] = SH_C2[0] * xy;
      basis[5] = SH_C2[1] * yz;
      basis[6] = SH_C2[2] * (2.0 * zz - xx - yy);
      basis[7] = SH_C2[3] * xz;
      basis[8] = SH_C2[4] * (xx - yy);
      for (int c = 0; c < 3; c++) {
#define SH(k) ((double)sh[3 * (k) + c])
        dRGBdx[c] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.0 * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.0 * x * SH(8);
        dRGBdy[c] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.0 * -y * SH(6) + SH_C2[4] * 2.0 * -y * SH(8);
        dRGBdz[c] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.0 * 2.0 * z * SH(6) + SH_C2[3] * x * SH(7);
        if (deg > 2) {
          dRGBdx[c] += SH_C3[0] * SH(9) * 3.0 * 2.0 * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.0 * xy +
                       SH_C3[3] * SH(12) * -3.0 * 2.0 * xz + SH_C3[4] * SH(13) * (-3.0 * xx + 4.0 * zz - yy) +
                       SH_C3[5] * SH(14) * 2.0 * xz + SH_C3[6] * SH(15) * 3.0 * (xx - yy);
          dRGBdy[c] += SH_C3[0] * SH(9) * 3.0 * (xx - yy) + SH_C3[1] * SH(10) * xz +
                       SH_C3[2] * SH(11) * (-3.0 * yy + 4.0 * zz - xx) + SH_C3[3] * SH(12) * -3.0 * 2.0 * yz +
                       SH_C3[4] * SH(13) * -2.0 * xy + SH_C3[5] * SH(14) * -2.0 * yz + SH_C3[6] * SH(15) * -3.0 * 2.0 * xy;
          dRGBdz[c] += SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.0 * 2.0 * yz +
                       SH_C3[3] * SH(12) * 3.0 * (2.0 * zz - xx - yy) + SH_C3[4] * SH(13) * 4.0 * 2.0 * xz +
                       SH_C3[5] * SH(14) * (xx - yy);
        }
#undef SH
      }
      if (deg > 2) {
        basis[9] = SH_C3[0] * y * (3.0 * xx - yy);
        basis[10] = SH_C3[1] * xy * z;
        basis[11] = SH_C3[2] * y * (4.0 * zz - xx - yy);
        basis[12] = SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy);
        basis[13] = SH_C3[4] * x * (4.0 * zz - xx - yy);
        basis[14] = SH_C3[5] * z * (xx - yy);
        basis[15] = SH_C3[6] * x * (xx - 3.0 * yy);
      }
    }
  }
  int nb = (deg + 1) * (deg + 1);
  for (int k = 0; k < M; k++)
    for (int c = 0; c < 3; c++) dL_dsh[3 * k + c] = (k < nb) ? (float)(basis[k] * dL_drgb[c]) : 0.f;
  /* gradient w.r.t. the (unit) direction, then through the normalisation d/|d| */
  double gdir[3] = {0, 0, 0};
  for (int c = 0; c < 3; c++) {
    gdir[0] += dRGBdx[c] * dL_drgb[c];
    gdir[1] += dRGBdy[c] * dL_drgb[c];
    gdir[2] += dRGBdz[c] * dL_drgb[c];
  }
  /* d(v/|v|)/dv = (I - n n^T)/|v| */
  double dot = gdir[0] * dirn[0] + gdir[1] * dirn[1] + gdir[2] * dirn[2];
  (void)d_unnorm;
  for (int k = 0; k < 3; k++) dL_dmean_add[k] = (gdir[k] - dirn[k] * dot) * inv_len;
}

void orc_preprocess_backward(const OrcScene* s, const OrcGeom* g, const OrcPixGrads* pg, OrcGrads* o) {
  const int P = s->P;
  const double fx = (double)((float)s->W / (2.0f * s->tanfovx));
  const double fy = (double)((float)s->H / (2.0f * s->tanfovy));
  const float* vm = s->viewmatrix;
  const float* pm = s->projmatrix;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    for (int k = 0; k < 3; k++) { o->dL_dmeans3D[3 * i + k] = 0.f; o->dL_dcolors[3 * i + k] = 0.f; o->dL_dmeans2D[3 * i + k] = 0.f; }
    for (int k = 0; k < 6; k++) o->dL_dcov3D[6 * i + k] = 0.f;
    if (o->dL_dsh) for (int k = 0; k < 3 * s->M; k++) o->dL_dsh[(size_t)3 * s->M * i + k] = 0.f;
    if (o->dL_dscales) for (int k = 0; k < 3; k++) o->dL_dscales[3 * i + k] = 0.f;
    if (o->dL_drots) for (int k = 0; k < 4; k++) o->dL_drots[4 * i + k] = 0.f;
    o->dL_dopacity[i] = 0.f;
    if (g->radii[i] <= 0) continue;

    o->dL_dopacity[i] = (float)pg->dL_dopacity[i];
    o->dL_dmeans2D[3 * i + 0] = (float)pg->dL_dmean2D[2 * i + 0];
    o->dL_dmeans2D[3 * i + 1] = (float)pg->dL_dmean2D[2 * i + 1];
    for (int k = 0; k < 3; k++) o->dL_dcolors[3 * i + k] = (float)pg->dL_dcolor[3 * i + k];

    const float* p = s->means3D + 3 * i;
    double mean[3] = {p[0], p[1], p[2]};
    double dmean[3] = {0, 0, 0};

    /* ---- conic -> cov2D ---- */
    float pvf[3];
    xform_point_4x3(vm, p, pvf);
    const float* c6f = g->cov3D + 6 * i;
    /* recompute cov2D in double from the stored cov3D */
    double limx = 1.3 * (double)s->tanfovx, limy = 1.3 * (double)s->tanfovy;
    double tz = pvf[2];
    double txtz = pvf[0] / tz, tytz = pvf[1] / tz;
    double tx = fmin(limx, fmax(-limx, txtz)) * tz, ty = fmin(limy, fmax(-limy, tytz)) * tz;
    double x_grad_mul = (txtz < -limx || txtz > limx) ? 0.0 : 1.0;
    double y_grad_mul = (tytz < -limy || tytz > limy) ? 0.0 : 1.0;
    double J00 = fx / tz, J02 = -(fx * tx) / (tz * tz), J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    double Wr[3][3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Wr[r][c] = vm[4 * c + r];
    double T0[3], T1[3];
    for (int j = 0; j < 3; j++) {
      T0[j] = J00 * Wr[0][j] + J02 * Wr[2][j];
      T1[j] = J11 * Wr[1][j] + J12 * Wr[2][j];
    }
    double S[3][3] = {{c6f[0], c6f[1], c6f[2]}, {c6f[1], c6f[3], c6f[4]}, {c6f[2], c6f[4], c6f[5]}};
    double a = 0, b = 0, c = 0;
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) {
        a += T0[r] * S[r][q] * T0[q];
        b += T0[r] * S[r][q] * T1[q];
        c += T1[r] * S[r][q] * T1[q];
      }
    a += 0.3;
    c += 0.3;
    double denom = a * c - b * b;
    double gxx = pg->dL_dconic[3 * i + 0], gxy = pg->dL_dconic[3 * i + 1], gyy = pg->dL_dconic[3 * i + 2];
    double dL_da = 0, dL_db = 0, dL_dc = 0;
    double denom2inv = 1.0 / (denom * denom + 0.0000001);
    if (denom2inv != 0.0) {
      /* conic = (c, -b, a)/denom ; d/d(a,b,c) */
      dL_da = denom2inv * (-c * c * gxx + 2.0 * b * c * gxy + (denom - a * c) * gyy);
      dL_dc = denom2inv * (-a * a * gyy + 2.0 * a * b * gxy + (denom - a * c) * gxx);
      dL_db = denom2inv * 2.0 * (b * c * gxx - (denom + 2.0 * b * b) * gxy + a * b * gyy);
    }
    /* cov2D = T S T^T: dL/dS (6 unique entries, off-diagonals count twice) */
    double dS[6];
    dS[0] = T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc;
    dS[3] = T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc;
    dS[5] = T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc;
    dS[1] = 2.0 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2.0 * T1[0] * T1[1] * dL_dc;
    dS[2] = 2.0 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2.0 * T1[0] * T1[2] * dL_dc;
    dS[4] = 2.0 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2.0 * T1[1] * T1[2] * dL_dc;
    for (int k = 0; k < 6; k++) o->dL_dcov3D[6 * i + k] = (float)dS[k];

    /* dL/dT (2x3) : a = T0 S T0, b = T0 S T1, c = T1 S T1 */
    double ST0[3], ST1[3];
    for (int r = 0; r < 3; r++) {
      ST0[r] = S[r][0] * T0[0] + S[r][1] * T0[1] + S[r][2] * T0[2];
      ST1[r] = S[r][0] * T1[0] + S[r][1] * T1[1] + S[r][2] * T1[2];
    }
    double dT0[3], dT1[3];
    for (int r = 0; r < 3; r++) {
      dT0[r] = 2.0 * ST0[r] * dL_da + ST1[r] * dL_db;
      dT1[r] = 2.0 * ST1[r] * dL_dc + ST0[r] * dL_db;
    }
    /* T0 = J00 Wr[0] + J02 Wr[2] ; T1 = J11 Wr[1] + J12 Wr[2] */
    double dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
    for (int j = 0; j < 3; j++) {
      dJ00 += Wr[0][j] * dT0[j];
      dJ02 += Wr[2][j] * dT0[j];
      dJ11 += Wr[1][j] * dT1[j];
      dJ12 += Wr[2][j] * dT1[j];
    }
    double tz2 = 1.0 / (tz * tz), tz3 = tz2 / tz;
    double dtx = x_grad_mul * -fx * tz2 * dJ02;
    double dty = y_grad_mul * -fy * tz2 * dJ12;
    double dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.0 * fx * tx) * tz3 * dJ02 + (2.0 * fy * ty) * tz3 * dJ12;
    /* p_view = Wr mean + t  =>  dL/dmean = Wr^T dL/dt */
    for (int k = 0; k < 3; k++) dmean[k] = Wr[0][k] * dtx + Wr[1][k] * dty + Wr[2][k] * dtz;

    /* ---- mean2D -> mean3D through the perspective projection ---- */
    {
      double hx = pm[0] * mean[0] + pm[4] * mean[1] + pm[8] * mean[2] + pm[12];
      double hy = pm[1] * mean[0] + pm[5] * mean[1] + pm[9] * mean[2] + pm[13];
      double hw = pm[3] * mean[0] + pm[7] * mean[1] + pm[11] * mean[2] + pm[15];
      double mw = 1.0 / (hw + 0.0000001);
      double g2x = pg->dL_dmean2D[2 * i + 0], g2y = pg->dL_dmean2D[2 * i + 1];
      double mul1 = hx * mw * mw, mul2 = hy * mw * mw;
      for (int k = 0; k < 3; k++)
        dmean[k] += (pm[4 * k + 0] * mw - pm[4 * k + 3] * mul1) * g2x + (pm[4 * k + 1] * mw - pm[4 * k + 3] * mul2) * g2y;
    }
    /* ---- depth = (mean @ V).z ---- */
    for (int k = 0; k < 3; k++) dmean[k] += (double)vm[4 * k + 2] * pg->dL_ddepth[i];

    /* ---- colour ---- */
    if (!s->colors_precomp && o->dL_dsh) {
      double d[3] = {mean[0] - s->campos[0], mean[1] - s->campos[1], mean[2] - s->campos[2]};
      double len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      double dirn[3] = {d[0] / len, d[1] / len, d[2] / len};
      double add[3];
      double dl[3] = {pg->dL_dcolor[3 * i], pg->dL_dcolor[3 * i + 1], pg->dL_dcolor[3 * i + 2]};
      sh_backward(s->D, s->M, s->shs + (size_t)3 * s->M * i, dirn, d, 1.0 / len, g->clamped + 3 * i, dl,
                  o->dL_dsh + (size_t)3 * s->M * i, add);
      for (int k = 0; k < 3; k++) dmean[k] += add[k];
    }
    for (int k = 0; k < 3; k++) o->dL_dmeans3D[3 * i + k] = (float)dmean[k];

    /* ---- cov3D -> scale, rotation ---- */
    if (!s->cov3D_precomp && o->dL_dscales && o->dL_drots) {
      const float* q = s->rotations + 4 * i;
      const float* sc = s->scales + 3 * i;
      double r = q[0], x = q[1], y = q[2], z = q[3];
      double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
                        {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                        {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
      double sv[3] = {(double)s->scale_modifier * sc[0], (double)s->scale_modifier * sc[1], (double)s->scale_modifier * sc[2]};
      /* Sigma = L L^T, L = R diag(sv).  dL/dL = 2 * dSigma_sym * L, where dSigma_sym has the
       * off-diagonal gradient split in halves */
      double G[3][3] = {{dS[0], 0.5 * dS[1], 0.5 * dS[2]}, {0.5 * dS[1], dS[3], 0.5 * dS[4]}, {0.5 * dS[2], 0.5 * dS[4], dS[5]}};
      double L[3][3], dLm[3][3];
      for (int a2 = 0; a2 < 3; a2++)
        for (int b2 = 0; b2 < 3; b2++) L[a2][b2] = R[a2][b2] * sv[b2];
      for (int a2 = 0; a2 < 3; a2++)
        for (int b2 = 0; b2 < 3; b2++) dLm[a2][b2] = 2.0 * (G[a2][0] * L[0][b2] + G[a2][1] * L[1][b2] + G[a2][2] * L[2][b2]);
      double dR[3][3];
      for (int b2 = 0; b2 < 3; b2++) {
        double ds = 0;
        for (int a2 = 0; a2 < 3; a2++) {
          ds += dLm[a2][b2] * R[a2][b2];
          dR[a2][b2] = dLm[a2][b2] * sv[b2];
        }
        o->dL_dscales[3 * i + b2] = (float)(ds * (double)s->scale_modifier);
      }
      double dq[4];
      dq[0] = 2 * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
      dq[1] = 2 * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2 * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2 * x * dR[2][2]);
      dq[2] = 2 * (-2 * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2 * y * dR[2][2]);
      dq[3] = 2 * (-2 * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2 * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
      for (int k = 0; k < 4; k++) o->dL_drots[4 * i + k] = (float)dq[k];
    }
  }
}

/* sizes so that Python can allocate without mirroring the struct layouts by hand */
int orc_sizeof_scene(void) { return (int)sizeof(OrcScene); }
int orc_sizeof_geom(void) { return (int)sizeof(OrcGeom); }
