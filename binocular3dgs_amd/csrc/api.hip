// extern "C" entry points of libb3gs_raster.so (see include/b3gs_raster.h for the contract and
// the reference call sites each one stands behind).  Host orchestration only: carve the opaque
// buffers, enqueue the kernels of preprocess.hip / binning.hip / render.hip on the caller's
// stream, perform the single num_rendered read-back of the synchronous forward.
#include "b3gs_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

thread_local char g_err[512] = "";
// bench-only timing sink: process-wide (autograd runs the backward on its own thread)
B3gsKernelTimes* g_timing = nullptr;
std::mutex g_timing_mu;

int fail(int code, const char* fmt, const char* detail) {
  snprintf(g_err, sizeof(g_err), fmt, detail);
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) return fail(B3GS_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
  } while (0)

int check_scene(const B3gsScene* sc) {
  if (!sc) return fail(B3GS_ERR_ARG, "%s", "scene is NULL");
  if (sc->P < 0 || sc->W <= 0 || sc->H <= 0) return fail(B3GS_ERR_ARG, "%s", "bad P/W/H");
  if (sc->P >= (1 << 24)) return fail(B3GS_ERR_ARG, "%s", "P must be below 2^24 (24-bit row offsets in the blend backward)");
  if (sc->D < 0 || sc->D > 3) return fail(B3GS_ERR_ARG, "%s", "SH degree must be 0..3");
  if ((sc->shs == nullptr) == (sc->colors_precomp == nullptr))
    return fail(B3GS_ERR_ARG, "%s", "provide exactly one of shs / colors_precomp");
  const bool has_sr = sc->scales != nullptr && sc->rotations != nullptr;
  if (has_sr == (sc->cov3D_precomp != nullptr))
    return fail(B3GS_ERR_ARG, "%s", "provide exactly one of (scales, rotations) / cov3D_precomp");
  if (sc->shs && sc->M < (sc->D + 1) * (sc->D + 1)) return fail(B3GS_ERR_ARG, "%s", "M too small for SH degree");
  if (((sc->W + B3GS_TILE - 1) / B3GS_TILE) > 65535 || ((sc->H + B3GS_TILE - 1) / B3GS_TILE) > 65535)
    return fail(B3GS_ERR_ARG, "%s", "image too large for the packed tile rect");
  if (!sc->background || !sc->viewmatrix || !sc->projmatrix || !sc->campos || (sc->P > 0 && (!sc->means3D || !sc->opacities)))
    return fail(B3GS_ERR_ARG, "%s", "NULL required tensor");
  return B3GS_OK;
}

SceneX wrap(const B3gsScene* sc) {
  SceneX x;
  x.sc = *sc;
  x.raw = B3gsRawParams{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  x.raw_mode = 0;
  x.tight = 0;
  return x;
}

int check_raw(const B3gsScene* sc, const B3gsRawParams* rp) {
  if (!sc || !rp) return fail(B3GS_ERR_ARG, "%s", "scene / raw params NULL");
  if (sc->P < 0 || sc->W <= 0 || sc->H <= 0) return fail(B3GS_ERR_ARG, "%s", "bad P/W/H");
  if (sc->P >= (1 << 24)) return fail(B3GS_ERR_ARG, "%s", "P must be below 2^24 (24-bit row offsets in the blend backward)");
  if (sc->D < 0 || sc->D > 3 || sc->M < (sc->D + 1) * (sc->D + 1)) return fail(B3GS_ERR_ARG, "%s", "bad SH degree / M");
  if (((sc->W + B3GS_TILE - 1) / B3GS_TILE) > 65535 || ((sc->H + B3GS_TILE - 1) / B3GS_TILE) > 65535)
    return fail(B3GS_ERR_ARG, "%s", "image too large for the packed tile rect");
  if (!sc->background || !sc->viewmatrix || !sc->projmatrix || !sc->campos) return fail(B3GS_ERR_ARG, "%s", "NULL camera tensor");
  if (sc->P > 0 && (!rp->xyz || !rp->features_dc || (sc->M > 1 && !rp->features_rest) || !rp->scaling || !rp->rotation ||
                    !rp->opacity))
    return fail(B3GS_ERR_ARG, "%s", "NULL raw parameter tensor");
  return B3GS_OK;
}

void blend_forward_one(const B3gsScene& sc, const GeomView& g, const BinView& b, const ImgView& im, float* out_color,
                       float* out_depth, float* out_alpha, hipStream_t s) {
  BlendBatch batch;
  batch.n = 1;
  batch.order = nullptr;
  batch.order_buf = nullptr;
  batch.cls_size = 0;
  batch.v[0] = b3gs_blend_view(sc, g, b, im);
  batch.v[0].out_color = out_color;
  batch.v[0].out_depth = out_depth;
  batch.v[0].out_alpha = out_alpha;
  b3gs_launch_blend_forward(batch, s);
}

BlendView blend_backward_view(const B3gsScene& sc, const GeomView& g, const BinView& b, const ImgView& im,
                              const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, float* m2d,
                              float* col, float* op, float* cov, uint32_t m2d_stride, uint32_t col_stride,
                              uint32_t op_stride, uint32_t cov_stride) {
  BlendView v = b3gs_blend_view(sc, g, b, im);
  v.dL_dcolor = dL_dcolor;
  v.dL_ddepth = dL_ddepth;
  v.dL_dalpha = dL_dalpha;
  v.dL_dmeans2D = m2d;
  v.dL_dcolors = col;
  v.dL_dopacity = op;
  v.dL_dcov3D = cov;
  v.m2d_stride = m2d_stride;
  v.col_stride = col_stride;
  v.op_stride = op_stride;
  v.cov_stride = cov_stride;
  return v;
}

// optional per-stage timing (bench only): events bracket each stage on the caller's stream.
// Nothing synchronises inside the call: the events are parked in a thread-local list and turned
// into milliseconds by b3gs_timing_collect() after the caller has synchronised the stream.
struct PendingStage {
  hipEvent_t a, b;
  int slot;  // 0 preprocess, 1 sort, 2 render fwd, 3 render bwd, 4 preprocess bwd
};
PendingStage g_pending[4096];
int g_npending = 0;

struct StageTimer {
  hipStream_t s;
  hipEvent_t prev = nullptr;
  bool on;
  explicit StageTimer(hipStream_t st) : s(st), on(g_timing != nullptr) {}
  // close the stage that started at the previous mark (slot < 0: just open a new stage)
  void mark(int slot) {
    if (!on) return;
    hipEvent_t e;
    (void)hipEventCreate(&e);
    (void)hipEventRecord(e, s);
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (slot >= 0 && prev && g_npending < 4096) {
      g_pending[g_npending++] = PendingStage{prev, e, slot};
    }
    prev = e;  // events are destroyed in b3gs_timing_collect (shared between adjacent stages)
  }
};

int debug_sync(const B3gsScene* sc, hipStream_t s, const char* what) {
  if (!sc->debug) return B3GS_OK;
  hipError_t e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "after %s: %s", what, hipGetErrorString(e));
    return B3GS_ERR_HIP;
  }
  return B3GS_OK;
}

}  // namespace

int b3gs_fail(int code, const char* what, const char* detail) {
  snprintf(g_err, sizeof(g_err), "%s%s%s", what ? what : "", (what && detail) ? ": " : "", detail ? detail : "");
  return code;
}

int b3gs_launch_status(const char* what) {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? B3GS_OK : b3gs_fail(B3GS_ERR_HIP, what, hipGetErrorString(e));
}

extern "C" {

int b3gs_abi_version(void) { return B3GS_ABI_VERSION; }
const char* b3gs_last_error(void) { return g_err; }
void b3gs_set_timing(B3gsKernelTimes* sink) { g_timing = sink; }

int b3gs_timing_collect(void) {
  // caller has synchronised the stream(s): resolve parked events into the sink
  std::lock_guard<std::mutex> lk(g_timing_mu);
  int n = g_npending;
  for (int i = 0; i < n; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_pending[i].a, g_pending[i].b) == hipSuccess && g_timing) {
      double* slots[5] = {&g_timing->preprocess_ms, &g_timing->sort_ms, &g_timing->render_fwd_ms,
                          &g_timing->render_bwd_ms, &g_timing->preprocess_bwd_ms};
      *slots[g_pending[i].slot] += (double)ms;
      if (g_pending[i].slot == 2) g_timing->calls++;
    }
  }
  // an event may be the end of one stage and the start of the next: destroy each once
  for (int i = 0; i < n; i++) {
    bool a_shared = i > 0 && g_pending[i - 1].b == g_pending[i].a;
    if (!a_shared) (void)hipEventDestroy(g_pending[i].a);
    (void)hipEventDestroy(g_pending[i].b);
  }
  g_npending = 0;
  return n;
}

size_t b3gs_geometry_bytes(int32_t P) { return b3gs_geom_view(nullptr, P, nullptr); }
size_t b3gs_image_bytes(int32_t W, int32_t H) { return b3gs_img_view(nullptr, W, H, nullptr); }
size_t b3gs_binning_bytes(int32_t P, int64_t num_rendered) { return b3gs_bin_view(nullptr, P, num_rendered, nullptr); }

int b3gs_forward(const B3gsScene* sc, b3gs_alloc_fn geometry_alloc, void* geometry_user, b3gs_alloc_fn binning_alloc,
                 void* binning_user, b3gs_alloc_fn image_alloc, void* image_user, float* out_color, float* out_depth,
                 float* out_alpha, int32_t* radii, int32_t* host_num_rendered, b3gs_stream_t stream) {
  int rc = check_scene(sc);
  if (rc) return rc;
  if (!geometry_alloc || !binning_alloc || !image_alloc || !out_color || !out_depth || !out_alpha ||
      (sc->P > 0 && !radii))
    return fail(B3GS_ERR_ARG, "%s", "NULL output / allocator");
  hipStream_t s = (hipStream_t)stream;

  char* gbuf = geometry_alloc(geometry_user, b3gs_geometry_bytes(sc->P));
  char* ibuf = image_alloc(image_user, b3gs_image_bytes(sc->W, sc->H));
  if (!gbuf || !ibuf) return fail(B3GS_ERR_ALLOC, "%s", "geometry/image allocation failed");
  GeomView g;
  ImgView im;
  b3gs_geom_view(gbuf, sc->P, &g);
  b3gs_img_view(ibuf, sc->W, sc->H, &im);

  StageTimer tm(s);
  tm.mark(-1);
  {
    const SceneX sx = wrap(sc);
    PreBatch pb;
    pb.n = 1;
    pb.raw_mode = sx.raw_mode;
    pb.tight = sx.tight;
    pb.raw = sx.raw;
    pb.sc[0] = sx.sc;
    pb.out[0] = b3gs_pre_out(sx.sc, g, im, radii);
    b3gs_launch_preprocess(pb, s);
  }
  if ((rc = debug_sync(sc, s, "preprocess"))) return rc;
  tm.mark(0);
  BinJob job{sc->W, sc->H, g, BinView{}, im, 0, nullptr, -1, nullptr, nullptr, 1};
  b3gs_launch_depth_order_batch(sc->P, 1, &job, s);
  if ((rc = debug_sync(sc, s, "depth sort + scan"))) return rc;

  // the one blocking read-back of the forward: N sizes the binning buffer
  uint32_t hdr[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(hdr, g.header, sizeof(hdr), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const int64_t N = (int64_t)hdr[0];
  if (host_num_rendered) *host_num_rendered = (int32_t)N;

  char* bbuf = binning_alloc(binning_user, b3gs_binning_bytes(sc->P, N));
  if (!bbuf) return fail(B3GS_ERR_ALLOC, "%s", "binning allocation failed");
  BinView b;
  b3gs_bin_view(bbuf, sc->P, N, &b);

  job.b = b;
  job.n_bound = N;
  b3gs_launch_tile_lists_batch(sc->P, 1, &job, s);
  if ((rc = debug_sync(sc, s, "binning"))) return rc;
  tm.mark(1);
  blend_forward_one(*sc, g, b, im, out_color, out_depth, out_alpha, s);
  if ((rc = debug_sync(sc, s, "render forward"))) return rc;
  tm.mark(2);
  HIP_TRY(hipGetLastError());
  return B3GS_OK;
}

static int forward_capacity_impl(const SceneX& sx, char* geometry, char* binning, int64_t binning_capacity, char* image,
                                 float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                                 int32_t* device_num_rendered, int phases, hipStream_t s) {
  const B3gsScene* sc = &sx.sc;
  if (!geometry || !binning || !image || (sc->P > 0 && !radii) || binning_capacity <= 0 ||
      binning_capacity > 0xFFFFFFFFll || ((phases & 2) && (!out_color || !out_depth || !out_alpha)))
    return fail(B3GS_ERR_ARG, "%s", "NULL buffer or bad capacity");
  GeomView g;
  ImgView im;
  BinView b;
  b3gs_geom_view(geometry, sc->P, &g);
  b3gs_img_view(image, sc->W, sc->H, &im);
  b3gs_bin_view(binning, sc->P, binning_capacity, &b);
  StageTimer tm(s);
  tm.mark(-1);
  if (phases & 1) {
    PreBatch pb;
    pb.n = 1;
    pb.raw_mode = sx.raw_mode;
    pb.tight = sx.tight;
    pb.raw = sx.raw;
    pb.sc[0] = sx.sc;
    pb.out[0] = b3gs_pre_out(sx.sc, g, im, radii);
    b3gs_launch_preprocess(pb, s);
    tm.mark(0);
    // every binning kernel clamps to min(N, capacity); an overflowing view renders a truncated
    // list, which the caller detects from *device_num_rendered > capacity and repeats
    BinJob job{sc->W, sc->H, g, b, im, binning_capacity, device_num_rendered, -1, nullptr, nullptr, 1};
    b3gs_launch_binning_batch(sc->P, 1, &job, s);
    tm.mark(1);
  }
  if (phases & 2) {
    blend_forward_one(*sc, g, b, im, out_color, out_depth, out_alpha, s);
    tm.mark(2);
  }
  HIP_TRY(hipGetLastError());
  return B3GS_OK;
}

int b3gs_forward_capacity(const B3gsScene* sc, char* geometry, char* binning, int64_t binning_capacity, char* image,
                          float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                          int32_t* device_num_rendered, b3gs_stream_t stream) {
  int rc = check_scene(sc);
  if (rc) return rc;
  return forward_capacity_impl(wrap(sc), geometry, binning, binning_capacity, image, out_color, out_depth, out_alpha,
                               radii, device_num_rendered, 3, (hipStream_t)stream);
}

int b3gs_forward_raw(const B3gsScene* view, const B3gsRawParams* params, char* geometry, char* binning,
                     int64_t binning_capacity, char* image, float* out_color, float* out_depth, float* out_alpha,
                     int32_t* radii, int32_t* device_num_rendered, int phases, b3gs_stream_t stream) {
  int rc = check_raw(view, params);
  if (rc) return rc;
  SceneX sx;
  sx.sc = *view;
  sx.raw = *params;
  sx.raw_mode = 1;
  sx.tight = 1;       // (the single-view entry point has no switch: b3gs_forward_raw_batch(1, ...) does)
  return forward_capacity_impl(sx, geometry, binning, binning_capacity, image, out_color, out_depth, out_alpha, radii,
                               device_num_rendered, phases, (hipStream_t)stream);
}

int b3gs_forward_raw_batch(int32_t nviews, const B3gsForwardView* views, const B3gsRawParams* params, int phases,
                           b3gs_stream_t stream) {
  if (nviews <= 0 || nviews > B3GS_MAX_FUSED_VIEWS || !views) return fail(B3GS_ERR_ARG, "%s", "nviews must be 1..8");
  hipStream_t s = (hipStream_t)stream;
  PreBatch pb;
  BinJob jobs[B3GS_MAX_FUSED_VIEWS];
  BlendBatch bb;
  bb.order = nullptr;
  bb.order_buf = nullptr;
  bb.cls_size = 0;
  pb.n = bb.n = nviews;
  pb.raw_mode = 1;
  pb.tight = views[0].reference_binning ? 0 : 1;      // (B3gsForwardView::reference_binning, ABI 10)
  for (int k = 0; k < nviews; k++) {
    const B3gsForwardView& fv = views[k];
    int rc = check_raw(fv.view, params);
    if (rc) return rc;
    const B3gsScene& sc = *fv.view;
    if (sc.P != views[0].view->P || sc.M != views[0].view->M || sc.D != views[0].view->D ||
        sc.scale_modifier != views[0].view->scale_modifier)
      return fail(B3GS_ERR_ARG, "%s", "views of one batch must share P, M, D, scale_modifier");
    if (!fv.geometry || !fv.binning || !fv.image || (sc.P > 0 && !fv.radii) || fv.binning_capacity <= 0 ||
        fv.binning_capacity > 0xFFFFFFFFll || ((phases & 2) && (!fv.out_color || !fv.out_depth || !fv.out_alpha)))
      return fail(B3GS_ERR_ARG, "%s", "NULL buffer or bad capacity");
    if (fv.depth_order_from != -1 &&
        (fv.depth_order_from < 0 || fv.depth_order_from >= nviews || views[fv.depth_order_from].depth_order_from != -1))
      return fail(B3GS_ERR_ARG, "%s", "depth_order_from must name a view of the batch that sorts its own keys");
    if (fv.depth_order_from != -1 && fv.hint_trusted && (fv.depth_order_from != k - 1 || !fv.overflow_flag))
      return fail(B3GS_ERR_ARG, "%s", "a verified shared depth order (depth_order_from with hint_trusted) needs the donor to be "
                                      "the previous view of the batch and an overflow word");
    if (fv.depth_order_hint && (fv.depth_order_from != -1 || !fv.hint_mismatch || fv.depth_order_hint == fv.geometry ||
                                (fv.hint_trusted && !fv.overflow_flag)))
      return fail(B3GS_ERR_ARG, "%s", "depth_order_hint needs depth_order_from == -1, a hint_mismatch word, another buffer "
                                      "(and an overflow word when trusted)");
    GeomView g;
    ImgView im;
    BinView b;
    b3gs_geom_view(fv.geometry, sc.P, &g);
    b3gs_img_view(fv.image, sc.W, sc.H, &im);
    b3gs_bin_view(fv.binning, sc.P, fv.binning_capacity, &b);
    pb.sc[k] = sc;
    pb.out[k] = b3gs_pre_out(sc, g, im, fv.radii);
    pb.out[k].visible = fv.visible;
    jobs[k] = BinJob{sc.W, sc.H, g, b, im, fv.binning_capacity, fv.device_num_rendered, fv.depth_order_from, nullptr,
                     nullptr, 1, 0, fv.high_water, fv.overflow_flag};
    if (fv.depth_order_hint && sc.P > 0) {   // ABI 7: an earlier forward's depth order, adopted while every key is equal
      GeomView hg;
      b3gs_geom_view(const_cast<char*>(fv.depth_order_hint), sc.P, &hg);
      pb.out[k].hint_key = hg.depth_key;
      pb.out[k].hint_word = fv.hint_mismatch;
      jobs[k].hint_sval = hg.sval[0];
      jobs[k].hint_skey = hg.skey[0];
      jobs[k].hint_word = fv.hint_mismatch;
      jobs[k].hint_trusted = fv.hint_trusted ? 1 : 0;
      pb.out[k].hint_fatal = fv.hint_trusted ? fv.overflow_flag : nullptr;
    }
    bb.v[k] = b3gs_blend_view(sc, g, b, im);
    bb.v[k].open_rows = im.open_rows;
    bb.v[k].out_color = fv.out_color;
    bb.v[k].out_depth = fv.out_depth;
    bb.v[k].out_alpha = fv.out_alpha;
  }
  // binocular pairs: the borrower's tile rects go to the odd slots of its donor's [P][2] rect array, so that
  // the one random access of the binning (rect in depth order) is a single 16-byte load for both views
  bool has_partner[B3GS_MAX_FUSED_VIEWS] = {};
  for (int k = 0; k < nviews; k++) {
    const int d = views[k].depth_order_from;
    if (d < 0 || has_partner[d]) continue;
    has_partner[d] = true;
    pb.out[d].rect_stride = 2;
    pb.out[k].rect = pb.out[d].rect + 1;
    pb.out[k].rect_stride = 2;
    if (d == k - 1) {   // adjacent in the batch: the projection kernel writes both rects with one 16-byte store
      pb.out[d].rect_role = 1;
      pb.out[k].rect_role = 2;
      if (views[k].hint_trusted) pb.out[k].pair_fatal = views[k].overflow_flag;   // ... and compares the two depth keys
    }
  }
  for (int k = 0; k < nviews; k++) {
    jobs[k].rect = pb.out[k].rect;
    jobs[k].rect_stride = pb.out[k].rect_stride;
  }
  // the forward follows the longest-tile-first order of the previous backward, if the image buffer can hold one
  bb.order_buf = views[0].fresh_image ? nullptr : jobs[0].im.order;
  pb.raw = *params;
  // two-round binning: the same K1 for every view of the batch; needs packed instance words and one tile-sort depth
  const int32_t P = views[0].view->P;
  int32_t K1 = 0;
  {
    float frac = views[0].seg1_fraction;
    static const char* const frac_env = getenv("B3GS_SEG1_FRAC");   // (A/B switches: read once per process)
    if (frac_env) frac = (float)atof(frac_env);
    if (frac > 0.0f && frac < 1.0f && P > 1 && !views[0].fresh_image) {
      K1 = (int32_t)((double)frac * (double)P + 0.999999);
      K1 = K1 < 1 ? 1 : K1;
      K1 = (int32_t)((((int64_t)K1 + 4095) / 4096) * 4096);   // whole 4096-Gaussian tiles of the depth order (binning.hip)
      for (int k = 0; k < nviews && K1; k++) {
        const B3gsScene& sc = *views[k].view;
        if ((b3gs_tile_bits(sc.W, sc.H) + 7) / 8 != (b3gs_tile_bits(views[0].view->W, views[0].view->H) + 7) / 8) K1 = 0;
      }
      if (K1 >= P) K1 = 0;
    }
  }
  // three-pass depth sort on 27-bit keys: only with the caller's overflow word to report a key outside the span to
  static const bool key27_env = getenv("B3GS_NO_KEY27") == nullptr;   // (A/B switch)
  bool key27 = key27_env;
  for (int k = 0; k < nviews; k++)
    key27 = key27 && views[k].depth_key_bits == 27 && views[k].overflow_flag != nullptr && !views[k].view->prefiltered;
  for (int k = 0; k < nviews; k++) {
    jobs[k].key_bits = key27 ? 27 : 0;
    pb.out[k].span_flag = key27 ? views[k].overflow_flag : nullptr;
    bb.v[k].z_base = key27 ? 0x3E4CCCCDu : 0u;
  }
  for (int k = 0; k < nviews; k++) {
    jobs[k].K1 = K1;
    if (!K1) continue;
    // the tiles the previous two-round forward into this image buffer left unterminated are predicted open now
    pb.out[k].pred_rows = jobs[k].im.pred_rows;
    bb.v[k].pred_rows = jobs[k].im.pred_rows;
    bb.v[k].pred_next = jobs[k].im.pred_next;
    const int d = views[k].depth_order_from;
    bb.v[k].z_clear = (d >= 0 ? jobs[d].g : jobs[k].g).skey[0] + (size_t)(3 * (int64_t)K1 / 4);   // sorted depth keys
  }
  StageTimer tm(s);
  tm.mark(-1);
  if (phases & 1) {
    b3gs_launch_preprocess(pb, s);
    tm.mark(0);
    b3gs_launch_binning_batch(P, nviews, jobs, s);
    tm.mark(1);
  }
  if (phases & 2) {
    b3gs_launch_blend_forward(bb, s);
    tm.mark(2);
    if (K1) {   // second round: the rest of the depth order, into the tiles segment 1 did not finish
      b3gs_launch_round2_batch(P, nviews, jobs, s);
      tm.mark(1);
      for (int k = 0; k < nviews; k++) bb.v[k].round = 1;
      b3gs_launch_blend_forward(bb, s);
      tm.mark(2);
    }
  }
  HIP_TRY(hipGetLastError());
  return B3GS_OK;
}

// Raw-mode scratch: one row of B3GS_SCRATCH_ROW floats (40 bytes) per Gaussian -- conic xx, xy, yy, depth |
// mean2D x, y | colour r, g, b | opacity -- so the 10-lane atomic of the blend backward lands on one
// row (one or two cache lines) instead of four arrays, and the per-Gaussian pass reads five float2.
size_t b3gs_backward_scratch_floats(int32_t P) { return (size_t)B3GS_SCRATCH_ROW * (size_t)(P > 0 ? P : 0); }

int b3gs_backward_raw(const B3gsScene* view, const B3gsRawParams* params, const int32_t* radii, const char* geometry,
                      const char* binning, const char* image, const float* dL_dcolor, const float* dL_ddepth,
                      const float* dL_dalpha, float* scratch, const B3gsRawGrads* grads, float* dL_dmeans2D,
                      int phases, b3gs_stream_t stream) {
  int rc = check_raw(view, params);
  if (rc) return rc;
  if (view->P == 0) return B3GS_OK;
  if (!radii || !geometry || !binning || !image || !dL_dcolor || !scratch || !grads || !grads->xyz ||
      !grads->features_dc || (view->M > 1 && !grads->features_rest) || !grads->scaling || !grads->rotation ||
      !grads->opacity)
    return fail(B3GS_ERR_ARG, "%s", "NULL required tensor in backward_raw");
  hipStream_t s = (hipStream_t)stream;
  SceneX sx;
  sx.sc = *view;
  sx.raw = *params;
  sx.raw_mode = 1;
  sx.tight = 0;  // irrelevant in the backward
  GeomView g;
  ImgView im;
  BinView b;
  b3gs_geom_view(const_cast<char*>(geometry), view->P, &g);
  b3gs_img_view(const_cast<char*>(image), view->W, view->H, &im);
  b3gs_bin_view(const_cast<char*>(binning), view->P, 1, &b);  // only val[0] (offset 0) is read
  // scratch (zero on entry, left zero on exit): rows of B3GS_SCRATCH_ROW floats, see b3gs_backward_scratch_floats
  const size_t P = (size_t)view->P;
  (void)P;
  StageTimer tm(s);
  tm.mark(-1);
  if (phases & 1) {
    BlendBatch batch;
    batch.n = 1;
    batch.order = nullptr;
    batch.order_buf = im.order;
    batch.v[0] = blend_backward_view(sx.sc, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, scratch + 4, scratch + 6,
                                     scratch + 9, scratch, B3GS_SCRATCH_ROW, B3GS_SCRATCH_ROW, B3GS_SCRATCH_ROW,
                                     B3GS_SCRATCH_ROW);
    b3gs_launch_blend_backward(batch, s);
    tm.mark(3);
  }
  if (phases & 2) {
    if (!(phases & 1)) tm.mark(-1);
    b3gs_launch_preprocess_backward(sx, g, radii, nullptr, nullptr, nullptr, nullptr, scratch, nullptr, nullptr, nullptr,
                                    grads, dL_dmeans2D, s);
    tm.mark(4);
  }
  HIP_TRY(hipGetLastError());
  return B3GS_OK;
}

int b3gs_backward(const B3gsScene* sc, int32_t num_rendered, const int32_t* radii, const char* geometry,
                  const char* binning, const char* image, const float* dL_dcolor, const float* dL_ddepth,
                  const float* dL_dalpha, float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
                  float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations,
                  b3gs_stream_t stream) {
  int rc = check_scene(sc);
  if (rc) return rc;
  if (sc->P == 0) return B3GS_OK;
  if (!radii || !geometry || !image || !dL_dcolor || !dL_dmeans2D || !dL_dcolors || !dL_dopacity || !dL_dmeans3D ||
      !dL_dcov3D)
    return fail(B3GS_ERR_ARG, "%s", "NULL required tensor in backward");
  if (sc->shs && !dL_dsh) return fail(B3GS_ERR_ARG, "%s", "dL_dsh required when shs given");
  if (sc->scales && (!dL_dscales || !dL_drotations)) return fail(B3GS_ERR_ARG, "%s", "dL_dscales/dL_drotations required");
  if (num_rendered != 0 && !binning) return fail(B3GS_ERR_ARG, "%s", "binning buffer is NULL");
  hipStream_t s = (hipStream_t)stream;
  GeomView g;
  ImgView im;
  BinView b;
  b3gs_geom_view(const_cast<char*>(geometry), sc->P, &g);
  b3gs_img_view(const_cast<char*>(image), sc->W, sc->H, &im);
  // the binning views only depend on the capacity through the position of hist, which the
  // backward never touches: key/val start at fixed offsets for a given element count
  b3gs_bin_view(const_cast<char*>(binning), sc->P, num_rendered < 0 ? 1 : num_rendered, &b);

  const size_t P = (size_t)sc->P;
  StageTimer tm(s);
  tm.mark(-1);
  // accumulation target of the blend backward: one 40-byte row per Gaussian in the geometry buffer's scratch region
  // (a single 10-lane atomic per (quadrant, Gaussian) lands on one row; four separate output arrays cost 26 % more)
  float* rows = g.bwd_rows;
  HIP_TRY(hipMemsetAsync(rows, 0, P * B3GS_SCRATCH_ROW * sizeof(float), s));
  if (num_rendered != 0) {
    BlendBatch batch;
    batch.n = 1;
    batch.order = nullptr;
    batch.order_buf = im.order;
    batch.v[0] = blend_backward_view(*sc, g, b, im, dL_dcolor, dL_ddepth, dL_dalpha, rows + 4, rows + 6, rows + 9, rows,
                                     B3GS_SCRATCH_ROW, B3GS_SCRATCH_ROW, B3GS_SCRATCH_ROW, B3GS_SCRATCH_ROW);
    b3gs_launch_blend_backward(batch, s);
  }
  if ((rc = debug_sync(sc, s, "render backward"))) return rc;
  tm.mark(3);
  b3gs_launch_preprocess_backward(wrap(sc), g, radii, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D,
                                  dL_dsh, dL_dscales, dL_drotations, nullptr, nullptr, s);
  if ((rc = debug_sync(sc, s, "preprocess backward"))) return rc;
  tm.mark(4);
  HIP_TRY(hipGetLastError());
  return B3GS_OK;
}

static int batch_views_ok(int32_t nviews, const B3gsBlendView* views) {
  if (nviews <= 0 || nviews > B3GS_MAX_FUSED_VIEWS || !views)
    return fail(B3GS_ERR_ARG, "%s", "bad view count (1..8) or NULL view table");
  for (int k = 0; k < nviews; k++)
    if (!views[k].view || !views[k].geometry || !views[k].binning || !views[k].image || views[k].view->W <= 0 ||
        views[k].view->H <= 0 || !views[k].view->background)
      return fail(B3GS_ERR_ARG, "%s", "NULL state buffer in a batched view");
  return B3GS_OK;
}

int b3gs_blend_forward_batch(int32_t nviews, const B3gsBlendView* views, b3gs_stream_t stream) {
  int rc = batch_views_ok(nviews, views);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  BlendBatch batch;
  batch.n = nviews;
  batch.order = nullptr;
  batch.order_buf = nullptr;
  batch.cls_size = 0;
  for (int k = 0; k < nviews; k++) {
    const B3gsBlendView& bv = views[k];
    if (!bv.out_color || !bv.out_depth || !bv.out_alpha) return fail(B3GS_ERR_ARG, "%s", "NULL output image");
    GeomView g;
    ImgView im;
    BinView b;
    b3gs_geom_view(const_cast<char*>(bv.geometry), bv.view->P, &g);
    b3gs_img_view(const_cast<char*>(bv.image), bv.view->W, bv.view->H, &im);
    b3gs_bin_view(const_cast<char*>(bv.binning), bv.view->P, bv.binning_capacity > 0 ? bv.binning_capacity : 1, &b);
    batch.v[k] = b3gs_blend_view(*bv.view, g, b, im);
    batch.v[k].round = 2;   // the state may come from a two-round forward: every tile walks segment 1 + segment 2
    batch.v[k].out_color = bv.out_color;
    batch.v[k].out_depth = bv.out_depth;
    batch.v[k].out_alpha = bv.out_alpha;
  }
  StageTimer tm(s);
  tm.mark(-1);
  b3gs_launch_blend_forward(batch, s);
  tm.mark(2);
  HIP_TRY(hipGetLastError());
  return B3GS_OK;
}

int b3gs_blend_backward_batch(int32_t nviews, const B3gsBlendView* views, b3gs_stream_t stream) {
  int rc = batch_views_ok(nviews, views);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  BlendBatch batch;
  batch.n = nviews;
  batch.order = nullptr;
  batch.order_buf = nullptr;
  batch.cls_size = 0;
  for (int k = 0; k < nviews; k++) {
    const B3gsBlendView& bv = views[k];
    if (!bv.dL_dcolor || !bv.scratch) return fail(B3GS_ERR_ARG, "%s", "NULL dL_dcolor / scratch");
    GeomView g;
    ImgView im;
    BinView b;
    b3gs_geom_view(const_cast<char*>(bv.geometry), bv.view->P, &g);
    b3gs_img_view(const_cast<char*>(bv.image), bv.view->W, bv.view->H, &im);
    if (k == 0) batch.order_buf = im.order;
    b3gs_bin_view(const_cast<char*>(bv.binning), bv.view->P, bv.binning_capacity > 0 ? bv.binning_capacity : 1, &b);
    batch.v[k] = blend_backward_view(*bv.view, g, b, im, bv.dL_dcolor, bv.dL_ddepth, bv.dL_dalpha, bv.scratch + 4,
                                     bv.scratch + 6, bv.scratch + 9, bv.scratch, B3GS_SCRATCH_ROW, B3GS_SCRATCH_ROW,
                                     B3GS_SCRATCH_ROW, B3GS_SCRATCH_ROW);
  }
  StageTimer tm(s);
  tm.mark(-1);
  b3gs_launch_blend_backward(batch, s);
  tm.mark(3);
  HIP_TRY(hipGetLastError());
  return B3GS_OK;
}

int b3gs_backward_raw_accumulate(int32_t nviews, const B3gsFusedView* views, const B3gsRawParams* params,
                                 const B3gsRawGrads* grads, int32_t overwrite, const B3gsDensifyStats* stats,
                                 b3gs_stream_t stream) {
  const int32_t P = (nviews > 0 && views && views[0].view) ? views[0].view->P : 0;
  return b3gs_backward_raw_accumulate_range(nviews, views, params, grads, overwrite, stats, 0, P, stream);
}

int b3gs_backward_raw_accumulate_range(int32_t nviews, const B3gsFusedView* views, const B3gsRawParams* params,
                                       const B3gsRawGrads* grads, int32_t overwrite, const B3gsDensifyStats* stats,
                                       int32_t first, int32_t count, b3gs_stream_t stream) {
  if (nviews <= 0 || nviews > B3GS_MAX_FUSED_VIEWS || !views || !params || !grads)
    return fail(B3GS_ERR_ARG, "%s", "bad view count / NULL argument (at most 8 views per call)");
  const B3gsScene* v0 = views[0].view;
  int rc = check_raw(v0, params);
  if (rc) return rc;
  if (v0->P == 0 || count == 0) return B3GS_OK;
  if (first < 0 || count < 0 || (int64_t)first + count > v0->P) return fail(B3GS_ERR_ARG, "%s", "bad Gaussian range");
  if (grads->touched_rows && overwrite && (first & 63))
    return fail(B3GS_ERR_ARG, "%s", "sparse-row gradients (touched_rows) need a range that starts at a multiple of 64");
  if (!grads->xyz || !grads->features_dc || (v0->M > 1 && !grads->features_rest) || !grads->scaling ||
      !grads->rotation || !grads->opacity)
    return fail(B3GS_ERR_ARG, "%s", "NULL gradient buffer");
  B3gsViewRef refs[B3GS_MAX_FUSED_VIEWS];
  uint32_t *list = nullptr, *counts = nullptr;
  static const bool no_staged = getenv("B3GS_NO_STAGED") != nullptr;   // (A/B switch, read once: every visible row is read)
  for (int k = 0; k < nviews; k++) {
    const B3gsFusedView& fv = views[k];
    if (!fv.view || !fv.radii || !fv.geometry || !fv.scratch) return fail(B3GS_ERR_ARG, "%s", "NULL view state");
    if (fv.view->P != v0->P || fv.view->M != v0->M || fv.view->D != v0->D || fv.view->scale_modifier != v0->scale_modifier)
      return fail(B3GS_ERR_ARG, "%s", "views of one call must share P, M, D and scale_modifier");
    GeomView g;
    b3gs_geom_view(const_cast<char*>(fv.geometry), v0->P, &g);
    if (k == 0) { list = g.skey[1]; counts = g.skey[0]; }   // idle after the forward's depth sort
    refs[k] = B3gsViewRef{fv.view->W, fv.view->H, fv.view->tan_fovx, fv.view->tan_fovy, fv.view->viewmatrix,
                          fv.view->projmatrix, fv.view->campos, fv.radii, reinterpret_cast<const float*>(g.rec), fv.scratch, fv.dL_dmeans2D,
                          fv.densify_stats, no_staged ? nullptr : g.staged, g.header + B3GS_GEOM_EPOCH};
  }
  if (stats && (!stats->xyz_gradient_accum || !stats->denom || !stats->max_radii2D))
    return fail(B3GS_ERR_ARG, "%s", "densify stats need all three arrays");
  hipStream_t s = (hipStream_t)stream;
  StageTimer tm(s);
  tm.mark(-1);
  b3gs_launch_accumulate_views(*v0, *params, nviews, refs, *grads, overwrite, stats, first, count, list, counts, s);
  tm.mark(4);
  HIP_TRY(hipGetLastError());
  return B3GS_OK;
}

int b3gs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present, b3gs_stream_t stream) {
  (void)projmatrix;
  if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return fail(B3GS_ERR_ARG, "%s", "bad arguments");
  b3gs_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return B3GS_OK;
}

int b3gs_debug_views(int32_t P, int32_t W, int32_t H, int64_t num_rendered, const char* geometry, const char* binning,
                     const char* image, B3gsDebugViews* out) {
  if (!out || !geometry || !image) return fail(B3GS_ERR_ARG, "%s", "NULL argument");
  GeomView g;
  ImgView im;
  BinView b;
  b3gs_geom_view(const_cast<char*>(geometry), P, &g);
  b3gs_img_view(const_cast<char*>(image), W, H, &im);
  memset(out, 0, sizeof(*out));
  out->tiles_touched = g.tiles_touched;
  out->depths = reinterpret_cast<const float*>(g.depth_key);
  out->records = reinterpret_cast<const float*>(g.rec);
  if (binning) {
    b3gs_bin_view(const_cast<char*>(binning), P, num_rendered, &b);
    out->point_list = b.val[0];
    out->packed_idx_bits = b3gs_packed_idx_bits(P, W, H);
    out->tile_ids = out->packed_idx_bits >= 0 ? nullptr : b.key[0];
  }
  out->ranges = reinterpret_cast<const uint32_t*>(im.ranges);
  out->ranges2 = reinterpret_cast<const uint32_t*>(im.ranges2);
  out->counts = g.header;
  if (binning) out->point_list2 = b.val[0];   // + N1 elements (counts[0])
  out->final_T = im.final_T;
  out->n_contrib = im.n_contrib;
  return B3GS_OK;
}

}  // extern "C"
