/*
 * b3gs_raster.h -- C ABI of libb3gs_raster.so, the MI355X-native differentiable Gaussian
 * rasterizer that sits behind the reference's `diff_gaussian_rasterization._C` extension.
 *
 * Boundary it replaces (reference = hanl2010/Binocular3DGS; the extension's own source is an
 * un-vendored submodule, .gitmodules:1-3, so the citations are the reference's call sites):
 *   gaussian_renderer/__init__.py:36-49   GaussianRasterizationSettings  -> B3gsScene scalars/matrices
 *   gaussian_renderer/__init__.py:85-93   rasterizer(means3D=..., ...)   -> b3gs_forward()
 *   train.py:149 (total_loss.backward())  _RasterizeGaussians.backward   -> b3gs_backward()
 *   (upstream `_C.mark_visible`, unused by the reference)                -> b3gs_mark_visible()
 *
 * Rules of the ABI:
 *   - plain C: pointers, sizes, a stream handle; no torch / C++ types cross it
 *   - every pointer is a DEVICE pointer unless the name says `host_`
 *   - the caller owns all memory.  The three opaque state buffers (geometry / binning / image)
 *     are obtained through caller-supplied allocation callbacks, exactly like the upstream
 *     extension asks torch to resize three uint8 tensors; they must stay alive until
 *     b3gs_backward() for the same view has been enqueued
 *   - all work is enqueued on `stream`; b3gs_forward() blocks the calling host thread once
 *     (to read back num_rendered) -- b3gs_forward_capacity() is the sync-free variant
 *   - every function returns B3GS_OK or a negative B3gsStatus and never throws
 *   - no global mutable state: re-entrant per stream / per device (one process per GPU)
 */
#ifndef B3GS_RASTER_H
#define B3GS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B3GS_ABI_VERSION 10
#define B3GS_TILE 16 /* 16x16-pixel tiles: the binning granularity (bit-exact with the oracle) */

typedef enum B3gsStatus {
  B3GS_OK = 0,
  B3GS_ERR_ARG = -1,      /* inconsistent arguments (exactly one of shs/colors_precomp, ...) */
  B3GS_ERR_ALLOC = -2,    /* an allocation callback returned NULL */
  B3GS_ERR_HIP = -3,      /* a HIP runtime call failed; see b3gs_last_error() */
  B3GS_ERR_CAPACITY = -4, /* b3gs_forward_capacity: binning capacity too small */
  B3GS_ERR_NO_DEVICE = -5
} B3gsStatus;

typedef void* b3gs_stream_t; /* hipStream_t */

/* Allocation callback: return a device pointer to at least `bytes` bytes, 256-byte aligned,
 * or NULL.  Mirrors the resize-a-uint8-tensor lambdas of the upstream torch binding. */
typedef char* (*b3gs_alloc_fn)(void* user, size_t bytes);

/* One view of one Gaussian cloud.  Field meaning = the 12 settings of
 * gaussian_renderer/__init__.py:36-49 plus the 8 tensors of :85-93. */
typedef struct B3gsScene {
  int32_t P;           /* number of Gaussians (< 2^24) */
  int32_t D;           /* active SH degree, 0..3 (raster_settings.sh_degree) */
  int32_t M;           /* SH coefficients per channel stored in `shs` (0 when shs == NULL) */
  int32_t W, H;        /* image_width, image_height */
  float tan_fovx, tan_fovy;
  float scale_modifier;
  int32_t prefiltered; /* reference always passes False */
  int32_t debug;       /* sync + check after every kernel */
  const float* background;     /* [3] */
  const float* means3D;        /* [P,3] */
  const float* shs;            /* [P,M,3] or NULL */
  const float* colors_precomp; /* [P,3]   or NULL  (exactly one of shs / colors_precomp) */
  const float* opacities;      /* [P,1] */
  const float* scales;         /* [P,3]   or NULL */
  const float* rotations;      /* [P,4] (w,x,y,z), used as given  or NULL */
  const float* cov3D_precomp;  /* [P,6] (xx,xy,xz,yy,yz,zz) or NULL (exactly one of cov3D / scales+rotations) */
  const float* viewmatrix;     /* [4,4] row-vector convention: [x y z 1] @ viewmatrix (scene/cameras.py:55) */
  const float* projmatrix;     /* [4,4] full projection, same convention (scene/cameras.py:57) */
  const float* campos;         /* [3] */
} B3gsScene;

/* Sizes of the opaque buffers (bytes).  geometry depends on P, image on W*H, binning on N. */
size_t b3gs_geometry_bytes(int32_t P);
size_t b3gs_image_bytes(int32_t W, int32_t H);
size_t b3gs_binning_bytes(int32_t P, int64_t num_rendered);

/* Forward: colour [3,H,W], depth [1,H,W] (= sum z a T, un-normalised), alpha [1,H,W] (= sum a T),
 * radii [P] int32 (0 = culled).  *host_num_rendered receives N (tile instances). */
int b3gs_forward(const B3gsScene* scene,
                 b3gs_alloc_fn geometry_alloc, void* geometry_user,
                 b3gs_alloc_fn binning_alloc, void* binning_user,
                 b3gs_alloc_fn image_alloc, void* image_user,
                 float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                 int32_t* host_num_rendered, b3gs_stream_t stream);

/* Sync-free forward for callers that keep persistent scratch (the build's own training step):
 * all three buffers are pre-sized by the caller; binning has room for `binning_capacity`
 * instances.  N is written to *device_num_rendered (device int32) and nothing is read back.
 * If N > binning_capacity the images are left untouched and *device_num_rendered still
 * holds the required N (caller checks it at its next natural sync and retries). */
int b3gs_forward_capacity(const B3gsScene* scene, char* geometry, char* binning, int64_t binning_capacity,
                          char* image, float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                          int32_t* device_num_rendered, b3gs_stream_t stream);

/* Backward.  Pixel gradients: dL_dcolor [3,H,W] (required), dL_ddepth / dL_dalpha [1,H,W] or NULL.
 * Outputs (all fully overwritten, culled Gaussians get zeros):
 *   dL_dmeans2D [P,3]  (x,y in the NDC-scaled units the densifier thresholds, z = 0)
 *   dL_dcolors  [P,3]  gradient of the per-Gaussian RGB (= grad of colors_precomp)
 *   dL_dopacity [P,1], dL_dmeans3D [P,3], dL_dcov3D [P,6]
 *   dL_dsh [P,M,3] (NULL if colours were precomputed), dL_dscales [P,3], dL_drotations [P,4]
 *   (NULL if cov3D was precomputed)
 * num_rendered < 0 means "read N from the image buffer" (forward_capacity path). */
int b3gs_backward(const B3gsScene* scene, int32_t num_rendered, const int32_t* radii,
                  const char* geometry, const char* binning, const char* image,
                  const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                  float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D,
                  float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations,
                  b3gs_stream_t stream);

/* ---- fused-activation ("raw parameter") path ------------------------------------------------
 * The reference computes the rasterizer's inputs with five small PyTorch kernels per view
 * (scene/gaussian_model.py:95-115: exp, normalize, sigmoid, cat) and autograd adds five more in
 * the backward plus one gradient-accumulation add per parameter per view.  These two entry
 * points take the PRE-activation parameters, apply the activations inside the per-Gaussian
 * kernels and accumulate (+=) the gradients straight into parameter-shaped buffers (e.g. views
 * into one flat slab that is later all-reduced).  Same arithmetic, same outputs; not part of the
 * reference's extension API -- used by the build's own training step. */
typedef struct B3gsRawParams {
  const float* xyz;            /* [P,3]                                   (GaussianModel._xyz) */
  const float* features_dc;    /* [P,1,3]                                 (_features_dc) */
  const float* features_rest;  /* [P,M-1,3] (may be NULL when M == 1)     (_features_rest) */
  const float* scaling;        /* [P,3]  scales    = exp(scaling)         (_scaling) */
  const float* rotation;       /* [P,4]  rotations = normalize(rotation)  (_rotation) */
  const float* opacity;        /* [P,1]  opacity   = sigmoid(opacity)     (_opacity) */
} B3gsRawParams;
typedef struct B3gsRawGrads {  /* accumulated into (+=); same shapes as B3gsRawParams */
  float* xyz; float* features_dc; float* features_rest; float* scaling; float* rotation; float* opacity;
  /* Optional sparse-row mode of b3gs_backward_raw_accumulate[_range] with overwrite != 0 (NULL = dense; ignored by
   * every other entry point): a bitmap of ceil(P / 64) words, bit (i & 63) of word (i >> 6) = "Gaussian i received a
   * gradient from at least one view of the call".  Rows whose bit is clear are NOT stored (most Gaussians of an
   * iteration: ~80 % at the headline workload) -- their content is stale and only a consumer that reads the bitmap
   * (b3gs_adam_step with row_mask) may use the buffers.  Range calls need first % 64 == 0. */
  uint64_t* touched_rows;
} B3gsRawGrads;

/* Sync-free forward (see b3gs_forward_capacity) on raw parameters.  `view` supplies P, D, M (= total
 * SH coefficients per channel), W, H, tan_fov*, scale_modifier, prefiltered, debug, background,
 * viewmatrix, projmatrix, campos; its per-Gaussian tensor pointers are ignored. */
/* `phases`: bit 0 = per-Gaussian projection + binning (ends with the tile lists in `binning`), bit 1 =
 * blend forward (the three output images); 3 = both.  A caller rendering several views runs phase 1 of
 * each view on its own stream and then ONE b3gs_blend_forward_batch() for all of them. */
int b3gs_forward_raw(const B3gsScene* view, const B3gsRawParams* params, char* geometry, char* binning,
                     int64_t binning_capacity, char* image, float* out_color, float* out_depth, float* out_alpha,
                     int32_t* radii, int32_t* device_num_rendered, int phases, b3gs_stream_t stream);

/* The whole forward of up to 8 views of the SAME Gaussians, every stage one launch for all views:
 * the projection reads and activates each Gaussian once for all views, the radix passes / scans / emission
 * / blend take the views as blockIdx.y (a single view's pass is ~250 workgroups on 256 CUs: latency-bound).
 * depth_order_from: -1 = this view sorts its own depth keys; k = reuse view k's depth order -- valid iff
 * both view matrices have the same z row, which is what the binocular shifted cameras are (the translation is
 * along the camera x axis, utils/pose_utils.py:148-163): one depth sort per input/shifted pair.
 * Views share P, M, D, scale_modifier; W, H may differ.  phases as in b3gs_forward_raw. */
typedef struct B3gsForwardView {
  const B3gsScene* view;
  char* geometry;
  char* binning;
  int64_t binning_capacity;
  char* image;
  float* out_color;
  float* out_depth;
  float* out_alpha;
  int32_t* radii;
  int32_t* device_num_rendered;  /* N = tile instances binned (segment 1 + segment 2) */
  int32_t depth_order_from;
  /* Two-round ("termination-aware") binning.  0 < seg1_fraction < 1: only the nearest ceil(seg1_fraction * P) Gaussians
   * of the depth order -- rounded UP to a whole number of the scan's 4096-Gaussian tiles; when that reaches P (always below
   * 4097 Gaussians) the forward is binned in one round -- are binned first (segment 1 of every tile list); the blend
   * forward marks the tiles whose pixels
   * all terminated inside it (T < 1e-4: nothing behind can contribute), and the remaining Gaussians are binned into the
   * OTHER tiles only (segment 2), which are then blended again over segment 1 + segment 2.  Images, n_contrib-relative
   * gradients and the order inside every list are those of one-round binning; only the instances no pixel could have
   * reached are never emitted or sorted.  0 or >= 1: one round.  All views of a batch use views[0]'s value.
   * The `image` buffer carries a prediction from one two-round forward to the next one into the SAME buffer: the tiles
   * left unterminated are predicted open and receive their complete list in segment 1 (then nothing is left for segment
   * 2 on a settled scene).  Any buffer content is valid -- the prediction only moves work between the rounds, the
   * results do not depend on it -- but a caller that wants the saving keeps one image buffer per camera (and zeroes a
   * fresh one). */
  float seg1_fraction;
  /* ABI 6 -- overflow that cannot corrupt a step (both optional, device int32 words the caller keeps across forwards):
   * high_water    <- max(high_water, N)           by the binning kernels themselves (no extra launch);
   * overflow_flag <- 1 (sticky) when N > binning_capacity, i.e. this view was rendered from truncated tile lists.
   * b3gs_adam_step(skip_if_nonzero = overflow_flag) and B3gsDensifyStats::skip_if_nonzero then turn every step from
   * the overflowing one on into a no-op for the parameters, the Adam state and the statistics, until the host has read
   * the flag, grown the buffers and cleared it: the steps since the last check can really be repeated.
   * Bits of the word: 0 = capacity, 1 = a depth key outside the 27-bit span (depth_key_bits), 2 (ABI 7) = the second
   * binning round's persistent launch timed out at its grid barrier (its workgroups were not co-resident: a shared or
   * partitioned device) -- the repaired tiles of that forward are wrong, the step is dropped like an overflowing one and
   * the caller goes back to one round (seg1_fraction = 0); 3 (ABI 7) = a TRUSTED depth-order hint was wrong
   * (hint_trusted): the view was rendered from another view's depth order. */
  int32_t* high_water;
  int32_t* overflow_flag;
  /* 27: the caller vouches that every visible Gaussian's depth key (float bits of view z > 0.2) lies within 2^27 of the
   * bits of 0.2f, i.e. z < ~13107: the depth sort then runs three 9-bit passes instead of four 8-bit ones (same
   * permutation).  The library CHECKS it: a key outside the span raises bit 1 of *overflow_flag (required non-NULL for
   * this mode), which drops the step like a capacity overflow; the caller then falls back to 0.  0 (or 32): full sort. */
  int32_t depth_key_bits;
  /* != 0: `image` is a fresh allocation with undefined content (not the buffer of an earlier forward): nothing is read
   * from it -- neither the tile order the previous backward of a persistent buffer leaves for the next forward nor the
   * open-tile prediction (two-round binning is then off).  A caller that re-uses image buffers across forwards zeroes a
   * new one ONCE and passes 0.  All views of a batch use views[0]'s value. */
  int32_t fresh_image;
  /* ABI 7 -- depth order of an EARLIER forward, checked on the device.  depth_order_hint (may be NULL) is the geometry
   * buffer of a completed earlier b3gs_forward_raw_batch() of the same P Gaussians (same depth_key_bits) on this stream;
   * it must stay alive until this forward has run.  The projection compares every depth key of this view with the key that
   * forward stored (keys are a function of the Gaussian's position and the z row of the view matrix only: the binocular
   * partner of train.py:124-128 differs from its input view by a translation along the camera x axis, so all keys are
   * equal); any difference -- another camera, moved Gaussians, a different near-plane cull -- sets *hint_mismatch
   * (device int32, must be ZERO on entry, required when a hint is given) and this view's own depth sort runs as usual.
   * While the word stays zero the sort launches exit at once and the earlier order is adopted: same lists bit for bit,
   * ~60 us less per view at 1M Gaussians.  The answer never depends on the caller being right about the hint.
   * Only for views that sort their own keys (depth_order_from == -1). */
  const char* depth_order_hint;
  int32_t* hint_mismatch;
  /* != 0: the caller KNOWS the keys are equal (it built both view matrices and their z rows are the same bits): the depth
   * sort is not even launched (an idle launch is ~5 us on this part, nine of them per view).  Still checked: a key that
   * differs sets *hint_mismatch and raises bit 3 of *overflow_flag (required non-NULL then) -- that view was rendered from
   * a wrong depth order, the step is dropped like an overflowing one and the caller stops trusting its knowledge.
   * With depth_order_from == k - 1 (view k shares the order of the PREVIOUS view of the batch, no hint buffer involved): the
   * projection compares this view's depth keys with that view's and raises bit 3 of *overflow_flag (required non-NULL) on
   * a difference -- the same check for a pair rendered in one batch.  0: the pair is the caller's word (fused path). */
  int32_t hint_trusted;
  /* ABI 8 (may be NULL): [P] bytes <- radii > 0, render()'s `visibility_filter` (gaussian_renderer/__init__.py:99), written
   * by the projection next to the radius instead of by a compare kernel per render afterwards. */
  uint8_t* visible;
  /* ABI 10 -- which tiles a Gaussian is binned into.  0 (default): the tiles its alpha >= 1/255 footprint can reach
   * ("tight" binning: same images, shorter lists -- every list is an order-preserving subsequence of the reference's).
   * != 0: every tile of the reference's rectangle (radius = ceil(3 sigma)): point_list / ranges
   * are then the reference's bit for bit, as on the b3gs_forward() surface.  All views of a batch use views[0]'s value.
   * (Until ABI 9 this was a process-wide environment switch read inside the library.) */
  int32_t reference_binning;
} B3gsForwardView;
int b3gs_forward_raw_batch(int32_t nviews, const B3gsForwardView* views, const B3gsRawParams* params, int phases,
                           b3gs_stream_t stream);

/* Blend (per-tile alpha compositing) of up to 8 views in ONE launch each way.  One view's ~1900 tiles fill
 * the 256 CUs roughly once, so a per-view launch pays its own tail; batched, the dispatcher packs the tiles
 * of all views (MI355X, 1M Gaussians, 800x600: forward 126 us per view alone, 74 us per view batched).
 * forward: needs every view's b3gs_forward_raw(..., phases = 1) complete on a stream `stream` waits for;
 * backward: = phase 1 of b3gs_backward_raw for every view (own zeroed `scratch` each). */
typedef struct B3gsBlendView {
  const B3gsScene* view;
  const char* geometry;
  const char* binning;
  const char* image;
  float* out_color;          /* forward */
  float* out_depth;
  float* out_alpha;
  const float* dL_dcolor;    /* backward */
  const float* dL_ddepth;    /* may be NULL */
  const float* dL_dalpha;    /* may be NULL */
  float* scratch;            /* backward: b3gs_backward_scratch_floats(P) zeroed floats */
  int64_t binning_capacity;  /* the capacity `binning` was carved with in the forward (two-word instance layout: locates
                              * nothing the blend reads; kept for symmetry with B3gsForwardView) */
} B3gsBlendView;
int b3gs_blend_forward_batch(int32_t nviews, const B3gsBlendView* views, b3gs_stream_t stream);
int b3gs_blend_backward_batch(int32_t nviews, const B3gsBlendView* views, b3gs_stream_t stream);

/* Backward of b3gs_forward_raw.  `scratch` holds b3gs_backward_scratch_floats(P) floats that must be
 * ZERO on entry and are left zero on exit (persistent across views: no per-view memset).
 * dL_dmeans2D ([P,3], optional) is overwritten with the screen-space mean gradients.
 * `phases`: bit 0 = blend backward (pixel gradients -> per-Gaussian sums in `scratch`), bit 1 = per-Gaussian
 * chain rule + accumulation into `grads`; 3 = both.  Views rendered concurrently on several streams call
 * phase 1 in parallel (own scratch each) and order their phase-2 calls with stream events, because phase 2
 * read-modify-writes the shared gradient buffers without atomics. */
size_t b3gs_backward_scratch_floats(int32_t P);
int b3gs_backward_raw(const B3gsScene* view, const B3gsRawParams* params, const int32_t* radii, const char* geometry,
                      const char* binning, const char* image, const float* dL_dcolor, const float* dL_ddepth,
                      const float* dL_dalpha, float* scratch, const B3gsRawGrads* grads, float* dL_dmeans2D,
                      int phases, b3gs_stream_t stream);

/* Phase 2 for SEVERAL views in one pass over the Gaussians (at most 8 per call): parameters are read
 * once, each view contributes its phase-1 sums (read and reset), the gradients are written once --
 * `overwrite` != 0 stores them (culled-everywhere Gaussians get zeros: no zero-fill of `grads` needed),
 * 0 accumulates (+=).  Every view must have completed b3gs_backward_raw(..., phases = 1) with its OWN
 * scratch on a stream that `stream` has been made to wait for.  All views share P, M, D, scale_modifier. */
typedef struct B3gsFusedView {
  const B3gsScene* view;     /* camera fields as passed to b3gs_forward_raw */
  const int32_t* radii;
  const char* geometry;
  float* scratch;            /* that view's phase-1 sums; left zero */
  float* dL_dmeans2D;        /* optional [P,3] */
  int32_t densify_stats;     /* != 0: this view feeds the densification statistics (train.py:178-179) */
} B3gsFusedView;
/* Densification statistics of scene/gaussian_model.py:147-152,409-411 and train.py:178, updated for every
 * Gaussian visible (radii > 0) in a view with densify_stats set:
 *   xyz_gradient_accum += ||dL_dmeans2D[:2]||,  denom += 1,  max_radii2D = max(max_radii2D, radii). */
typedef struct B3gsDensifyStats {
  float* xyz_gradient_accum; /* [P,1] */
  float* denom;              /* [P,1] */
  float* max_radii2D;        /* [P]   */
  const int32_t* skip_if_nonzero;  /* ABI 6, may be NULL: device word; != 0 -> the statistics are left untouched (see
                                    * B3gsForwardView::overflow_flag) */
} B3gsDensifyStats;
int b3gs_backward_raw_accumulate(int32_t nviews, const B3gsFusedView* views, const B3gsRawParams* params,
                                 const B3gsRawGrads* grads, int32_t overwrite, const B3gsDensifyStats* stats,
                                 b3gs_stream_t stream);
/* The same for the Gaussians [first, first + count) only.  Every pointer (parameters, gradients, statistics,
 * per-view state) is still indexed by the GLOBAL Gaussian index, so a caller that keeps the gradients of a range
 * in one contiguous block passes block_base - first * row_width.  Disjoint ranges issued back to back let the
 * gradient all-reduce of one range overlap the chain rule of the next (step.ViewShardedStep, data parallel). */
int b3gs_backward_raw_accumulate_range(int32_t nviews, const B3gsFusedView* views, const B3gsRawParams* params,
                                       const B3gsRawGrads* grads, int32_t overwrite, const B3gsDensifyStats* stats,
                                       int32_t first, int32_t count, b3gs_stream_t stream);

/* ---- fused optimiser step -----------------------------------------------------------
 * Adam exactly as torch.optim.Adam (no amsgrad, no weight decay) for up to 8 parameter tensors with their
 * own learning rates in ONE launch: the reference's six parameter groups (scene/gaussian_model.py:154-167,
 * eps = 1e-15) and optimizer.step() at train.py:196-198.  `device_step` (int32 on the device, incremented
 * by the call unless bump_step_after == 0: a step issued as several calls over disjoint slices -- the pipelined
 * data-parallel tail -- bumps on the last one) keeps the bias correction replayable from a HIP graph.  opacity_decay > 0 additionally
 * applies  o <- logit(sigmoid(o) * opacity_decay)  to segment `opacity_segment` (gaussian_model.py:307-309,
 * train.py:171-173): after its update when opacity_decay_first == 0; when != 0, in the reference's order -- the
 * decay runs BEFORE optimizer.step() (train.py:171-173 vs :196-198), so the Adam update (computed from the gradient at
 * the un-decayed value) is subtracted from the decayed logit.  A segment with count == 0 may carry NULL pointers
 * (features_rest at SH degree 0). */
typedef struct B3gsAdamSegment {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t count;  /* floats */
  float lr;
  int32_t row_len;    /* floats per Gaussian row of this tensor; only read when row_mask != NULL (0 = segment not masked) */
  int32_t first_row;  /* Gaussian index of the segment's first float (a segment starts at a row boundary when masked) */
  const float* lr_dev; /* ABI 7, may be NULL: a device float that REPLACES `lr` (the position learning rate follows a
                        * schedule, train.py:83: a step replayed as a HIP graph reads it from device memory) */
} B3gsAdamSegment;
/* `row_mask` (may be NULL): the touched_rows bitmap of B3gsRawGrads.  Element e of a segment with row_len > 0 belongs to
 * Gaussian first_row + e / row_len; when that Gaussian's bit is clear the gradient is taken as 0 WITHOUT reading it
 * (moments and parameter still follow Adam: same result as a dense zero gradient, 4 of 28 bytes per float less). */
/* ABI 7: `device_step` points to B3GS_ADAM_STEP_WORDS int32 words the optimiser owns, ZERO apart from the first:
 * {step, completion counter, 64 first-level completion counters a cache line apart} -- the workgroup that finishes last
 * advances the step (no second launch); two levels because thousands of workgroups arriving at one address cost ~15 ns
 * each (85 us of the launch at 500k Gaussians).  Optimisers stepping concurrently on several streams own their words.  `skip_if_nonzero` (may be NULL): device word; when it is != 0 at launch time the call changes
 * NOTHING (parameters, moments, step counter): the update of a step rendered from truncated tile lists is dropped on the
 * device, without a host round trip (B3gsForwardView::overflow_flag). */
#define B3GS_ADAM_STEP_WORDS (2 + 64 * 32)
int b3gs_adam_step(int32_t nseg, const B3gsAdamSegment* segs, int32_t* device_step, float beta1, float beta2,
                   float eps, float opacity_decay, int32_t opacity_segment, int32_t opacity_decay_first,
                   int32_t bump_step_after, const uint64_t* row_mask, const int32_t* skip_if_nonzero,
                   b3gs_stream_t stream);

/* ---- fused loss block ----------------------------------------------------------------
 * Value and pixel gradients of the per-pair training loss of train.py:123-148 in 4 launches:
 *   total = (1-lambda_dssim) L1(image, gt) + lambda_dssim (1 - SSIM(image, gt))
 *         + L1(warp(shifted, disp) mask, gt mask) + lambda_smooth smooth(disp mask, gt)     [shifted_image != NULL]
 *         + mean(|alpha| alpha_weight)                                                      [alpha_weight  != NULL]
 * disp = focal_x (-trans_dist) / (depth + 1e-5); warp / mask = utils/graphics_utils.py:80-125; smooth =
 * utils/loss_utils.py:68-91; SSIM = utils/loss_utils.py:36-66.  alpha_weight is (1 - gt_alpha_mask) or the DTU
 * background mask (train.py:139-143).  All images are [C,H,W] fp32 device tensors.  The four gradient images are
 * OVERWRITTEN with d(total)/d(input) * grad_scale; parts (8 device floats) receives total, Ll1, ssim,
 * l1_masked, smooth, alpha_loss.  No host synchronisation: graph-capturable. */
typedef struct B3gsLossIO {
  int32_t W, H;
  const float* image;          /* [3,H,W] primary render */
  const float* depth;          /* [1,H,W] */
  const float* alpha;          /* [1,H,W] */
  const float* gt_image;       /* [3,H,W] */
  const float* shifted_image;  /* [3,H,W] or NULL */
  const float* alpha_weight;   /* [1,H,W] or NULL */
  float focal_x, trans_dist, lambda_dssim, lambda_smooth, grad_scale;
  float* dL_dimage;            /* [3,H,W] */
  float* dL_ddepth;            /* [1,H,W] */
  float* dL_dalpha;            /* [1,H,W] */
  float* dL_dshifted;          /* [3,H,W]; required iff shifted_image */
  float* parts;                /* [8] */
  /* (`workspace` below.)  ABI 7, last field of the struct: trans_dist_dev (may be NULL) -- a device float that REPLACES
   * `trans_dist` when given: the shift of the binocular partner is drawn anew every iteration (train.py:125-126), and an
   * iteration replayed as a HIP graph can only see it through device memory. */
  float* workspace;            /* b3gs_loss_workspace_floats(W, H) floats.  ABI 7: its first 512 floats (the partial-sum
                                * slots) must be ZERO before the first call with this workspace; every call leaves them zero
                                * (self-cleaning: no memset per call).  One workspace per pair in flight. */
  const float* trans_dist_dev;
} B3gsLossIO;
size_t b3gs_loss_workspace_floats(int32_t W, int32_t H);
int b3gs_binocular_loss(const B3gsLossIO* io, b3gs_stream_t stream);
/* The same for up to 8 pairs (ios[0..npairs)) with ONE launch per stage: the images of one pair are ~1900 tiles. */
int b3gs_binocular_loss_batch(int32_t npairs, const B3gsLossIO* ios, b3gs_stream_t stream);

/* ---- densify / clone / split / prune ------------------------------------------------------
 * Net effect of the reference's densify_and_prune (scene/gaussian_model.py:393-407 with :258-391) including the
 * Adam-state surgery, as: b3gs_densify_classify (per-Gaussian decisions) -> caller forms the three exclusive prefix
 * sums and their totals (the only host read-back) -> b3gs_densify_scatter (one pass writes every surviving /
 * new row of the six parameter tensors and of exp_avg / exp_avg_sq).
 * Tensor order everywhere: xyz[P,3], features_dc[P,3], features_rest[P,3(M-1)], scaling[P,3], rotation[P,4],
 * opacity[P] (raw, pre-activation).  flags[i]: bit 0 = original kept, bit 1 = clone appended, bit 2 = two split
 * children appended.  Output rows: kept originals | clones | children k=0 | children k=1, each in index order
 * (the reference's order); new rows get zero Adam state.  noise: [2,P,3] standard-normal samples addressed by
 * the ORIGINAL index (child k of Gaussian i uses noise[k][i]): the caller owns the random stream, so replicas
 * that share it stay identical.  max_screen_size <= 0: opacity rule only (what train.py passes). */
typedef struct B3gsDensifyIO {
  int32_t P, M;
  const float* param[6];
  const float* exp_avg[6];        /* all six, or all NULL */
  const float* exp_avg_sq[6];
  const float* xyz_gradient_accum; /* [P] */
  const float* denom;              /* [P] */
  float grad_threshold, min_opacity, extent, percent_dense, max_screen_size;
} B3gsDensifyIO;
int b3gs_densify_classify(const B3gsDensifyIO* io, int32_t* flags, b3gs_stream_t stream);
int b3gs_densify_scatter(const B3gsDensifyIO* io, const int32_t* flags, const int32_t* off_keep,
                         const int32_t* off_clone, const int32_t* off_split, int32_t n_keep, int32_t n_clone,
                         int32_t n_split, const float* noise, float* const* out_param, float* const* out_exp_avg,
                         float* const* out_exp_avg_sq, b3gs_stream_t stream);

/* ---- ABI 9: the statements of the training loop ONE BY ONE, behind the reference's own call signatures ------------
 * An unchanged train.py:123-198 calls l1_loss / ssim / SmoothLoss.forward / inverse_warp_images as separate statements
 * with PyTorch glue between them, then gaussians.opacity_decay(), add_densification_stats() and optimizer.step(): each
 * of them is one launch here (value forward; EVERY input gradient backward), wrapped on the python side as
 * autograd.Functions / methods with exactly the reference's signatures (binocular3dgs_amd/loss_utils.py,
 * graphics_utils.py, gaussian_model.py, optim.py).  All tensors fp32, contiguous, on the device.
 *
 * Scalar results are deterministic (per-workgroup partial sums folded in index order by the workgroup that arrives
 * last: no float atomics, no second launch).  `workspace`: b3gs_lossfn_workspace_floats(planes, H, W) floats whose
 * header (the first 2080 words: two levels of arrival counters) is ZERO before the first call; every call leaves it zero.
 * One workspace per stream. */
size_t b3gs_lossfn_workspace_floats(int64_t planes, int32_t H, int32_t W);

/* utils/loss_utils.py:18-21  l1_loss(network_output, gt, mask=None) = mean |x*mask - y*mask|.
 * x, y: [batch, channels, hw]; mask: NULL, or [batch, hw] (broadcast over the channels; a mask of x's own shape is
 * passed as channels = 1, hw = numel / batch).  out: [1].  backward: grad_out = the upstream gradient, a device scalar;
 * grad_x / grad_y / grad_mask may each be NULL (not wanted). */
int b3gs_l1_loss_forward(const float* x, const float* y, const float* mask, int64_t batch, int32_t channels, int64_t hw,
                         float* out, float* workspace, b3gs_stream_t stream);
int b3gs_l1_loss_backward(const float* x, const float* y, const float* mask, int64_t batch, int32_t channels, int64_t hw,
                          const float* grad_out, float* grad_x, float* grad_y, float* grad_mask, b3gs_stream_t stream);

/* utils/graphics_utils.py:80-125  inverse_warp_images(image [B,C,H,W], disparity [B,1,H,W], row_indices, column_indices)
 * -> [B,C,H,W]: linear interpolation between columns c + floor(d) and the next one, zero where either leaves the row.
 * zero_fill (may be NULL): a [B,C,H,W] buffer the forward fills with zeros on its way -- the grad_image of the backward,
 * which ADDS into it (bilinear scatter); grad_image / grad_disparity may each be NULL. */
int b3gs_inverse_warp_forward(const float* image, const float* disparity, int32_t B, int32_t C, int32_t H, int32_t W,
                              float* out, float* zero_fill, b3gs_stream_t stream);
int b3gs_inverse_warp_backward(const float* image, const float* disparity, const float* grad_out, int32_t B, int32_t C,
                               int32_t H, int32_t W, float* grad_image, float* grad_disparity, b3gs_stream_t stream);

/* utils/loss_utils.py:68-91  SmoothLoss.forward(disparity [B,1,H,W], image [B,C,H,W]): edge-aware first-order
 * smoothness on the interior (the reference's fixed 3x3 convolutions have no padding).  out: [1]. */
int b3gs_smooth_loss_forward(const float* disparity, const float* image, int32_t B, int32_t C, int32_t H, int32_t W,
                             float* out, float* workspace, b3gs_stream_t stream);
int b3gs_smooth_loss_backward(const float* disparity, const float* image, int32_t B, int32_t C, int32_t H, int32_t W,
                              const float* grad_out, float* grad_disparity, float* grad_image, b3gs_stream_t stream);

/* utils/loss_utils.py:36-66  ssim(img1, img2, window_size=11, size_average=True): 11x11 Gaussian window (sigma 1.5),
 * zero padding, every [H,W] plane by itself.  out: [1], or [batch] when size_average == 0.  maps (NULL: value only):
 * 5 * batch * channels * H * W floats the backward reads (the last two fifths only written when maps_for_img2 != 0,
 * i.e. when img2 wants a gradient too).  backward: grad_out [1] or [batch]; grad_img1 / grad_img2 may be NULL. */
int b3gs_ssim_forward(const float* img1, const float* img2, int32_t batch, int32_t channels, int32_t H, int32_t W,
                      int32_t size_average, float* maps, int32_t maps_for_img2, float* out, float* workspace,
                      b3gs_stream_t stream);
int b3gs_ssim_backward(const float* img1, const float* img2, const float* maps, int32_t batch, int32_t channels, int32_t H,
                       int32_t W, int32_t size_average, const float* grad_out, float* grad_img1, float* grad_img2,
                       b3gs_stream_t stream);

/* scene/gaussian_model.py:307-309  opacity_decay(factor): o <- inverse_sigmoid(sigmoid(o) * factor), in place. */
int b3gs_opacity_decay(float* opacity, int64_t count, float factor, b3gs_stream_t stream);
/* scene/gaussian_model.py:409-411  add_densification_stats: for the rows update_filter (one byte per row) selects,
 * xyz_gradient_accum += ||viewspace_grad[row, :2]|| and denom += 1.  row_stride: floats per row of viewspace_grad (3). */
int b3gs_add_densification_stats(int64_t P, const float* viewspace_grad, int64_t row_stride, const uint8_t* update_filter,
                                 float* xyz_gradient_accum, float* denom, b3gs_stream_t stream);
/* Data-parallel tail: statistics of a step staged by the chain rule (B3gsDensifyStats pointing at three zeroed [P] arrays)
 * are added to the model's once the ranks agree that nobody overflowed -- `agreed_word` (device, any 32-bit pattern: the
 * float SUM of the ranks' flags that travelled inside the first gradient range's all-reduce) == 0 -- and discarded
 * otherwise, in which case *local_overflow_flag (may be NULL) gets bit 0 set.  The staging arrays are left zero. */
int b3gs_apply_staged_densify_stats(int64_t P, float* staged_accum, float* staged_denom, float* staged_max_radii,
                                    float* xyz_gradient_accum, float* denom, float* max_radii2D, const int32_t* agreed_word,
                                    int32_t* local_overflow_flag, b3gs_stream_t stream);
/* train.py:196-198 optimizer.step() for an optimiser that keeps torch.optim.Adam's state layout: b3gs_adam_step with the
 * step number given by the host (1-based, the value of state["step"] after its increment); no decay, no row mask. */
int b3gs_adam_step_at(int32_t nseg, const B3gsAdamSegment* segs, int32_t step, float beta1, float beta2, float eps,
                      b3gs_stream_t stream);

/* ---- scale initialisation -------------------------------------------------------------
 * mean_dist2[i] = mean squared distance from point i to its 3 nearest OTHER points: the distCUDA2 of the
 * reference's simple-knn extension (scene/gaussian_model.py:134: scales = log(sqrt(max(dist2, 1e-7)))).
 * Exact; points [P,3] and mean_dist2 [P] on the device; workspace = b3gs_knn_workspace_bytes(P) bytes. */
size_t b3gs_knn_workspace_bytes(int32_t P);
int b3gs_knn_mean_dist2(int32_t P, const float* points, float* mean_dist2, char* workspace, b3gs_stream_t stream);

/* Frustum test only: present[i] = 1 if Gaussian i passes the near-plane cull (view z > 0.2). */
int b3gs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present, b3gs_stream_t stream);

/* Read-only views into the opaque buffers, for parity tests (tile/bin indices must be
 * bit-exact with the oracle).  Pointers are device pointers inside the caller's buffers. */
typedef struct B3gsDebugViews {
  const uint32_t* tiles_touched; /* [P] */
  const float* depths;           /* [P] */
  const float* records;          /* [P,16] x,y,cxx,cxy,cyy,opacity,r,g,b,depth,ext_x,ext_y, 4 pad (zero: written so that a slot is one unmasked 64-byte store) */
  const uint32_t* point_list;    /* [N] one word per instance, tile-major, depth-sorted: the Gaussian index, or
                                  *     (tile << packed_idx_bits) | index when packed_idx_bits >= 0 */
  const uint32_t* tile_ids;      /* [N] tile id per sorted instance; NULL when the words are packed */
  const uint32_t* ranges;        /* [tiles,2] begin, end; an EMPTY tile holds (0xFFFFFFFF, 0) */
  const float* final_T;          /* [H*W] */
  const uint32_t* n_contrib;     /* [H*W] */
  int32_t packed_idx_bits;       /* >= 0 whenever bits(P) + bits(tiles) <= 32 (e.g. <= 2M Gaussians at 800x600) */
  const uint32_t* counts;        /* [3] N1 (segment 1 / the only segment), V, N2 (segment 2 of a two-round forward, else stale) */
  const uint32_t* point_list2;   /* segment 2 of the tile lists: N2 entries starting at element counts[0] (= N1) of this array (it is
                                  * point_list: segment 2 sits behind segment 1); tile ids likewise at tile_ids + N1 */
  const uint32_t* ranges2;       /* [tiles,2] segment 2 ranges; EMPTY = (0xFFFFFFFF, 0) */
} B3gsDebugViews;
int b3gs_debug_views(int32_t P, int32_t W, int32_t H, int64_t num_rendered, const char* geometry,
                     const char* binning, const char* image, B3gsDebugViews* out);

/* Parity hook (ABI 9): the accessors of scene/gaussian_model.py:95-115 exactly as the raw-parameter kernels evaluate them
 * -- exp(_scaling) [P,3], F.normalize(_rotation) [P,4], sigmoid(_opacity) [P] -- so that a test can hold them against the
 * torch operators bit for bit (integer radii on the raw-parameter path depend on it). */
int b3gs_debug_activations(int32_t P, const B3gsRawParams* raw, float* scales, float* rotations, float* opacity,
                           b3gs_stream_t stream);

/* Per-stage timing hook (thread-local): when non-NULL, b3gs_forward/backward record HIP events
 * around every stage on `stream` WITHOUT synchronising.  After the caller has synchronised the
 * stream, b3gs_timing_collect() adds the elapsed milliseconds of every parked stage to the sink
 * and returns the number of stages resolved (at most 4096 may be parked).  Used by bench.py for
 * the roofline figure; never set in production. */
typedef struct B3gsKernelTimes {
  double preprocess_ms, sort_ms, render_fwd_ms, render_bwd_ms, preprocess_bwd_ms;
  int64_t calls;
} B3gsKernelTimes;
void b3gs_set_timing(B3gsKernelTimes* sink);
int b3gs_timing_collect(void);

const char* b3gs_last_error(void); /* thread-local message of the last failure */
int b3gs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* B3GS_RASTER_H */
