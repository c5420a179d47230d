// Internal declarations shared by the HIP translation units of libb3gs_raster.so.
// gfx950 / wave64 only: there is deliberately no other code path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/b3gs_raster.h"

#define B3GS_WAVE 64
#define B3GS_NEAR 0.2f
#define B3GS_ALPHA_MIN (1.0f / 255.0f)
#define B3GS_ALPHA_MAX 0.99f
#define B3GS_T_EPS 0.0001f
#define B3GS_REC_FLOATS 16 /* one 64-byte line per Gaussian */

// ---- layout of the three opaque buffers --------------------------------------------------
// Every sub-array starts on a 256-byte boundary.  All sizes are functions of (P), (W,H), (P,N)
// only, so forward and backward carve identical views out of the caller's bytes.

#define B3GS_MAX_FUSED_VIEWS_ 8   /* == B3GS_MAX_FUSED_VIEWS (defined below the layout helpers) */
#define B3GS_SCRATCH_ROW 10   /* floats per Gaussian of the blend backward's per-Gaussian sums (one 40-byte row) */

struct GeomView {       // sized by P
  uint32_t* header;     // [64]  header[0] = N (tile instances), header[1] = V (visible)
  float4* rec;          // [P*4] render record: x,y,cxx,cxy | cyy,op,r,g | b,depth,ext_x,ext_y | SH clamp bits (bit c: channel c clamped at 0), spare x 3
  uint32_t* depth_key;  // [P]   float bits of view z, 0xFFFFFFFF when culled
  uint32_t* tiles_touched;  // [P]
  uint2* rect;          // [2P]  packed u16 (x0 | y0<<16, x1 | y1<<16); [P] used unless a partner view interleaves its own
  uint32_t* skey[2];    // [P]   depth-sort ping/pong keys
  uint32_t* sval[2];    // [P]   depth-sort ping/pong values (Gaussian index)
  uint32_t* soffs;      // [P]   segment 1: sums of the 256-Gaussian sub-blocks of the depth order ([ceil(K1/256)] words);
                        //       segment 2: inclusive scan of scount over [K1,P) (written after segment 1 was emitted)
  uint2* srect;         // [P]   rect in depth order
  uint32_t* scount;     // [P]   segment 2: tiles of the rect that were not finished after segment 1, in depth order
  uint32_t* hist;       // radix histogram scratch, 256 * nblk(P)
  uint32_t* scan_tmp;   // [2048] chunk offsets of the scan | [2048][4] partial sums | [2048][4] partial visible counts
  float* bwd_rows;      // [P * B3GS_SCRATCH_ROW] drop-in backward: per-Gaussian sums of the blend backward (one row each)
  unsigned long long* pflag;  // [ceil(P / 64)] two-round forward: bit i set = the tile rect of Gaussian i reaches a tile that
                        //       is predicted open (written by the projection: the scan gathers the rects of the Gaussians
                        //       behind segment 1 only where this bit is set)
  uint32_t* flist;      // [P]   two-round forward: per 4096-Gaussian tile of the depth order behind segment 1, the flagged
                        //       Gaussians in depth order (local index | view flags); their number in fcount
  uint32_t* fcount;     // [ceil(P / 4096)]
  uint32_t* tsum;       // [ceil(P / 4096)] tile instances of every 4096-Gaussian tile of the depth order
  uint8_t* staged;      // [P]   staged[i] == the forward's epoch (header[B3GS_GEOM_EPOCH]): Gaussian i sits in some tile's list
                        //       below that tile's deepest used position, i.e. the blend backward MAY flush into its scratch
                        //       row; any other value (an older forward's epoch, garbage of a fresh buffer): it cannot.  The
                        //       chain rule's scan reads the 40-byte rows of the marked Gaussians only (~5 % of the visible ones).
};
// word of GeomView::header: the forward's epoch, 1..255, advanced by the first scan of every forward (binning.hip); a stale
// mark that happens to carry the current value again (255 forwards later, or garbage) only costs the scan a row read
#define B3GS_GEOM_EPOCH 8

struct BinView {        // sized by N (and P for the histogram)
  uint32_t* key[2];     // [N] tile id ping/pong
  uint32_t* val[2];     // [N] Gaussian index ping/pong
  uint32_t* hist;       // 256 * nblk(N)
};

// words of ImgView::header beyond the counts (0: N1, 1: V, 2: N2, 3: open tiles)
#define B3GS_HDR_REPAIR_BARRIER 8   /* [8] arrival counter of the repair kernel's grid barrier, [9] its status (bit 0: time-out),
                                     * [10] two-round forwards that needed the repair round, [11] two-round forwards (view 0's buffer) */

struct ImgView {        // sized by W*H
  uint32_t* header;     // [64]  header[0] = N used by the forward that filled this buffer
  float* final_T;       // [H*W]
  uint32_t* n_contrib;  // [H*W]
  uint2* ranges;        // [tiles]  segment 1 of every tile list
  uint2* ranges2;       // [tiles]  segment 2 (two-round binning, see BinJob::K1); empty = (0xFFFFFFFF, 0)
  uint32_t* tile_work;  // [tiles]  deepest list position any pixel of the tile used (forward) = the tile's backward work
  uint32_t* order;      // [8 * (tiles + 8)]  scheduling order of a batched blend-backward launch (view 0's array is used)
  unsigned long long* open_rows;  // [grid_y * ceil(grid_x / 64)] bit x % 64 of word (y, x / 64) set: tile (x, y) still has
                        //          an unterminated pixel after segment 1 AND was not predicted open (header[3] = their
                        //          number): the tiles the second binning round repairs; zeroed per forward
  unsigned long long* pred_rows;  // same shape: tiles PREDICTED open for this forward (= the tiles the previous two-round
                        //          forward into this buffer left unterminated): segment 1 gives them their complete list
  unsigned long long* pred_next;  // same shape: written by this forward, becomes pred_rows of the next one
};

#ifndef B3GS_SORT_ITEMS
#define B3GS_SORT_ITEMS 16                       /* keys per thread in a radix tile */
#endif
#define B3GS_SORT_THREADS 256
#define B3GS_SORT_TILE (B3GS_SORT_ITEMS * B3GS_SORT_THREADS) /* 4096 keys per workgroup */

static inline size_t b3gs_align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline uint32_t b3gs_sort_blocks(int64_t n) { return (uint32_t)((n + B3GS_SORT_TILE - 1) / B3GS_SORT_TILE); }

// bits of a tile id at W x H
static inline int b3gs_tile_bits(int W, int H) {
  const size_t tiles = (size_t)((W + B3GS_TILE - 1) / B3GS_TILE) * (size_t)((H + B3GS_TILE - 1) / B3GS_TILE);
  int tbits = 0;
  while (((size_t)1 << tbits) < tiles) tbits++;
  return tbits;
}
// Tile instances are ONE 32-bit word (tile << idx_bits) | gaussian_index whenever both fit (e.g. up to 2M
// Gaussians at 800x600); returns idx_bits, or -1 when they do not and (tile, index) are two words.
// A pure function of (P, W, H): forward, backward and debug views all derive the same answer.
static inline int b3gs_packed_idx_bits(int32_t P, int W, int H) {
  int pbits = 1;
  while (((int64_t)1 << pbits) < (int64_t)P) pbits++;
  return pbits + b3gs_tile_bits(W, H) <= 32 ? pbits : -1;
}

// radix-sort scratch: 1280 header words (global digit histograms, tickets) + one status word per
// (pass <= 4, workgroup, digit) for the chained scan; also covers the 3-launch variant's 256*(nblk+1)
// (+ 4096: the histogram rows are padded to a multiple of 16 columns, binning.hip::hist_stride)
// (the 9-bit depth sort needs 512 rows of hist_stride(nblk) <= nblk + 15 words + 512 totals: 512 nblk + 8192, inside this)
static inline size_t b3gs_sort_scratch_words(int64_t n) { return 1280 + 8192 + (size_t)4 * 256 * (b3gs_sort_blocks(n) + 1); }

// carve: if base == nullptr only the size is computed
template <typename T>
static inline T* b3gs_carve(char*& cur, size_t count) {
  T* p = reinterpret_cast<T*>(cur);
  cur += b3gs_align256(count * sizeof(T));
  return p;
}

static inline size_t b3gs_geom_view(char* base, int32_t P, GeomView* v) {
  char* cur = base;
  size_t p = (size_t)(P > 0 ? P : 1);
  GeomView t;
  t.header = b3gs_carve<uint32_t>(cur, 64);
  t.rec = b3gs_carve<float4>(cur, p * 4);
  t.depth_key = b3gs_carve<uint32_t>(cur, p);
  t.tiles_touched = b3gs_carve<uint32_t>(cur, p);
  t.rect = b3gs_carve<uint2>(cur, 2 * p);  // second half: the binocular partner's rects, interleaved (stride 2)
  for (int i = 0; i < 2; i++) t.skey[i] = b3gs_carve<uint32_t>(cur, p);
  for (int i = 0; i < 2; i++) t.sval[i] = b3gs_carve<uint32_t>(cur, p);
  t.soffs = b3gs_carve<uint32_t>(cur, p);
  t.srect = b3gs_carve<uint2>(cur, p);
  t.scount = b3gs_carve<uint32_t>(cur, p);
  t.hist = b3gs_carve<uint32_t>(cur, b3gs_sort_scratch_words((int64_t)p));
  t.scan_tmp = b3gs_carve<uint32_t>(cur, 2048 + 2 * 4 * 2048);
  t.bwd_rows = b3gs_carve<float>(cur, p * B3GS_SCRATCH_ROW);
  t.pflag = b3gs_carve<unsigned long long>(cur, (p + 63) / 64);
  t.flist = b3gs_carve<uint32_t>(cur, (p + 4095) / 4096 * 4096);
  t.fcount = b3gs_carve<uint32_t>(cur, (p + 4095) / 4096);
  t.tsum = b3gs_carve<uint32_t>(cur, (p + 4095) / 4096);
  t.staged = b3gs_carve<uint8_t>(cur, p);
  if (v) *v = t;
  return (size_t)(cur - base);
}

static inline size_t b3gs_bin_view(char* base, int32_t P, int64_t N, BinView* v) {
  char* cur = base;
  size_t n = (size_t)(N > 0 ? N : 1);
  BinView t;
  // val[0] (the final point_list) sits at offset 0 so that consumers which do not know the
  // capacity the buffer was carved with (backward of the sync-free forward) still find it
  t.val[0] = b3gs_carve<uint32_t>(cur, n);
  t.key[0] = b3gs_carve<uint32_t>(cur, n);
  t.val[1] = b3gs_carve<uint32_t>(cur, n);
  t.key[1] = b3gs_carve<uint32_t>(cur, n);
  t.hist = b3gs_carve<uint32_t>(cur, b3gs_sort_scratch_words((int64_t)n));
  (void)P;
  if (v) *v = t;
  return (size_t)(cur - base);
}

static inline size_t b3gs_img_view(char* base, int32_t W, int32_t H, ImgView* v) {
  char* cur = base;
  size_t hw = (size_t)W * (size_t)H;
  size_t tiles = (size_t)((W + B3GS_TILE - 1) / B3GS_TILE) * (size_t)((H + B3GS_TILE - 1) / B3GS_TILE);
  ImgView t;
  t.header = b3gs_carve<uint32_t>(cur, 64);
  t.final_T = b3gs_carve<float>(cur, hw ? hw : 1);
  t.n_contrib = b3gs_carve<uint32_t>(cur, hw ? hw : 1);
  t.ranges = b3gs_carve<uint2>(cur, tiles ? tiles : 1);
  t.ranges2 = b3gs_carve<uint2>(cur, tiles ? tiles : 1);
  t.tile_work = b3gs_carve<uint32_t>(cur, tiles ? tiles : 1);
  t.order = b3gs_carve<uint32_t>(cur, B3GS_MAX_FUSED_VIEWS_ * (tiles + 8));
  const size_t row_words = (size_t)((H + B3GS_TILE - 1) / B3GS_TILE + 1) * (size_t)(((W + B3GS_TILE - 1) / B3GS_TILE + 63) / 64);
  t.open_rows = b3gs_carve<unsigned long long>(cur, row_words);
  t.pred_rows = b3gs_carve<unsigned long long>(cur, row_words);
  t.pred_next = b3gs_carve<unsigned long long>(cur, row_words);
  if (v) *v = t;
  return (size_t)(cur - base);
}

// scene + optional raw (pre-activation) parameters; raw_mode selects the fused-activation kernels
struct SceneX {
  B3gsScene sc;
  B3gsRawParams raw;
  int raw_mode;
  int tight;  // bin only the tiles the alpha >= 1/255 footprint can reach (fused path; see preprocess.hip)
};

#define B3GS_MAX_FUSED_VIEWS 8

// A set of tiles as a bitmap: word (y, x / 64), bit x % 64 (ImgView::open_rows / pred_rows).
struct OpenMap {
  const unsigned long long* rows;
  uint32_t row_words, grid_x, grid_y;
};
__host__ __device__ inline OpenMap open_map(const unsigned long long* rows, int W, int H) {
  const int gx = (W + B3GS_TILE - 1) / B3GS_TILE, gy = (H + B3GS_TILE - 1) / B3GS_TILE;
  return OpenMap{rows, (uint32_t)((gx + 63) / 64), (uint32_t)gx, (uint32_t)gy};
}
#ifdef __HIPCC__
// bits of row y inside columns [x0, x1) of 64-column block wb, shifted so that bit 0 is column max(x0, 64 wb)
__device__ __forceinline__ unsigned long long open_bits(const OpenMap& om, uint32_t y, uint32_t wb, uint32_t x0, uint32_t x1,
                                                        uint32_t* col0) {
  const uint32_t lo = max(x0, wb * 64u), hi = min(x1, wb * 64u + 64u);
  const unsigned long long m = om.rows[(size_t)y * om.row_words + wb] >> (lo & 63u);
  *col0 = lo;
  const uint32_t w = hi - lo;
  return w >= 64u ? m : (m & ((1ull << w) - 1ull));
}
// number of open tiles inside the rect (packed u16: x0 | y0 << 16, x1 | y1 << 16)
__device__ __forceinline__ uint32_t open_tiles(uint2 rc, const OpenMap& om) {
  const uint32_t x0 = rc.x & 0xFFFFu, y0 = rc.x >> 16, x1 = rc.y & 0xFFFFu, y1 = rc.y >> 16;
  if (x1 <= x0) return 0u;
  uint32_t n = 0, c0;
  for (uint32_t y = y0; y < y1; y++)
    for (uint32_t wb = x0 >> 6; wb <= (x1 - 1u) >> 6; wb++) n += (uint32_t)__builtin_popcountll(open_bits(om, y, wb, x0, x1, &c0));
  return n;
}
#endif

// ---- error reporting shared by every translation unit (api.hip owns the thread-local message) -------
// b3gs_fail: store the message b3gs_last_error() returns and hand back `code`;
// b3gs_launch_status: B3GS_OK, or B3GS_ERR_HIP with "<what>: <hip error string>" when a launch / runtime call failed
int b3gs_fail(int code, const char* what, const char* detail);
int b3gs_launch_status(const char* what);

// ---- launchers implemented in the individual .hip files -----------------------------------
// (all enqueue on `s`, none synchronise)

// Projection of every Gaussian into up to B3GS_MAX_FUSED_VIEWS views in ONE pass over the parameters
// (thread = Gaussian, loop over views: position / covariance / opacity are read and activated once).
// In raw mode all views share `raw`; otherwise nviews must be 1 and the tensors come from sc[0].
// Also zeroes every view's tile ranges.
struct PreOut {
  float4* rec;
  uint32_t* depth_key;
  uint32_t* tiles_touched;
  uint2* rect;          // element i at rect[i * rect_stride]
  int32_t rect_stride;  // 1, or 2 when a binocular pair shares one [P][2] array (one 16-byte gather serves both views)
  int32_t rect_role;    // 0: store rect[i * rect_stride]; 1: first of a pair, the NEXT view of the batch is its partner
                        //    (keep the rect in a register); 2: second of that pair: store both as one 16-byte word
  int32_t* radii;
  uint8_t* visible = nullptr;   // optional: radii > 0 as bytes (B3gsForwardView::visible)
  uint2* ranges;
  uint2* ranges2;
  unsigned long long* open_rows;
  int32_t* span_flag;              // non-null: raise bit 1 of this word when a visible depth key lies outside the 27-bit span
  const unsigned long long* pred_rows;   // two-round forward only (else null): the tiles predicted open for THIS forward
                                   //   (rotated from pred_next by the previous forward's last kernel, render.hip)
  unsigned long long* pflag;       //   ... and where the per-Gaussian "reaches a predicted tile" bits go (GeomView::pflag)
  int32_t ntiles, nrowwords;
  // ABI 7 (B3gsForwardView::depth_order_hint): the depth keys an earlier forward stored; a key of this view that differs
  // sets *hint_word (then this view's own depth sort runs instead of adopting that forward's order)
  const uint32_t* hint_key = nullptr;
  int32_t* hint_word = nullptr;
  int32_t* hint_fatal = nullptr;   // trusted hint (no sort launched): a differing key also raises bit 3 of this overflow word
  int32_t* pair_fatal = nullptr;   // rect_role 2 (second of an adjacent pair that shares its partner's depth order): a key that
                                   // differs from the partner's raises bit 3 of this overflow word (B3gsForwardView::hint_trusted)
};
struct PreBatch {
  int32_t n, raw_mode, tight;
  B3gsRawParams raw;
  B3gsScene sc[B3GS_MAX_FUSED_VIEWS];
  PreOut out[B3GS_MAX_FUSED_VIEWS];
};
PreOut b3gs_pre_out(const B3gsScene& sc, const GeomView& g, const ImgView& im, int32_t* radii);
void b3gs_launch_preprocess(const PreBatch& pb, hipStream_t s);
void b3gs_launch_preprocess_backward(const SceneX& sx, const GeomView& g, const int32_t* radii,
                                     float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
                                     float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                                     float* dL_drotations, const B3gsRawGrads* rg, float* m2d_out, hipStream_t s);
void b3gs_launch_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, uint8_t* present,
                              hipStream_t s);

// Binning of one or several views of the same P Gaussians (every kernel takes all views: blockIdx.y):
// depth sort of all P Gaussians (those behind the near plane sink to the end), scan of tiles_touched in
// depth order (N -> g.header[0], V -> g.header[1], N also -> im.header[0] and *n_out when non-null), emission of
// the (tile, idx) instances in depth order, stable sort by tile id, tile ranges.  Final lists end up in
// b.key[0] / b.val[0].  `n_bound` = number of instances the launches must cover (host-known N, or the
// capacity when N lives only on the device; kernels clamp to the device-side N).
// order_from: -1 = sort this view's own depth keys; k >= 0 = reuse the depth order of view k of the batch
// (whose order_from must be -1), valid when both views assign the same view-space z to every Gaussian.
struct BinJob {
  int32_t W, H;
  GeomView g;
  BinView b;
  ImgView im;
  int64_t n_bound;
  int32_t* n_out;
  int32_t order_from;
  const uint32_t* order;  // internal (order_from == -2): explicit depth order
  const uint2* rect;      // tile rects as written by the projection (PreOut::rect / rect_stride); null: g.rect, stride 1
  int32_t rect_stride;
  // Two-round ("termination-aware") binning, fused path: K1 in (0, P) bins only the nearest K1 Gaussians of the depth
  // order first (segment 1); after the blend forward has marked the tiles whose pixels all terminated, the remaining
  // Gaussians are binned into the OTHER tiles only (segment 2, b3gs_launch_round2_batch).  0 or >= P: one round.
  // Tiles PREDICTED open (im.pred_rows: the ones the previous forward into this image buffer left unterminated) also
  // receive the Gaussians behind K1 in segment 1 -- their complete list in one round; segment 2 then only repairs the
  // tiles the prediction missed (none once it has settled).  Any bitmap gives the same images: it only moves work.
  int32_t K1;
  // ABI 6 (optional): high_water <- max(high_water, N); overflow_flag <- 1 when N > n_bound (B3gsForwardView)
  int32_t* high_water = nullptr;
  int32_t* overflow_flag = nullptr;
  // 27: every visible depth key lies within 2^27 of the float bits of B3GS_NEAR (checked by the projection, which raises
  // bit 1 of overflow_flag otherwise): three 9-bit passes instead of four 8-bit ones.  0 / 32: the full 32-bit sort.
  int32_t key_bits = 0;
  // ABI 7: depth order (sval[0]) and sorted keys (skey[0]) of an earlier forward, adopted while *hint_word == 0; the
  // sort launches of this view exit at once then (order_from must be -1)
  const uint32_t* hint_sval = nullptr;
  const uint32_t* hint_skey = nullptr;
  const int32_t* hint_word = nullptr;
  int32_t hint_trusted = 0;        // != 0: the sort is not launched at all (the projection flags a differing key as fatal)
};
void b3gs_launch_binning_batch(int32_t P, int nviews, const BinJob* jobs, hipStream_t s);
void b3gs_launch_sort_u32_index(const uint32_t* keys, uint32_t* const skey[2], uint32_t* const sval[2], uint32_t n,
                                uint32_t* hist, hipStream_t s);   // hist: b3gs_sort_scratch_words(n) words
// the two halves, for callers that size the binning buffer from N in between (b3gs_forward)
void b3gs_launch_depth_order_batch(int32_t P, int nviews, const BinJob* jobs, hipStream_t s);  // sort + scan
void b3gs_launch_tile_lists_batch(int32_t P, int nviews, const BinJob* jobs, hipStream_t s);   // emit + split + ranges
// segment 2 of every view's tile lists (jobs with 0 < K1 < P; packed instance words only): count / scan / emit the
// Gaussians [K1, P) of the depth order into the tiles still open in im.open_rows, stable split by tile id into
// b.key[0] (the idle half of the packed-word binning buffer), ranges -> im.ranges2, N2 -> header[2], *n_out += N2
void b3gs_launch_round2_batch(int32_t P, int nviews, const BinJob* jobs, hipStream_t s);
static inline int32_t b3gs_seg1_count(const BinJob& j, int32_t P) { return (j.K1 > 0 && j.K1 < P) ? j.K1 : P; }

// ---- blend (per-tile alpha compositing) launches: one or several views per launch ---------------
struct BlendView {
  int32_t W, H, grid_x, ntiles, block_base;   // block_base is filled by the launcher
  const uint2* ranges;
  const uint32_t* point_list;
  const uint2* ranges2;    // segment 2 of the tile lists (two-round binning): list position q >= len(segment 1) reads
  const uint32_t* point_list2;   //   point_list2[ranges2[tile].x + q - len1]
  uint32_t* tile_work;     // forward: written; backward scheduling: read
  unsigned long long* open_rows;  // forward, round 0: bitmap of the unterminated tiles that were not predicted open, i.e.
                           //   whose list is only the K1 prefix (null: not wanted)
  uint32_t* open_count;    //   ... and their number (image header word 3)
  const unsigned long long* pred_rows;   // forward, round 0: the tiles predicted open (complete list in segment 1)
  unsigned long long* pred_next;         //   ... and the prediction for the next forward (every unterminated tile, and the
  const uint32_t* z_clear; //   predicted ones that needed more than the depth key *z_clear = rank 3/4 K1 of the order)
  uint32_t z_base;         //   ... which is stored minus this base when the depth sort ran on 27-bit keys (binning.hip)
  int32_t row_words;       //   64-bit words per tile row of the bitmap
  int32_t round;           // forward: 0 = first pass over all tiles (segment 1); 1 = second pass, only tiles with a segment 2;
                           //          2 = all tiles over segment 1 + segment 2 (re-blend of a finished forward's state)
  uint32_t idx_mask;       // Gaussian index = point_list[j] & idx_mask (packed tile|index words, see b3gs_packed_idx_bits)
  const float4* rec;
  uint8_t* staged;         // forward: GeomView::staged (null: no marks) and the epoch to write there
  const uint32_t* epoch;
  const float* bg;
  float* final_T;          // forward: written; backward: read
  uint32_t* n_contrib;
  float* out_color;        // forward outputs
  float* out_depth;
  float* out_alpha;
  const float* dL_dcolor;  // backward inputs (dL_ddepth / dL_dalpha may be null)
  const float* dL_ddepth;
  const float* dL_dalpha;
  float* dL_dmeans2D;      // backward accumulation targets (rows of 3, 3, 1, cov_stride floats)
  float* dL_dcolors;
  float* dL_dopacity;
  float* dL_dcov3D;
  uint32_t m2d_stride, col_stride, op_stride;  // row strides (floats) of the three arrays above
  uint32_t cov_stride;     // B3GS_SCRATCH_ROW when all four live in one row (both the drop-in and the raw path)
};
struct BlendBatch {
  int32_t n;
  // Backward only: longest-tile-first scheduling.  Workgroup b serves default block 8 * order[(b & 7) * cls_size + (b >> 3)]
  // + (b & 7): inside every XCD class (b & 7: the tiles of one band of every view) the tiles are visited in order of
  // decreasing work, so the launch does not end on a late long tile.  null: default order.
  const uint32_t* order;
  int32_t cls_size;
  uint32_t* order_buf;     // host side: where the launcher may build the order (view 0's ImgView::order), or null
  // Forward: the order the PREVIOUS backward of the same batch shape left in order_buf is reused (tile work changes little
  // between iterations).  It is trusted only when the two words at order[sig_off] carry the magic number and this
  // launch's shape signature: a fresh or differently shaped buffer keeps the default order.
  uint32_t sig_off, sig;
  BlendView v[B3GS_MAX_FUSED_VIEWS];
};
BlendView b3gs_blend_view(const B3gsScene& sc, const GeomView& g, const BinView& b, const ImgView& im);
void b3gs_launch_blend_forward(BlendBatch batch, hipStream_t s);
void b3gs_launch_blend_backward(BlendBatch batch, hipStream_t s);

// one pass over the Gaussians for `nviews` views whose blend backward (phase 1) has completed
struct B3gsViewRef {
  int32_t W, H;
  float tan_fovx, tan_fovy;
  const float* viewmatrix;
  const float* projmatrix;
  const float* campos;
  const int32_t* radii;
  const float* rec_words;    // the view's render records as words (16 per Gaussian): word 12 = the SH clamp bits
  float* scratch;            // the view's phase-1 sums (reset to zero here)
  float* dL_dmeans2D;        // optional
  int32_t densify_stats;
  const uint8_t* staged;     // GeomView::staged / the epoch its forward wrote (null: read every visible Gaussian's row)
  const uint32_t* epoch;
};
void b3gs_launch_accumulate_views(const B3gsScene& base, const B3gsRawParams& raw, int nviews, const B3gsViewRef* views,
                                  const B3gsRawGrads& rg, int overwrite, const B3gsDensifyStats* stats, int first, int count,
                                  uint32_t* list, uint32_t* counts, hipStream_t s);   // list, counts: P words of scratch each
