// mgs_common.h -- shared host/device definitions for libmgsplat (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mgsplat.h"

namespace mgs {

constexpr int TILE = 16;         // tile membership granularity: must stay 16 (RAST config.h:17-18)
constexpr int SUB = 8;           // execution granularity: one wave64 per 8x8 pixel block
constexpr int SUBS_PER_TILE = 4;
constexpr int WAVE = 64;
constexpr int PRE_BLOCK = 1024;   // Gaussians per workgroup of the forward preprocess and of the bin scatter (must match)
// Sort slices.  The segment-sort binning sorts "slices" of the instance list independently: one slice per tile (bin_mode 1)
// or one per (tile, coarse depth bucket) (bin_mode 2: buckets are disjoint depth ranges in increasing order, so a tile's
// slices concatenated ARE its sorted list, and slices are small enough to be single sort segments -- no merge pass).
// NB = buckets per tile = the largest power of two with T * NB <= LDS_TILES; tables are always carved for T * NB slices.
inline int bin_buckets_max(int T);
constexpr int LDS_TILES = 4096;   // max tiles whose per-tile tables fit the binning kernels' LDS (else: legacy rocPRIM binning)

// ---- workspace carving (replaces obtain()/fromChunk, RAST rasterizer_impl.h:19-63) -------------
constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) & ~(ALIGN - 1); }

inline int bin_buckets_max(int T) {
  int nb = 1;
  while (T > 0 && (long long)T * nb * 2 <= 4096 && nb < 64) nb *= 2;
  return nb;
}
inline int bin_slices_max(int T) { return T * bin_buckets_max(T); }
struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)), off(0) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
  size_t total() const { return align_up(off) + ALIGN; }
};

// Per-Gaussian state written by the forward preprocess (SoA).
struct GeomView {
  float* depths;            // [P]   view-space z
  float4* rec;              // [2P]  packed per-Gaussian record, the unit the binning gathers:
                            //       {x, y (pixels), conic.x, conic.y} {conic.z, opacity, hx, hy}; hx, hy = half extents (px)
                            //       of the bbox of {alpha >= 1/255}, <0: never visible
  float* rgb;               // [3P]  SH -> RGB (unused with colors_precomp)
  float* cov3D;             // [6P]
  uint8_t* clamped;         // [P]   bit c set: SH colour channel c was clamped at 0
  uint2* rect;              // [P]   tile rect the Gaussian is binned into: {x0 | x1<<16, y0 | y1<<16} (x1,y1 exclusive)
  uint32_t* tiles_touched;  // [P]   (legacy binning only)
  uint32_t* point_offsets;  // [P]   inclusive scan of tiles_touched (legacy binning only)
  uint32_t* flags;          // [4]   [0]: prefiltered violation, [1]: number of instances (bin_mode 1)
  uint32_t* blk_base;       // [ceil(P/PRE_BLOCK)][T]  offset of preprocess workgroup b's instances inside tile t's slice
  void* scan_temp;
  size_t scan_temp_bytes;
};

struct ImgView {
  float* final_T;         // [N]
  uint32_t* n_contrib;    // [N]  1-based position in the tile list of the last blended entry
  uint2* ranges;          // [T]
  // tile-binning state (mgs_binning.hip); flags .. seg_base are one contiguous block zeroed per forward
  uint32_t* flags;        // [4]    [0]: prefiltered violation, [1]: number of instances (bin_mode 1)
  uint32_t* tile_hist;    // [T]    instances per tile (filled by the forward preprocess)
  uint32_t* tile_cursor;  // [T]    (unused)
  uint32_t* seg_base;     // [T+1]  exclusive scan of the tiles' segment counts
  size_t zero_bytes;      // bytes from flags to the end of seg_base
};

// Segment-sort binning (bin_mode 1): a tile's instance list of L entries is cut into ceil(L/SEG) equal segments.
constexpr int SEG_MIN = 512;    // smallest selectable segment size (sizes the segment table)

struct BinView {
  uint64_t* keys_unsorted;  // [R]  (depth bits << 32 | id), tile-major, unordered inside a tile  (legacy: tile<<32|depth)
  uint64_t* keys;           // [R]  segment-sorted keys                                            (legacy: sorted keys)
  uint32_t* vals_unsorted;  // [R]  legacy only
  uint32_t* point_list;     // [R]  sorted Gaussian ids
  float4* inst;             // [2R] sorted packed records {x,y,cx,cy}{cz,opacity,hx,hy}
  uint4* seg_desc;          // [R/SEG_MIN + T + 2]  segment -> {first key, count, tile slice start, tile slice length}
  void* sort_temp;          // legacy only
  size_t sort_temp_bytes;
};

// Per-(tile, 8x8 block, chunk) state of the chunk-parallel render (mgs_render_chunked.hip).
struct ChunkView {
  int CH;                 // entries per chunk
  int max_chunks;         // upper bound of sum_t ceil(len_t / CH) = R/CH + T
  uint32_t* chunk_base;   // [T+2]
  uint32_t* last_chunk;   // [T*4*64]  per pixel: number of chunks visited by the forward
  float* Tprod;           // [items][64]
  float* T_end;           // [items][64]
  uint32_t* last_pos;     // [items][64]
  float* q;               // [items][64]
  float* partial;         // [items][3+F][64]
  // survivor-dense chunks (fwd_mode 2): per 8x8 block the in-order list of entries that reach it, kept for the backward
  float* T_mid;           // [items][64]  transmittance after the first 32 survivors of a chunk
  uint32_t* surv;         // [4][surv_stride]  block (tile, sub): surv[sub * surv_stride + ranges[tile].x + i] = instance id
  size_t surv_stride;     // = capacity of the instance list
  uint32_t* nsurv;        // [T*4]  survivors found by the forward (a prefix of the block's full list)
};

size_t scan_temp_bytes(int P);
size_t sort_temp_bytes(int R);

inline GeomView carve_geom(void* p, int P, int M, int T, size_t* total) {
  Carver c(p);
  GeomView g;
  size_t Pa = P > 0 ? (size_t)P : 1;
  g.depths = c.take<float>(Pa);
  g.rec = c.take<float4>(2 * Pa);
  g.rect = c.take<uint2>(Pa);
  g.rgb = c.take<float>(3 * Pa);
  g.cov3D = c.take<float>(6 * Pa);
  g.clamped = c.take<uint8_t>(Pa);
  g.tiles_touched = c.take<uint32_t>(Pa);
  g.point_offsets = c.take<uint32_t>(Pa);
  g.flags = c.take<uint32_t>(4);
  g.blk_base = c.take<uint32_t>(((Pa + PRE_BLOCK - 1) / PRE_BLOCK) * (size_t)(T > 0 && T <= LDS_TILES ? bin_slices_max(T) : 0) + 1);
  g.scan_temp_bytes = scan_temp_bytes((int)Pa);
  g.scan_temp = c.take<char>(g.scan_temp_bytes);
  (void)M;
  if (total) *total = c.total();
  return g;
}

inline ImgView carve_img(void* p, int W, int H, size_t* total) {
  Carver c(p);
  ImgView v;
  size_t N = (size_t)W * H;
  size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
  v.final_T = c.take<float>(N ? N : 1);
  v.n_contrib = c.take<uint32_t>(N ? N : 1);
  v.ranges = c.take<uint2>(T ? T : 1);
  const size_t S = T ? (T <= (size_t)LDS_TILES ? (size_t)bin_slices_max((int)T) : T) : 1;  // sort slices (>= tiles)
  const size_t nz = (4 + 3 * S + 1 + 63) & ~(size_t)63;  // whole 256-B units: one fill kernel, no tail
  v.flags = c.take<uint32_t>(nz);  // flags | hist | cursor | seg_base, contiguous
  v.tile_hist = v.flags ? v.flags + 4 : nullptr;
  v.tile_cursor = v.flags ? v.tile_hist + S : nullptr;
  v.seg_base = v.flags ? v.tile_hist + 2 * S : nullptr;
  v.zero_bytes = nz * sizeof(uint32_t);
  if (total) *total = c.total();
  return v;
}

inline BinView carve_binning(void* p, int R, int T, int F, int CH, bool legacy, ChunkView* cv, size_t* total) {
  Carver c(p);
  BinView b;
  size_t Ra = R > 0 ? (size_t)R : 1;
  b.keys_unsorted = c.take<uint64_t>(Ra);
  b.keys = c.take<uint64_t>(Ra);
  b.vals_unsorted = c.take<uint32_t>(Ra);
  b.point_list = c.take<uint32_t>(Ra);
  b.inst = c.take<float4>(2 * Ra);
  b.seg_desc = c.take<uint4>(Ra / SEG_MIN + (size_t)(T <= LDS_TILES ? bin_slices_max(T) : T) + 2);
  b.sort_temp_bytes = legacy ? sort_temp_bytes((int)Ra) : 0;
  b.sort_temp = c.take<char>(b.sort_temp_bytes);
  if (CH > 0) {  // chunk-parallel render state
    ChunkView v;
    v.CH = CH;
    v.max_chunks = (int)((Ra + (size_t)CH - 1) / (size_t)CH) + T;
    const size_t items = 4 * (size_t)(8 * ((v.max_chunks + 7) / 8));
    v.chunk_base = c.take<uint32_t>((size_t)T + 2);
    v.last_chunk = c.take<uint32_t>((size_t)T * 4 * 64);
    v.Tprod = c.take<float>(items * 64);
    v.T_end = c.take<float>(items * 64);
    v.last_pos = c.take<uint32_t>(items * 64);
    v.q = c.take<float>(items * 64);
    v.partial = c.take<float>(items * (size_t)(3 + F) * 64);
    v.T_mid = c.take<float>(items * 64);
    v.surv = c.take<uint32_t>(4 * Ra);
    v.surv_stride = Ra;
    v.nsurv = c.take<uint32_t>((size_t)T * 4);
    if (cv) *cv = v;
  }
  if (total) *total = c.total();
  return b;
}

// Backward scratch: per-Gaussian accumulators the render backward adds into.
struct BwdScratch {
  float* acc8;      // [P][8]: dmean2D.x, dmean2D.y, dconic.x, dconic.y, dconic.w, dopacity, -, -
};
inline BwdScratch carve_bwd(void* p, int P, int M, int F, size_t* total) {
  Carver c(p);
  BwdScratch s;
  size_t Pa = P > 0 ? (size_t)P : 1;
  s.acc8 = c.take<float>(8 * Pa);
  (void)M; (void)F;
  if (total) *total = c.total();
  return s;
}

// ---- run-time options (mgs_set_option) ----------------------------------------------------------
struct Options {
  int tight_bins = 1;      // 1: drop (Gaussian,tile) instances whose alpha>=1/255 footprint misses the tile
  int bwd_reduce = 1;      // 0: shuffle reference reduction, 1: butterfly (permlane swap + DPP)
  int fast_exp = 1;        // 1: v_exp_f32 based exp in the render kernels (rel. error ~2e-7 |x|), 0: ocml expf
  int profile = 0;         // 0: off, 1: hipEvents around the render backward only, 2: around every stage
  // The next two shape the binning workspace: do not change them between a forward and its backward.
  int render_mode = 2;     // 0: one wave per 8x8 block walks the whole tile list, 1: chunk items, 2: cooperative
  int chunk = 64;          // entries per chunk (multiple of 64) for render_mode 1 and 2
  int exact_cull = 1;      // render_mode 2: exact ellipse-vs-block cull on top of the bbox cull
  int bwd_mode = 1;        // render_mode 2, chunk 64: 1 = Gaussian-major backward (scans + fp32 MFMA), 0 = pixel-major + butterfly
  int gm_waves = 16;       // waves per workgroup of the Gaussian-major backward (8 or 16)
  int dbg = 0;             // see RenderArgs::dbg
  int dense_variant = 1;   // fwd_mode 2: survivors per chunk: 1 = 64 (two full backward groups), 2 = 32
  int fwd_mode = 2;        // render_mode 2, chunk 64: 2 = survivor-dense chunks + MFMA blend (mgs_render_dense.hip),
                           // 1 = entry chunks, LDS-staged rows (coop_fwd64_kernel), 0 = original
  int bin_octaves = 4;     // bin_mode 2: the depth buckets span this many octaves from the near plane (0.2)
  int bin_mode = 1;        // 1: histogram + scatter + LDS segment sort + rank merge, 0: legacy rocPRIM scan + radix sort
  int seg = 2048;          // bin_mode 1: entries per LDS-sorted segment (512, 1024 or 2048)
};
Options& options();

// ---- launch wrappers (one per .hip translation unit) --------------------------------------------
void set_error(const char* fmt, ...);

constexpr int MAX_VIEWS = 16;
struct ViewCam {  // per-view camera of a multi-view batch
  float tanfovx, tanfovy, focal_x, focal_y;
  const float *viewmatrix, *projmatrix, *campos;
};

struct FwdPreArgs {
  int P, D, M, W, H, tiles_x, tiles_y;
  int V, Pg, Hp;             // P = V * Pg virtual Gaussians, H = view height, tiles_y = tile rows of ONE view
  int use_cam;               // 1: cameras come from cam[] (the multi-view entry points), 0: from the fields below
  ViewCam cam[MAX_VIEWS];
  uint32_t* tile_hist;  // [T * NB] instance histogram per sort slice (zeroed by the caller), or nullptr (legacy binning)
  uint32_t* blk_base;   // [gridDim][T * NB] (with tile_hist)
  int NB, bshift;       // depth buckets per tile (1: none) and the shift of depth_bucket()
  float4* zero_ptr;     // optional: block the kernel zeroes on the side (the later backward's accumulators)
  size_t zero_f4;       // ... in float4 units
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  int prefiltered, tight_bins;
  const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
  const float *viewmatrix, *projmatrix, *campos;
};
hipError_t launch_preprocess_fwd(const FwdPreArgs& a, const GeomView& g, int32_t* radii, hipStream_t s);
hipError_t launch_scan(const GeomView& g, int P, hipStream_t s);
// bin_mode 1: scatter -> segment sort -> rank merge + emit (hist is the host copy of im.tile_hist)
hipError_t launch_bin_segsort(const GeomView& g, const BinView& b, const ImgView& im, int Pg, int V, int capacity,
                              int NB, int bshift,
                              int tiles_x, int tiles_y, int seg, bool emit_inst, uint64_t* host_status, hipStream_t s);
hipError_t launch_duplicate(const GeomView& g, const BinView& b, const ImgView& im, const int32_t* radii, int P,
                            int R, int tiles_x, int tiles_y, int tight_bins, hipStream_t s);
hipError_t launch_sort(const BinView& b, int R, int tiles_x, int tiles_y, hipStream_t s);
hipError_t launch_ranges(const GeomView& g, const BinView& b, const ImgView& im, int R, hipStream_t s);
hipError_t launch_mark_visible(int P, const float* means3D, const float* view, const float* proj,
                               uint8_t* present, hipStream_t s);

struct RenderArgs {
  int W, H, tiles_x, tiles_y, F, include_feature, fast_exp, bwd_reduce, exact_cull;
  // Multi-view batches render into an ATLAS: V views stacked vertically, each padded to Hp = tiles_y_view * 16 rows, so
  // that binning and compositing see one image of H = V * Hp rows (V == 1: H is the image height, Hv == H).
  // Instance ids are then "virtual": id = view * Pg + Gaussian.
  int V, Pg, Hv, Hp;
  int colors_per_view;  // 1: `colors` is indexed by the virtual id (SH colours, per view), 0: by the Gaussian
  int dbg;  // timing experiments only (results invalid when non-zero): bit0 skip blend, bit1 skip phase A, bit2 skip final sum, bit3 skip row staging
  const float* bg;
  const float* colors;   // [P,3] colors_precomp or geom.rgb
  const float* feats;    // [P,F]
  const float4* rec;     // geom.rec: [V*P][2] packed per-Gaussian record (the dense render kernels gather it by id)
};
hipError_t launch_render_fwd(const RenderArgs& r, const BinView& b, const ImgView& im, float* out_color,
                             float* out_feat, hipStream_t s);
hipError_t launch_render_bwd(const RenderArgs& r, const BinView& b, const ImgView& im, const float* dL_dcolor_px,
                             const float* dL_dfeat_px, float* acc8, float* dL_dcolors, float* dL_dfeat,
                             hipStream_t s);

hipError_t launch_render_fwd_dense(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                   float* out_color, float* out_feat, hipStream_t s);
hipError_t launch_render_fwd_chunked(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                     float* out_color, float* out_feat, hipStream_t s);
hipError_t launch_render_bwd_chunked(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                     const float* dL_dcolor_px, const float* dL_dfeat_px, float* acc8,
                                     float* dL_dcolors, float* dL_dfeat, hipStream_t s);

hipError_t launch_render_fwd_coop(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                  float* out_color, float* out_feat, hipStream_t s);
hipError_t launch_render_bwd_coop(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                  const float* dL_dcolor_px, const float* dL_dfeat_px, float* acc8, float* dL_dcolors,
                                  float* dL_dfeat, hipStream_t s);

hipError_t launch_render_bwd_gm(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                const float* dL_dcolor_px, const float* dL_dfeat_px, float* acc8, float* dL_dcolors,
                                float* dL_dfeat, bool dense, hipStream_t s);

struct BwdPreArgs {
  int P, D, M, W, H;
  int cov3D_per_view;        // cov3D is the forward's [V][P][6] workspace copy (else the caller's [P][6])
  int use_cam;               // 1: cameras come from cam[] (the multi-view entry points)
  int V;                     // views; P = Gaussians (not virtual); radii, clamped, acc8, dL_dcolor, dL_dmeans2D, dL_dconic
  ViewCam cam[MAX_VIEWS];    // are [V][P][.] when V > 1 and cam[v] replaces the single-view camera fields below
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  const float *means3D, *shs, *scales, *rotations, *cov3D, *viewmatrix, *projmatrix, *campos;
  const int32_t* radii;
  const uint8_t* clamped;
  const float* acc8;
  const float* dL_dcolor;  // [P,3] gradient w.r.t. the per-Gaussian RGB
  float *dL_dmeans2D, *dL_dconic, *dL_dopacity, *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales, *dL_drot;
};
hipError_t launch_preprocess_bwd(const BwdPreArgs& a, hipStream_t s);

hipError_t launch_selftest(int* result_dev, hipStream_t s);

}  // namespace mgs
