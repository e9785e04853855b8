// mgs_common.h -- shared host/device definitions for libmgsplat (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/mgsplat.h"

namespace mgs {

constexpr int TILE = 16;         // tile membership granularity: must stay 16 (RAST config.h:17-18)
constexpr int SUB = 8;           // execution granularity: one workgroup per 8x8 pixel block
constexpr int SUBS_PER_TILE = 4;
constexpr int WAVE = 64;
constexpr int CHUNK = 64;        // survivors per chunk of the render kernels (two 32-lane groups of the backward)
#ifndef MGS_PRE_BLOCK
#define MGS_PRE_BLOCK 1024
#endif
constexpr int PRE_BLOCK = MGS_PRE_BLOCK;  // most Gaussians per workgroup of the forward preprocess and of the bin scatter (pre_block())
constexpr int LDS_TILES = 4096;  // max tiles whose per-tile tables fit the binning kernels' LDS (else: tables in memory)
constexpr int SEG_MIN = 512;     // smallest selectable sort segment (sizes the segment table)

// ---- workspace carving (replaces obtain()/fromChunk, RAST rasterizer_impl.h:19-63) -------------
// The layout of every workspace is a function of (its byte size, the problem shape) ONLY -- never of a per-call option --
// so a forward and its backward agree on it without a side channel.
constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) & ~(ALIGN - 1); }

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
  float4* rec;              // [2P]  packed per-Gaussian record, the unit the render kernels gather by id:
                            //       {x, y (pixels), conic.x, conic.y} {conic.z, opacity, hx, hy}; hx, hy = half extents (px)
                            //       of the bbox of {alpha >= 1/255}, <0: never visible
  float* rgb;               // [3P]  SH -> RGB (unused with colors_precomp)
  float* cov3D;             // [6P]
  uint8_t* clamped;         // [P]   bit c set: SH colour channel c was clamped at 0
  uint2* rect;              // [P]   tile rect the Gaussian is binned into: {x0 | x1<<16, y0 | y1<<16} (x1,y1 exclusive)
  uint32_t* flags;          // -> ImgView::flags
  uint32_t* blk_base;       // [preprocess_blocks(P, V)][T]  offset of preprocess workgroup b's instances inside tile t's slice
};

// flags[] words of the img workspace (zeroed per forward)
enum { FLAG_PREFILTERED = 0, FLAG_NUM_RENDERED = 1, FLAG_CHUNKS_USED = 2, FLAG_BLOCKS_DONE = 3 };

struct ImgView {
  float* final_T;         // [N]
  uint2* ranges;          // [T]
  // tile-binning state (mgs_binning.hip); flags .. seg_base are one contiguous block zeroed per forward (LDS tables: by
  // the preprocess's workgroup 0; tables in memory: by a zero-fill launch)
  uint32_t* flags;        // [4]    see FLAG_*
  uint32_t* tile_hist;    // [T]    instances per tile (filled by the forward preprocess)
  uint32_t* seg_base;     // [T+1]  exclusive scan of the tiles' segment counts
  uint32_t* ref_count;    // [1]    instances of the reference's 3-sigma rects (its num_rendered); a spare word of the zeroed block
  size_t zero_bytes;      // bytes from flags to the end of seg_base
  uint32_t* cursor;       // [T]    tables in memory only: the scatter's per-tile write cursors (zeroed by the table kernel)
  // Hand-shake word of the forward preprocess (segment-sort binning): workgroup 0 zeroes the block above and then stores
  // the launch's nonce here; every workgroup waits for the nonce before its first atomic on the block; the bin scatter
  // kernel (next in the chain) stores 0 again.  Replaces a separate zero-fill launch per forward.
  unsigned long long* ready;   // [2]: [0] the hand-shake word, [1] = the launch's nonce if a workgroup gave up waiting for it
  unsigned long long nonce;    // host side only: the nonce of the forward in flight on this workspace (0: none)
  // host side only, "direct" binning (round 6): the forward preprocess of THIS call wrote the keys itself, tile t's unsorted
  // slice at direct_keys[t * direct_stride ...] (nullptr: the bin scatter kernel writes them, compact, into BinView::keys_unsorted)
  uint64_t* direct_keys;
  uint32_t direct_stride;
};
// Direct binning: a tile holds every Gaussian of its view at most once, so a slice of Pg keys per tile can never overflow and
// the preprocess can write a Gaussian's keys the moment it has reserved its workgroup's slots -- no prefix over the tiles, hence
// no scatter kernel (5 us + a kernel boundary per forward).  The bucket rank (mgs_binning.hip) reads the strided slices, scans
// the tile histogram itself and publishes ranges and the status words.  T * Pg keys: 51 MB at BASELINE configs[2], 8 MB at
// ManiGaussian's own shape, 410 MB for 8 views of 100 000; offered up to DIRECT_MAX_KEYS (512 MB: −1 % at 4 / 8 views; the
// 1 GB the configs[4] shape would need buys the same 1 % and is not offered) and whenever the caller's capacity already covers it (the
// worst-case workspaces of the default forward mode do).  Returns the keys the region must hold, 0: not applicable.
constexpr size_t DIRECT_MAX_KEYS = (size_t)64 << 20;
inline size_t direct_keys_needed(size_t P, int V, int T) {
  const size_t v = V > 0 ? (size_t)V : 1, Pg = (P + v - 1) / v;
  return (T > 0 && T <= LDS_TILES && P > 0) ? (size_t)T * Pg : 0;
}

struct BinView {
  uint64_t* keys_unsorted;  // [R]  (depth bits << 32 | id), tile-major, unordered inside a tile
  uint64_t* keys;           // [R]  segment-sorted keys
  uint32_t* point_list;     // [R]  sorted Gaussian ids
  uint4* seg_desc;          // [R/SEG_MIN + T + 66] segment -> {first key, count, tile slice start, tile slice length}
};

// Per-(8x8 block, chunk of 64 survivors) state the render forward keeps for the backward.  Chunk records live in a POOL:
// every round of a block takes its <= 16 consecutive records with one atomic on flags[FLAG_CHUNKS_USED]; round_base maps
// (block, round) -> first record.  pool = number of records the workspace holds (the worst case is 4 * (R/64 + T): every
// chunk of every block visited; a forward typically visits a sixth of that, and the caller may size the pool from the
// high-water mark it has seen: mgs_binning_bytes2).
struct ChunkView {
  uint32_t pool;          // records available
  uint32_t* round_base;   // [4 * (R / 512 + 2T + 2)]  block (tile, sub), round r -> first record (round_entry())
  uint32_t* last_chunk;   // [T*4*64]  per pixel: number of chunks visited by the forward
  float* T_end;           // [pool][64]
  float* T_mid;           // [pool][64]  transmittance after the first 32 survivors of a chunk
  uint32_t* last_pos;     // [pool][64]
  float* q;               // [pool][64]  backward scratch: dL . partial
  float* partial;         // [pool][3+F][64]
  uint32_t* surv;         // [4][surv_stride]  block (tile, sub): surv[sub * surv_stride + ranges[tile].x + i] = instance id
  size_t surv_stride;     // = capacity of the instance list
  uint2* nsurv;           // [T*4]  {survivors found by the forward (a prefix of the block's full list), first record of round 0}
};

// Gaussians per workgroup of the forward preprocess and of the bin scatter for views of Pg Gaussians each.  Both kernels are
// chains of dependent round trips on few workgroups, and every workgroup reserves its slots with one returning atomic per tile of
// its view it has instances in -- Pg / pre_block atomics per tile counter.  Up to 131 072 Gaussians per view half-size workgroups
// put twice as many CUs to work (preprocess 12.8 -> 10.7 us at ManiGaussian's own 16 384, 51 -> 42 us for 8 views of 100 000,
// equal for one view of 100 000); at 500 000 the 977 atomics per counter cost 5 us more than the CUs give
// (profiles/EXPERIMENTS.md part A0).
inline int pre_block(size_t Pg) { return Pg <= 131072 ? PRE_BLOCK / 2 : PRE_BLOCK; }
// P: (virtual) Gaussians = V * Pg for a batch of V views; the preprocess and scatter kernels launch V * ceil(Pg / pre_block(Pg))
// workgroups (a workgroup never straddles views), and blk_base has one row per launched workgroup.
inline size_t preprocess_blocks(size_t P, int V) {
  const size_t v = V > 0 ? (size_t)V : 1, Pg = (P + v - 1) / v, pb = (size_t)pre_block(Pg);
  return v * ((Pg + pb - 1) / pb);
}
inline GeomView carve_geom(void* p, int P, int M, int T, int V, size_t* total) {
  Carver c(p);
  GeomView g;
  size_t Pa = P > 0 ? (size_t)P : 1;
  g.depths = c.take<float>(Pa);
  g.rec = c.take<float4>(2 * Pa);
  g.rect = c.take<uint2>(Pa);
  g.rgb = c.take<float>(3 * Pa);
  g.cov3D = c.take<float>(6 * Pa);
  g.clamped = c.take<uint8_t>(Pa);
  g.flags = nullptr;
  g.blk_base = c.take<uint32_t>(preprocess_blocks(Pa, V) * (size_t)(T > 0 && T <= LDS_TILES ? T : 0) + 1);
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
  v.ranges = c.take<uint2>(T ? T : 1);
  const size_t S = T ? T : 1;
  const size_t nz = (4 + 2 * S + 1 + 63) & ~(size_t)63;  // whole 256-B units: one fill, no tail
  v.flags = c.take<uint32_t>(nz);  // flags | hist | seg_base, contiguous
  v.tile_hist = v.flags ? v.flags + 4 : nullptr;
  v.seg_base = v.flags ? v.tile_hist + S : nullptr;
  v.ref_count = v.flags ? v.seg_base + S + 1 : nullptr;  // (4 + 2S + 1 is odd, nz a multiple of 64: the word exists and is zeroed)
  v.zero_bytes = nz * sizeof(uint32_t);
  v.ready = c.take<unsigned long long>(2);
  v.cursor = c.take<uint32_t>(S);
  v.nonce = 0ull;
  v.direct_keys = nullptr;
  v.direct_stride = 0u;
  if (total) *total = c.total();
  return v;
}

// Records of the worst case: every chunk of every 8x8 block visited; a block takes its records 16 at a time (one round of
// the render forward), so every block may leave up to 15 of its last round unused.
inline uint32_t chunk_pool_max(size_t R, int T) { return (uint32_t)(4 * ((R + CHUNK - 1) / CHUNK + 16 * (size_t)T)); }
// Entries of the (block, round) -> record table (round_entry() in mgs_render_common.h: granule 512 survivors).
inline size_t round_table_entries(size_t R, int T) { return 4 * (R / 512 + 2 * (size_t)T + 2); }

// R: capacity of the instance list; pool: chunk records (0: the worst case for R).
inline BinView carve_binning(void* p, int R, int T, int F, uint32_t pool, ChunkView* cv, size_t* total) {
  Carver c(p);
  BinView b;
  size_t Ra = R > 0 ? (size_t)R : 1;
  b.keys_unsorted = c.take<uint64_t>(Ra);
  b.keys = c.take<uint64_t>(Ra);
  b.point_list = c.take<uint32_t>(Ra);
  b.seg_desc = c.take<uint4>(Ra / SEG_MIN + (size_t)T + 66);  // (+64: the merge kernel's grid is rounded up to 64 and every workgroup reads its entry)
  ChunkView v;
  v.pool = pool ? pool : chunk_pool_max(Ra, T);
  const size_t items = v.pool;
  v.round_base = c.take<uint32_t>(round_table_entries(Ra, T));
  v.last_chunk = c.take<uint32_t>((size_t)T * 4 * 64);
  v.nsurv = c.take<uint2>((size_t)T * 4);
  v.surv = c.take<uint32_t>(4 * Ra);
  v.surv_stride = Ra;
  v.T_end = c.take<float>(items * 64);
  v.T_mid = c.take<float>(items * 64);
  v.last_pos = c.take<uint32_t>(items * 64);
  v.q = c.take<float>(items * 64);
  v.partial = c.take<float>(items * (size_t)(3 + F) * 64);
  if (cv) *cv = v;
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

// ---- per-call options: MgsOptions of include/mgsplat.h with the defaults filled in -------------------------------
struct Options {
  int tight_bins = 1;      // 1: drop (Gaussian,tile) instances whose alpha>=1/255 footprint misses the tile
  int fast_exp = 0;        // 0: ocml's expf, bit for bit (exp_ocml_unclamped), 1: v_exp_f32 of x log2(e) (rel. error ~2e-7 |x|)
  int exact_cull = 1;      // exact ellipse-vs-block cull on top of the bbox cull in the render forward
  int bin_mode = 2;        // 2: tables in LDS + ONE bucket-rank launch; 1: tables in LDS (up to LDS_TILES tiles) + segment sort + rank
                           // merge; 0: tables in memory + segment sort + rank merge (see mgs_binning.hip)
  int rank_mode = 1;       // derived from bin_mode: 1 = bucket rank, 0 = segment sort + rank merge
  int seg = 2048;          // bin_mode 1: entries per LDS-sorted segment (512, 1024 or 2048)
  int gm_waves = 12;       // render backward at one workgroup per CU: 12 = 12 waves, two pixels per step; 16 / 8 = the one-pixel forms
  int dbg = 0;             // see RenderArgs::dbg
  int table_init = 0;      // 0: the preprocess launch zeroes its tables itself (workgroup 0 + hand-shake); 1: a zero-fill launch first
};
// process-wide DIAGNOSTIC state only (never results or layouts): stage timers
int profile_level();

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
  uint32_t* tile_hist;  // [T] instance histogram per tile
  uint32_t* blk_base;   // [gridDim][T] LDS tables: the workgroups' reservations; nullptr: tables in memory (the caller zeroed
                        //              the flags | tile_hist | seg_base block; one atomic per instance on tile_hist)
  float4* zero_ptr;     // optional: block the kernel zeroes on the side (the later backward's accumulators)
  size_t zero_f4;       // ... in float4 units
  int zero_blocks;      // workgroups at the end of the grid that do nothing else (set by launch_preprocess_fwd)
  uint32_t* ref_count;  // ImgView::ref_count (inside the zeroed block)
  uint32_t* tables;     // LDS tables: the flags | tile_hist | seg_base block, zeroed by workgroup 0 of this launch
  uint32_t tables_words;
  unsigned long long* ready;  // ImgView::ready
  unsigned long long nonce;   // this launch's (non-zero) nonce; 0: the tables were zeroed by an earlier launch, no hand-shake
  int wg0_delay;              // test hook (MgsOptions.dbg & 512): workgroup 0 sleeps this many x ~3 us before it zeroes the tables
  uint64_t* direct_keys;      // direct binning (ImgView::direct_keys): this launch writes the keys; nullptr: the bin scatter does
  uint32_t direct_stride;     // ... keys per tile slice (= Pg)
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  int prefiltered, tight_bins;
  const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
  const float *viewmatrix, *projmatrix, *campos;
};
hipError_t launch_preprocess_fwd(const FwdPreArgs& a, const GeomView& g, int32_t* radii, hipStream_t s);
hipError_t launch_zero_bytes(void* p, size_t bytes, hipStream_t s);
// Status the device reports to the host (pinned, device-mapped memory): word 0 = tag<<48 | flags<<32 | num_rendered, written
// by the bin scatter kernel as soon as the preprocess is done; word 1 = tag<<48 | overflow<<32 | chunk records used, written
// by the last workgroup of the render forward.  tag: 16 bits chosen by the caller (a stale write is recognisable).
struct StatusSink { uint64_t* host; uint32_t tag; };
// scatter -> segment sort -> rank merge
unsigned long long next_nonce();  // process-wide counter (never 0) mixed with a per-process random word
// (three launches, issued one by one so that the stage timers of mgs_api.hip see each kernel: `which` = 0 scatter, 1 segment
//  sort, 2 rank merge)
// lds_tables: the scatter keeps the per-tile tables in LDS and uses the preprocess's reservations (T <= LDS_TILES); else a
// one-workgroup table kernel + a scatter with one atomic per instance on cursors in memory (any tile count).
// bucket: stage 1 is the one-launch bucket rank (mgs_binning.hip, round 6) and stage 2 is empty, instead of segment sort + rank merge
hipError_t launch_bin_segsort(int which, bool lds_tables, bool bucket, const GeomView& g, const BinView& b, const ImgView& im, int Pg,
                              int V, int capacity, int tiles_x, int tiles_y, int seg, int dbg, StatusSink status, hipStream_t s);
hipError_t launch_mark_visible(int P, const float* means3D, const float* view, const float* proj,
                               uint8_t* present, hipStream_t s);

// Waves (= chunks per round) of the render forward, RenderArgs::nwf: 16 waves x 128 registers fill a CU with ONE workgroup;
// wide rows (F = 64) need 256 registers (8 waves).  When the grid has more blocks than the chip has CUs (multi-view
// batches, images above 128 x 128), 8-wave workgroups are used as well, TWO per CU (<= 80 KB of LDS, 128 registers): a CU then
// interleaves the latency chains of two pixel blocks instead of idling on one.
inline int fwd_waves(int F, int tiles) { return (F > 32 || 4 * tiles > 256) ? 8 : 16; }

struct RenderArgs {
  int W, H, tiles_x, tiles_y, F, include_feature, fast_exp, exact_cull, gm_waves;
  int nwf;  // waves of the render forward = chunk records per round (fwd_waves(): a function of F and the tile count)
  // Multi-view batches render into an ATLAS: V views stacked vertically, each padded to Hp = tiles_y_view * 16 rows, so
  // that binning and compositing see one image of H = V * Hp rows (V == 1: H is the image height, Hv == H).
  // Instance ids are then "virtual": id = view * Pg + Gaussian.
  int V, Pg, Hv, Hp;
  int colors_per_view;  // 1: `colors` is indexed by the virtual id (SH colours, per view), 0: by the Gaussian
  int dbg;  // timing experiments only (results invalid when non-zero); 256: phase timeline of the forward
  const float* bg;
  const float* colors;   // [P,3] colors_precomp or geom.rgb
  const float* feats;    // [P,F]
  const float4* rec;     // geom.rec: [V*P][2] packed per-Gaussian record (the render kernels gather it by id)
};

hipError_t launch_render_fwd_dense(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                   float* out_color, float* out_feat, StatusSink status, hipStream_t s);
hipError_t launch_render_bwd_gm(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                const float* dL_dcolor_px, const float* dL_dfeat_px, float* acc8, float* dL_dcolors,
                                float* dL_dfeat, hipStream_t s);

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
hipError_t launch_calibration(int iters, float* sink, hipStream_t s);

}  // namespace mgs
