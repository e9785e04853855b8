// mgs_api.hip -- the C ABI of libmgsplat.so (declared in include/mgsplat.h): argument validation,
// workspace carving, stage sequencing on the caller's stream.  No torch, no global device state.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "mgs_common.h"

namespace mgs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Nonces of the preprocess's table hand-shake: a process-wide counter under a per-process random word, never 0.  Not
// option state and not device state: two calls never share a value, that is all.
unsigned long long next_nonce() {
  static std::atomic<unsigned long long> counter{0};
  static const unsigned long long salt = [] {
    unsigned long long v = 0x9e3779b97f4a7c15ull ^ (unsigned long long)(uintptr_t)&counter;
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) { unsigned long long r = 0; if (fread(&r, sizeof(r), 1, f) == 1) v ^= r; fclose(f); }
    return v << 32;
  }();
  const unsigned long long c = counter.fetch_add(1) + 1;
  const unsigned long long n = salt ^ c ^ (c << 40);
  return n ? n : 1ull;
}

static int g_profile = 0;  // diagnostics only (mgs_set_option("profile", .)): never results or layouts
int profile_level() { return g_profile; }

// MgsOptions of a call with the defaults filled in
static Options options_of(const MgsRasterArgs* a) {
  Options o;
  if (a && a->opt.set) {
    o.tight_bins = a->opt.tight_bins; o.fast_exp = a->opt.fast_exp; o.exact_cull = a->opt.exact_cull;
    o.bin_mode = (a->opt.bin_mode >= 0 && a->opt.bin_mode <= 2) ? a->opt.bin_mode : 2; o.gm_waves = (a->opt.gm_waves == 8 || a->opt.gm_waves == 16) ? a->opt.gm_waves : 12; o.dbg = a->opt.dbg;
    o.seg = (a->opt.seg == 512 || a->opt.seg == 1024 || a->opt.seg == 4096) ? a->opt.seg : 2048;
    o.table_init = a->opt.table_init ? 1 : 0;
  }
  if (a && a->debug) o.table_init = 1;  // a debugged device may hold any workgroup: no workgroup waits for another
  return o;
}

static bool supported_F(int F) {
  switch (F) {
    case 0: case 3: case 4: case 8: case 16: case 32: case 64: return true;
    default: return false;
  }
}

// ---- per-stage timing with hipEvents on the caller's stream (mgs_set_option("profile", 1|2)) ----
enum Stage { ST_PREPROCESS = 0, ST_RENDER_FWD, ST_BWD_MEMSET, ST_RENDER_BWD, ST_PREPROCESS_BWD, ST_BIN_SCATTER, ST_BIN_SEGSORT,
             ST_BIN_MERGE, ST_COUNT };
// (bin_scatter covers the table kernel too when the tables live in memory)
static const char* const kStageNames[ST_COUNT] = {"preprocess_fwd", "render_fwd", "bwd_memset", "render_bwd",
                                                  "preprocess_bwd", "bin_scatter", "bin_segsort", "bin_merge"};
struct Profiler {
  std::mutex mu;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> used[ST_COUNT];
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
};
static Profiler& profiler() { static Profiler p; return p; }

struct StageTimer {  // RAII: records start now, stop at scope exit
  int stage; hipStream_t stream; hipEvent_t e0 = nullptr, e1 = nullptr;
  StageTimer(int st, hipStream_t s) : stage(st), stream(s) {
    const int lvl = g_profile;
    if (lvl == 0 || (lvl == 1 && st != ST_RENDER_BWD)) return;
    Profiler& p = profiler();
    std::lock_guard<std::mutex> lk(p.mu);
    e0 = p.get(); e1 = p.get();
    if (e0 && e1) (void)hipEventRecord(e0, stream); else e0 = e1 = nullptr;
  }
  ~StageTimer() {
    if (!e0) return;
    (void)hipEventRecord(e1, stream);
    Profiler& p = profiler();
    std::lock_guard<std::mutex> lk(p.mu);
    p.used[stage].push_back({e0, e1});
  }
};

#define MGS_HIP(expr, what)                                                      \
  do {                                                                           \
    hipError_t _e = (expr);                                                      \
    if (_e != hipSuccess) {                                                      \
      set_error("%s failed: %s", what, hipGetErrorString(_e));                   \
      return MGS_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)

// debug=1: synchronise and check after a stage (RAST auxiliary.h:166-173 CHECK_CUDA)
#define MGS_STAGE(expr, what, dbg, stream)                                       \
  do {                                                                           \
    MGS_HIP(expr, what);                                                         \
    if (dbg) MGS_HIP(hipStreamSynchronize(stream), what " (debug sync)");        \
  } while (0)

static int check_common(const MgsRasterArgs* a) {
  if (!a) { set_error("args is NULL"); return MGS_ERR_INVALID_ARG; }
  if (a->P < 0 || a->W <= 0 || a->H <= 0) { set_error("bad P/W/H (%d,%d,%d)", a->P, a->W, a->H); return MGS_ERR_INVALID_ARG; }
  if (a->P > 0) {
    if (!a->means3D || !a->viewmatrix || !a->projmatrix || !a->campos || !a->background) {
      set_error("means3D/viewmatrix/projmatrix/campos/background must be non-NULL");
      return MGS_ERR_INVALID_ARG;
    }
    if ((a->shs == nullptr) == (a->colors_precomp == nullptr)) {
      // the reference hard-codes NUM_CHANNELS = 3; with neither it throws (rasterizer_impl.cu:245-248)
      set_error(a->shs ? "both shs and colors_precomp given" : "For non-RGB, provide precomputed Gaussian colors!");
      return a->shs ? MGS_ERR_INVALID_ARG : MGS_ERR_NON_RGB;
    }
    if (a->shs && a->M <= 0) { set_error("shs given but M <= 0"); return MGS_ERR_INVALID_ARG; }
    if (a->shs && (a->D < 0 || (a->D + 1) * (a->D + 1) > a->M || a->D > 3)) {
      set_error("sh_degree %d needs %d coefficients, M = %d (max degree 3)", a->D, (a->D + 1) * (a->D + 1), a->M);
      return MGS_ERR_INVALID_ARG;
    }
    const bool sr = a->scales && a->rotations;
    if (sr == (a->cov3D_precomp != nullptr) || (!sr && (a->scales || a->rotations))) {
      set_error("provide exactly one of scales+rotations or cov3D_precomp");
      return MGS_ERR_INVALID_ARG;
    }
    if (a->include_feature) {
      if (!a->language_feature) { set_error("include_feature set but language_feature is NULL"); return MGS_ERR_INVALID_ARG; }
      if (!supported_F(a->F) || a->F == 0) {
        set_error("feature width F=%d not compiled in (supported: 3,4,8,16,32,64; pad to the next one)", a->F);
        return MGS_ERR_INVALID_ARG;
      }
      // the render forward addresses feature rows by 32-bit byte offsets (buffer loads)
      if ((unsigned long long)a->P * (unsigned long long)a->F * 4ull >= (1ull << 32)) {
        set_error("P * F * 4 = %llu bytes of features: rows are addressed by 32-bit offsets (< 4 GiB)",
                  (unsigned long long)a->P * (unsigned long long)a->F * 4ull);
        return MGS_ERR_INVALID_ARG;
      }
    }
  }
  return MGS_OK;
}

}  // namespace mgs

using namespace mgs;

extern "C" {

int mgs_abi_version(void) { return MGS_ABI_VERSION; }
#ifndef MGS_BUILD_ID
#define MGS_BUILD_ID "unknown"
#endif
const char* mgs_build_id(void) { return MGS_BUILD_ID; }
const char* mgs_last_error(void) { return g_err; }

void mgs_options_default(MgsOptions* o) {
  if (!o) return;
  const Options d;
  o->set = 1; o->tight_bins = d.tight_bins; o->fast_exp = d.fast_exp; o->exact_cull = d.exact_cull;
  o->bin_mode = d.bin_mode; o->seg = d.seg; o->gm_waves = d.gm_waves; o->dbg = d.dbg; o->table_init = d.table_init;
}

int mgs_set_option(const char* key, int value) {
  if (!strcmp(key, "profile")) { g_profile = value; return MGS_OK; }
  set_error("unknown option %s (tuning switches are per call: MgsRasterArgs.opt)", key);
  return MGS_ERR_INVALID_ARG;
}
int mgs_get_option(const char* key) {
  if (!strcmp(key, "profile")) return g_profile;
  set_error("unknown option %s", key);
  return MGS_ERR_INVALID_ARG;
}

static int num_tiles(int W, int H) { return ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE); }
// the bin scatter keeps its per-tile tables in LDS; larger tile grids (or bin_mode 0) keep them in memory
static bool lds_tables(const Options& o, int T) { return o.bin_mode >= 1 && T <= LDS_TILES; }
// ... and the lists are ordered by ONE bucket-rank launch (bin_mode 2) instead of segment sort + rank merge
static bool bucket_rank(const Options& o, int T) { return o.bin_mode == 2 && T <= LDS_TILES; }
size_t mgs_geom_bytes(int P, int M, int W, int H) { size_t t; carve_geom(nullptr, P, M, num_tiles(W, H), 1, &t); return t; }
size_t mgs_img_bytes(int W, int H) { size_t t; carve_img(nullptr, W, H, &t); return t; }
static size_t binning_bytes_T(int R, int pool, int T, int F) {
  size_t t;
  carve_binning(nullptr, R, T, F, pool > 0 ? (uint32_t)pool : 0u, nullptr, &t);
  return t;
}
size_t mgs_binning_bytes2(int R, int chunk_pool, int W, int H, int F) { return binning_bytes_T(R, chunk_pool, num_tiles(W, H), F); }
size_t mgs_binning_bytes(int R, int W, int H, int F) { return binning_bytes_T(R, 0, num_tiles(W, H), F); }
int mgs_chunk_pool_max(int R, int W, int H) { return (int)chunk_pool_max(R > 0 ? (size_t)R : 1, num_tiles(W, H)); }
size_t mgs_backward_scratch_bytes(int P, int M, int F) { size_t t; carve_bwd(nullptr, P, M, F, &t); return t; }

static const uint64_t kStatusPending = ~0ull;

// How the caller carved the binning workspace: {instances it holds, chunk records}.  Explicit (binning_capacity > 0) or,
// for a buffer sized by mgs_binning_bytes, the largest count that fits with a worst-case pool.
struct BinShape { int cap; uint32_t pool; };
static int binning_capacity_uncached(size_t bytes, int T, int F) {
  auto need = [&](int R) { return binning_bytes_T(R, 0, T, F); };
  if (need(0) > bytes) return -1;
  int lo = 0, hi = 1;
  while (hi < (1 << 30) && need(hi) <= bytes) { lo = hi; hi <<= 1; }
  while (hi - lo > 1) {
    const int mid = lo + (hi - lo) / 2;
    if (need(mid) <= bytes) lo = mid; else hi = mid;
  }
  return lo;
}
static BinShape bin_shape(const MgsRasterArgs* a, int T, int F) {
  BinShape s = {-1, 0};
  if (a->binning_capacity > 0) {
    if (binning_bytes_T(a->binning_capacity, a->chunk_pool, T, F) > a->binning_bytes) return s;
    s.cap = a->binning_capacity;
    s.pool = a->chunk_pool > 0 ? (uint32_t)a->chunk_pool : chunk_pool_max((size_t)s.cap, T);
    return s;
  }
  struct Memo { size_t bytes; int T, F, cap; };
  static thread_local Memo memo = {0, -1, -1, -1};
  if (!(memo.bytes == a->binning_bytes && memo.T == T && memo.F == F))
    memo = {a->binning_bytes, T, F, binning_capacity_uncached(a->binning_bytes, T, F)};
  s.cap = memo.cap;
  s.pool = s.cap >= 0 ? chunk_pool_max((size_t)(s.cap > 0 ? s.cap : 1), T) : 0u;
  return s;
}

// Direct binning (mgs_common.h): where the forward preprocess of THIS call may write its keys -- inside the caller's binning
// workspace: the two key arrays if the capacity covers T * Pg keys (the worst-case workspaces of the default forward mode do),
// else bytes the caller added behind the carved arrays (mgs_binning_direct_extra); nullptr: the bin scatter kernel writes
// compact slices as before.  MgsOptions.dbg & 32768 switches it off (A/B).
static uint64_t* direct_region(const MgsRasterArgs* a, const BinShape& bs, int T, int F, size_t P, int V, const Options& o) {
  if (!bucket_rank(o, T) || (o.dbg & 32768)) return nullptr;
  const size_t need = direct_keys_needed(P, V, T);
  if (!need) return nullptr;
  size_t total = 0;
  const BinView b = carve_binning(a->binning, bs.cap, T, F, bs.pool, nullptr, &total);
  const size_t span = (size_t)(reinterpret_cast<const char*>(b.point_list) - reinterpret_cast<const char*>(b.keys_unsorted));
  if (span >= need * sizeof(uint64_t)) return b.keys_unsorted;  // (the sorted-key array is unused by the bucket rank)
  const size_t off = (total + 255) & ~(size_t)255;
  if (need <= DIRECT_MAX_KEYS && a->binning_bytes >= off + need * sizeof(uint64_t))
    return reinterpret_cast<uint64_t*>(static_cast<char*>(a->binning) + off);
  return nullptr;
}
size_t mgs_binning_direct_extra(int P, int V, int W, int H) {
  const int v = V > 0 ? V : 1;
  const size_t need = direct_keys_needed((size_t)(P > 0 ? P : 0) * v, v, num_tiles(W, H) * v);
  return need && need <= DIRECT_MAX_KEYS ? need * sizeof(uint64_t) + 256 : 0;
}

// Everything of the forward before the instance count is known: (zero tables,) preprocess.
// direct_keys: see direct_region (nullptr from the two-call path: its second call may come with another workspace).
static int enqueue_preprocess(const MgsRasterArgs* a, const Options& o, int32_t* radii, hipStream_t stream, GeomView& g,
                              ImgView& im, bool& lds, uint64_t* direct_keys = nullptr) {
  if (!radii || !a->opacities) { set_error("radii/opacities must be non-NULL"); return MGS_ERR_INVALID_ARG; }
  if (!a->geom || a->geom_bytes < mgs_geom_bytes(a->P, a->M, a->W, a->H) || !a->img ||
      a->img_bytes < mgs_img_bytes(a->W, a->H)) {
    set_error("geom/img workspace too small: %zu < %zu or %zu < %zu", a->geom_bytes,
              mgs_geom_bytes(a->P, a->M, a->W, a->H), a->img_bytes, mgs_img_bytes(a->W, a->H));
    return MGS_ERR_WORKSPACE;
  }
  g = carve_geom(a->geom, a->P, a->M, num_tiles(a->W, a->H), 1, nullptr);
  im = carve_img(a->img, a->W, a->H, nullptr);
  g.flags = im.flags;
  FwdPreArgs p;
  p.V = 1; p.Pg = a->P; p.Hp = 0; p.use_cam = 0;
  p.P = a->P; p.D = a->D; p.M = a->M; p.W = a->W; p.H = a->H;
  p.tiles_x = (a->W + TILE - 1) / TILE; p.tiles_y = (a->H + TILE - 1) / TILE;
  p.tanfovx = a->tanfovx; p.tanfovy = a->tanfovy;
  p.focal_y = a->H / (2.0f * a->tanfovy);  // rasterizer_impl.cu:225-226
  p.focal_x = a->W / (2.0f * a->tanfovx);
  p.scale_modifier = a->scale_modifier;
  p.prefiltered = a->prefiltered; p.tight_bins = o.tight_bins;
  p.means3D = a->means3D; p.shs = a->shs; p.colors_precomp = a->colors_precomp; p.opacities = a->opacities;
  p.scales = a->scales; p.rotations = a->rotations; p.cov3D_precomp = a->cov3D_precomp;
  p.viewmatrix = a->viewmatrix; p.projmatrix = a->projmatrix; p.campos = a->campos;
  lds = lds_tables(o, p.tiles_x * p.tiles_y);
  // LDS tables, table_init 0: workgroup 0 of the preprocess launch zeroes the tables and the others wait for its nonce (one
  // launch fewer; relies on workgroup 0 being dispatched first, bounded wait).  Otherwise a zero-fill launch of its own, ahead
  // of the preprocess in stream order: nonce 0, no workgroup waits for another.
  const bool handshake = lds && !o.table_init;
  if (!handshake) MGS_HIP(launch_zero_bytes(im.flags, im.zero_bytes, stream), "zero flags + tile tables");
  p.tables = im.flags; p.tables_words = (uint32_t)(im.zero_bytes / 4); p.ready = im.ready;
  p.nonce = handshake ? next_nonce() : 0ull;
  p.wg0_delay = (o.dbg & 1024) ? -1 : (o.dbg & 512) ? 100 : 0;
  im.nonce = p.nonce;
  p.zero_ptr = nullptr; p.zero_f4 = 0;
  if (a->bwd_accum) {
    if ((reinterpret_cast<uintptr_t>(a->bwd_accum) & 15u) || (a->bwd_accum_bytes & 15u)) {
      set_error("bwd_accum must be 16-byte aligned and a multiple of 16 bytes");
      return MGS_ERR_INVALID_ARG;
    }
    p.zero_ptr = reinterpret_cast<float4*>(a->bwd_accum);
    p.zero_f4 = a->bwd_accum_bytes / 16;
  }
  p.tile_hist = im.tile_hist;
  p.blk_base = lds ? g.blk_base : nullptr;
  p.ref_count = im.ref_count;
  p.direct_keys = lds ? direct_keys : nullptr; p.direct_stride = (uint32_t)a->P;
  { StageTimer t(ST_PREPROCESS, stream);
    MGS_STAGE(launch_preprocess_fwd(p, g, radii, stream), "preprocess", a->debug, stream); }
  return MGS_OK;
}

// Blocking read-back of {instance count, flags} (the reference's cudaMemcpy, rasterizer_impl.cu:284).
static const char* const kHandshakeMsg =
    "the forward preprocess gave up waiting for its zeroed tile tables (a workgroup of the launch did not make progress for "
    "about a second): nothing was binned for this call";
// *R_ref: the reference's count (instances of the 3-sigma rects).
static int read_count_blocking(const GeomView& g, const ImgView& im, hipStream_t stream, uint32_t* R, uint32_t* fl,
                               uint32_t* R_ref) {
  uint32_t host[2] = {0, 0};
  uint32_t ref = 0;
  MGS_HIP(hipMemcpyAsync(&ref, im.ref_count, sizeof(ref), hipMemcpyDeviceToHost, stream), "reference-count read-back");
  unsigned long long mark = 0ull;
  if (im.nonce)
    MGS_HIP(hipMemcpyAsync(&mark, im.ready + 1, sizeof(mark), hipMemcpyDeviceToHost, stream), "hand-shake read-back");
  MGS_HIP(hipMemcpyAsync(&host[0], g.flags + FLAG_NUM_RENDERED, sizeof(uint32_t), hipMemcpyDeviceToHost, stream),
          "num_rendered read-back");
  MGS_HIP(hipMemcpyAsync(&host[1], g.flags + FLAG_PREFILTERED, sizeof(uint32_t), hipMemcpyDeviceToHost, stream),
          "flag read-back");
  MGS_HIP(hipStreamSynchronize(stream), "stream sync");
  *R = host[0]; *fl = host[1];
  if (R_ref) *R_ref = ref;
  if (im.nonce && mark == im.nonce) { set_error("%s", kHandshakeMsg); return MGS_ERR_HIP; }
  return MGS_OK;
}

// The device counts instances in 32 unsigned bits; the C ABI (like the reference, rasterizer_impl.cu:282 `int num_rendered`)
// hands them out as int32: saturate instead of wrapping negative (advisor r4; 2^31 instances would need > 80 GB of lists).
static inline int32_t sat_i32(uint32_t v) { return v > 0x7fffffffu ? 0x7fffffff : (int32_t)v; }

// status word layout: tag (16) | flags (16) | count (32)
static inline bool status_arrived(uint64_t w, uint32_t tag) { return w != kStatusPending && (uint32_t)(w >> 48) == (tag & 0xffffu); }

// Wait for word 0 = {tag, flags, R} on the pinned status block (written by the binning kernel right after the preprocess).
// *R_ref: the reference's 3-sigma-rect count from word 2, which the device stores BEFORE word 0 (release order).
static int wait_status(uint64_t* host_status, uint32_t tag, hipStream_t stream, uint32_t* R, uint32_t* fl, uint32_t* R_ref) {
  volatile uint64_t* hs = host_status;
  uint64_t st = *hs;
  for (uint64_t spins = 0; !status_arrived(st, tag); spins++) {
    if ((spins & 0x3ff) == 0x3ff) {  // every ~1k polls: has the stream died or finished without reporting?
      hipError_t q = hipStreamQuery(stream);
      if (q != hipErrorNotReady) {
        st = *hs;
        if (status_arrived(st, tag)) break;
        set_error("forward finished without reporting the instance count: %s", hipGetErrorString(q));
        return MGS_ERR_HIP;
      }
    }
    __builtin_ia32_pause();
    st = *hs;
  }
  *R = (uint32_t)st; *fl = (uint32_t)(st >> 32) & 0xffffu;
  uint64_t w2 = hs[2];
  for (int spins = 0; !status_arrived(w2, tag) && spins < (1 << 20); spins++) { __builtin_ia32_pause(); w2 = hs[2]; }
  if (!status_arrived(w2, tag)) { set_error("forward reported its instance count without the reference count"); return MGS_ERR_HIP; }
  if (R_ref) *R_ref = (uint32_t)w2;
  return MGS_OK;
}

static int check_prefiltered(uint32_t fl) {
  if (fl & 2u) { set_error("%s", kHandshakeMsg); return MGS_ERR_HIP; }  // (status-word flag of the bin scatter kernel)
  if (fl & 1u) {
    set_error("Point is filtered although prefiltered is set. This shouldn't happen!");  // auxiliary.h:158
    return MGS_ERR_INVALID_ARG;
  }
  return MGS_OK;
}

static RenderArgs render_args(const MgsRasterArgs* a, const Options& o, const GeomView& g) {
  RenderArgs r;
  const int F = a->include_feature ? a->F : 0;
  r.W = a->W; r.H = a->H; r.tiles_x = (a->W + TILE - 1) / TILE; r.tiles_y = (a->H + TILE - 1) / TILE;
  r.F = F; r.include_feature = F > 0;
  r.fast_exp = o.fast_exp; r.exact_cull = o.exact_cull; r.gm_waves = o.gm_waves; r.dbg = o.dbg & ~(512 | 1024);
  r.nwf = fwd_waves(F, r.tiles_x * r.tiles_y);
  r.V = 1; r.Pg = a->P; r.Hv = a->H; r.Hp = a->H; r.colors_per_view = 0;
  r.bg = a->background;
  r.colors = a->colors_precomp ? a->colors_precomp : g.rgb;
  r.feats = a->language_feature;
  r.rec = g.rec;
  return r;
}

// Binning + render.  R: instance count if the host knows it (checked against the capacity here), else -1.
// status: where the device reports (see StatusSink), host == nullptr: nowhere.
// direct_keys: the preprocess of this call wrote the keys there (direct_region); nullptr: the bin scatter kernel runs.
static int enqueue_render(const MgsRasterArgs* a, const Options& o, int R, const int32_t* radii, float* out_color,
                          float* out_feature, StatusSink status, unsigned long long nonce, hipStream_t stream,
                          uint64_t* direct_keys = nullptr) {
  const int F = a->include_feature ? a->F : 0;
  const int T = num_tiles(a->W, a->H);
  const bool lds = lds_tables(o, T);
  GeomView g = carve_geom(a->geom, a->P, a->M, T, 1, nullptr);
  ImgView im = carve_img(a->img, a->W, a->H, nullptr);
  g.flags = im.flags;
  im.nonce = nonce;  // (0: the caller has already looked at the preprocess's outcome)
  im.direct_keys = lds ? direct_keys : nullptr; im.direct_stride = (uint32_t)a->P;
  const BinShape bs = bin_shape(a, T, F);
  if (!a->binning || bs.cap < 0 || (R >= 0 && R > bs.cap)) {
    set_error("binning workspace too small: %zu bytes hold %d instances, need %d", a->binning_bytes, bs.cap, R);
    return MGS_ERR_WORKSPACE;
  }
  ChunkView cv;
  BinView b = carve_binning(a->binning, bs.cap, T, F, bs.pool, &cv, nullptr);
  const int tiles_x = (a->W + TILE - 1) / TILE, tiles_y = (a->H + TILE - 1) / TILE;
  (void)radii;
  for (int k = 0; k < 3; k++) {
    if (k == 2 && bucket_rank(o, T)) break;  // (the bucket rank of stage 1 wrote the sorted ids)
    if (k == 0 && bucket_rank(o, T) && im.direct_keys) continue;  // (direct binning: the preprocess wrote the keys, no scatter launch)
    StageTimer t(ST_BIN_SCATTER + k, stream);
    MGS_STAGE(launch_bin_segsort(k, lds, bucket_rank(o, T), g, b, im, a->P, 1, bs.cap, tiles_x, tiles_y, o.seg, o.dbg, status, stream),
              "binning", a->debug, stream);
  }
  const RenderArgs r = render_args(a, o, g);
  { StageTimer t(ST_RENDER_FWD, stream);
    MGS_STAGE(launch_render_fwd_dense(r, b, im, cv, out_color, out_feature, status, stream), "render forward", a->debug,
              stream); }
  return MGS_OK;
}

static int check_render_args(const MgsRasterArgs* a, float* out_color, float* out_feature, hipStream_t stream,
                             bool* done) {
  *done = false;
  if (!out_color) { set_error("out_color is NULL"); return MGS_ERR_INVALID_ARG; }
  const size_t N = (size_t)a->W * a->H;
  const int F = a->include_feature ? a->F : 0;
  if (F > 0 && !out_feature) { set_error("out_feature is NULL"); return MGS_ERR_INVALID_ARG; }
  if (a->P == 0) {  // rasterize_points.cu:70-92: zero-filled outputs, nothing launched
    MGS_HIP(launch_zero_bytes(out_color, 3 * N * sizeof(float), stream), "memset out_color");
    if (F > 0) MGS_HIP(launch_zero_bytes(out_feature, F * N * sizeof(float), stream), "memset out_feature");
    *done = true;
  }
  return MGS_OK;
}

int mgs_rasterize_forward_preprocess(const MgsRasterArgs* a, int32_t* radii, int32_t* num_rendered,
                                     mgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(a);
  if (rc) return rc;
  if (!num_rendered) { set_error("num_rendered is NULL"); return MGS_ERR_INVALID_ARG; }
  *num_rendered = 0;
  if (a->P == 0) return MGS_OK;  // rasterize_points.cu:92
  const Options o = options_of(a);
  GeomView g; ImgView im; bool lds;
  rc = enqueue_preprocess(a, o, radii, stream, g, im, lds);
  if (rc) return rc;
  uint32_t R = 0, fl = 0, R_ref = 0;
  rc = read_count_blocking(g, im, stream, &R, &fl, &R_ref);
  if (rc) return rc;
  rc = check_prefiltered(fl);
  if (rc) return rc;
  *num_rendered = sat_i32(R_ref);  // the reference's integer (>= the instances actually binned: a safe size for stage 2)
  return MGS_OK;
}

int mgs_rasterize_forward_render(const MgsRasterArgs* a, int32_t R, const int32_t* radii, float* out_color,
                                 float* out_feature, mgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(a);
  if (rc) return rc;
  bool done;
  rc = check_render_args(a, out_color, out_feature, stream, &done);
  if (rc || done) return rc;
  if (R < 0 || !radii) { set_error("num_rendered < 0 or radii NULL"); return MGS_ERR_INVALID_ARG; }
  if (!a->geom || a->geom_bytes < mgs_geom_bytes(a->P, a->M, a->W, a->H) || !a->img ||
      a->img_bytes < mgs_img_bytes(a->W, a->H)) {
    set_error("geom/img workspace too small");
    return MGS_ERR_WORKSPACE;
  }
  {  // the render forward's counters (chunk records taken, blocks done) may be left over from an earlier render on this
     // img workspace (the capacity-retry path): reset them; the preprocess's words (flags 0-1, histogram) stay
    ImgView im = carve_img(a->img, a->W, a->H, nullptr);
    MGS_HIP(launch_zero_bytes(im.flags + FLAG_CHUNKS_USED, 2 * sizeof(uint32_t), stream), "reset render counters");
  }
  return enqueue_render(a, options_of(a), R, radii, out_color, out_feature, StatusSink{nullptr, 0}, 0ull, stream);
}

int mgs_rasterize_forward(const MgsRasterArgs* a, int32_t* radii, float* out_color, float* out_feature,
                          int32_t* num_rendered, uint64_t* host_status, mgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(a);
  if (rc) return rc;
  if (!num_rendered) { set_error("num_rendered is NULL"); return MGS_ERR_INVALID_ARG; }
  *num_rendered = 0;
  bool done;
  rc = check_render_args(a, out_color, out_feature, stream, &done);
  if (rc || done) {
    if (!rc && host_status) {  // P == 0: nothing will report; leave a completed status behind
      const uint64_t w = (uint64_t)(a->status_tag & 0xffffu) << 48;
      host_status[0] = w; host_status[1] = w; host_status[2] = w;
    }
    return rc;
  }
  Options o = options_of(a);
  GeomView g; ImgView im; bool lds;
  const int F = a->include_feature ? a->F : 0;
  const BinShape bs = bin_shape(a, num_tiles(a->W, a->H), F);
  if (!a->binning || bs.cap < 0) { set_error("binning workspace missing or smaller than its fixed part"); return MGS_ERR_WORKSPACE; }
  uint64_t* const dk = direct_region(a, bs, num_tiles(a->W, a->H), F, (size_t)a->P, 1, o);
  for (int attempt = 0;; attempt++) {
  rc = enqueue_preprocess(a, o, radii, stream, g, im, lds, dk);
  if (rc) return rc;
  uint32_t R = 0, fl = 0;
  if (!host_status || a->debug) {
    // no device->host status channel: read back (blocking) like the two-call path
    if (a->async_forward) { set_error("async_forward needs host_status and debug == 0"); return MGS_ERR_INVALID_ARG; }
    uint32_t R_ref = 0;
    rc = read_count_blocking(g, im, stream, &R, &fl, &R_ref);
    if (rc) return rc;
    rc = check_prefiltered(fl);
    if (rc) return rc;
    *num_rendered = sat_i32(R_ref);  // the reference's integer; the workspace has to hold the R instances actually binned
    if ((int)R > bs.cap) return MGS_NEED_CAPACITY;
    if (host_status) {  // (debug: the words are reported by the kernels as usual)
      volatile uint64_t* hs = host_status;
      hs[0] = kStatusPending; hs[1] = kStatusPending; hs[2] = kStatusPending;
    }
    return enqueue_render(a, o, (int)R, radii, out_color, out_feature, StatusSink{host_status, a->status_tag}, 0ull, stream, dk);
  }
  // sync-free: everything is enqueued; the binning kernel stores {tag, reference count} and {tag, flags, R} to the mapped host
  // words as soon as the preprocess is done, and refuses to bin (empty ranges, zero segments) when R exceeds the capacity.
  volatile uint64_t* hs = host_status;
  hs[0] = kStatusPending; hs[1] = kStatusPending; hs[2] = kStatusPending;
  rc = enqueue_render(a, o, -1, radii, out_color, out_feature, StatusSink{host_status, a->status_tag}, im.nonce, stream, dk);
  if (rc) return rc;
  if (a->async_forward) { *num_rendered = -1; return MGS_OK; }  // the caller reads mgs_forward_result later
  // Wait for the PREPROCESS only: word 0 arrives when the bin scatter starts; binning and render run on while this call
  // returns.  The reference's integer comes through word 2 of the same block (round 4 read it back with a copy + stream
  // synchronise, i.e. waited for the whole render).
  uint32_t R_ref = 0;
  rc = wait_status(host_status, a->status_tag, stream, &R, &fl, &R_ref);
  if (rc) return rc;
  if ((fl & 2u) && attempt == 0 && !o.table_init) {
    // A preprocess workgroup gave up waiting for the zeroed tables (workgroup 0 of the launch made no progress for about a
    // second): nothing was binned.  Run the forward again with the tables zeroed by a launch of their own -- no workgroup
    // waits for another on that path.  (The first run's kernels still report through the same words: drain them first.)
    MGS_HIP(hipStreamSynchronize(stream), "stream sync before the hand-shake retry");
    o.table_init = 1; o.dbg &= ~(512 | 1024);
    continue;
  }
  rc = check_prefiltered(fl);
  if (rc) return rc;
  *num_rendered = sat_i32(R_ref);
  return (int)R > bs.cap ? MGS_NEED_CAPACITY : MGS_OK;
  }
}

static int forward_result_T(const MgsRasterArgs* a, int T, const uint64_t* host_status, int32_t* num_rendered,
                            int32_t* chunks_used, int32_t* ref_rendered) {
  if (num_rendered) *num_rendered = -1;
  if (chunks_used) *chunks_used = -1;
  if (ref_rendered) *ref_rendered = -1;
  if (!a || !host_status) { set_error("forward_result: NULL argument"); return MGS_ERR_INVALID_ARG; }
  const volatile uint64_t* hs = host_status;
  const uint64_t w0 = hs[0], w1 = hs[1];
  const bool a0 = status_arrived(w0, a->status_tag), a1 = status_arrived(w1, a->status_tag);
  if (a0 && num_rendered) *num_rendered = sat_i32((uint32_t)w0);
  if (a0 && ref_rendered) {  // (stored before word 0: arrived if word 0 has)
    const uint64_t w2 = hs[2];
    if (status_arrived(w2, a->status_tag)) *ref_rendered = sat_i32((uint32_t)w2);
  }
  if (a1 && chunks_used) *chunks_used = (int32_t)(uint32_t)w1;
  if (a0) {
    const uint32_t fl = (uint32_t)(w0 >> 32) & 0xffffu;
    if (fl & 2u) { set_error("%s", kHandshakeMsg); return MGS_RETRY_TABLE_INIT; }  // (nothing was binned; word 1 reports 0 chunks)
    const int rc = check_prefiltered(fl);
    if (rc) return rc;
    if (a->P > 0) {
      const BinShape bs = bin_shape(a, T, a->include_feature ? a->F : 0);
      if ((int64_t)(uint32_t)w0 > (int64_t)bs.cap) return MGS_NEED_CAPACITY;  // nothing was binned: word 1 reports 0 chunks
    }
  }
  if (!a0 || !a1) return MGS_PENDING;
  if ((uint32_t)(w1 >> 32) & 1u) return MGS_NEED_CAPACITY;  // the chunk pool overflowed
  return MGS_OK;
}
int mgs_forward_result(const MgsRasterArgs* a, const uint64_t* host_status, int32_t* num_rendered, int32_t* chunks_used,
                       int32_t* ref_rendered) {
  if (!a) { set_error("forward_result: NULL argument"); return MGS_ERR_INVALID_ARG; }
  return forward_result_T(a, num_tiles(a->W, a->H), host_status, num_rendered, chunks_used, ref_rendered);
}

int mgs_rasterize_backward(const MgsRasterArgs* a, int32_t R, const int32_t* radii, const float* dL_dout_color,
                           const float* dL_dout_feature, float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity,
                           float* dL_dcolors, float* dL_dfeature, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                           float* dL_dscales, float* dL_drotations, void* scratch, size_t scratch_bytes,
                           mgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(a);
  if (rc) return rc;
  if (a->P == 0) return MGS_OK;  // rasterize_points.cu:186: empty gradient tensors
  const int F = a->include_feature ? a->F : 0;
  if (!radii || !dL_dout_color || !dL_dmeans2D || !dL_dopacity || !dL_dcolors || !dL_dmeans3D || !dL_dcov3D ||
      !dL_dscales || !dL_drotations || (a->M > 0 && !dL_dsh) || (F > 0 && (!dL_dfeature || !dL_dout_feature))) {
    set_error("backward: a required pointer is NULL");
    return MGS_ERR_INVALID_ARG;
  }
  if (!scratch || scratch_bytes < mgs_backward_scratch_bytes(a->P, a->M, F) || !a->geom ||
      a->geom_bytes < mgs_geom_bytes(a->P, a->M, a->W, a->H) || !a->img || a->img_bytes < mgs_img_bytes(a->W, a->H) ||
      !a->binning) {
    set_error("backward: workspace too small");
    return MGS_ERR_WORKSPACE;
  }
  if (F > 0 && (reinterpret_cast<uintptr_t>(a->language_feature) & 15u)) {  // the render backward reads feature rows as float4
    set_error("backward: language_feature must be 16-byte aligned");
    return MGS_ERR_INVALID_ARG;
  }
  const Options o = options_of(a);
  const int T = num_tiles(a->W, a->H);
  const BinShape bs = bin_shape(a, T, F);
  // (R is the integer the forward handed back -- the reference's 3-sigma-rect count, at least the instances binned -- or -1:
  //  only R == 0 is acted on here; that the workspace holds the binned lists was the forward's check)
  if (bs.cap < 0) { set_error("backward: binning workspace smaller than its fixed part"); return MGS_ERR_WORKSPACE; }
  GeomView g = carve_geom(a->geom, a->P, a->M, T, 1, nullptr);
  ImgView im = carve_img(a->img, a->W, a->H, nullptr);
  ChunkView cv;
  BinView b = carve_binning(a->binning, bs.cap, T, F, bs.pool, &cv, nullptr);
  BwdScratch sc = carve_bwd(scratch, a->P, a->M, F, nullptr);
  const size_t P = (size_t)a->P;
  // accumulators the render backward adds into
  // dL_dcolors is the gradient w.r.t. the per-Gaussian RGB whether it came from colors_precomp or from SH
  // (the reference returns it in both cases, rasterize_points.cu:169,224)
  float* dcol = dL_dcolors;
  if (!a->accum_prezeroed)
  { StageTimer t(ST_BWD_MEMSET, stream);
    // one fill when the caller laid acc8 | dL_dcolors | dL_dfeature out back to back (manigaussian_amd/_C.py does)
    char* z0 = reinterpret_cast<char*>(sc.acc8);
    char* z_end = z0 + 8 * P * sizeof(float);
    const size_t scratch_total = mgs_backward_scratch_bytes(a->P, a->M, F);
    const bool adj_col = reinterpret_cast<char*>(dcol) >= z_end && reinterpret_cast<char*>(dcol) <= z0 + scratch_total + 64;
    const bool adj_feat = F == 0 || reinterpret_cast<char*>(dL_dfeature) == reinterpret_cast<char*>(dcol) + 3 * P * sizeof(float);
    if (adj_col && adj_feat) {
      char* end = reinterpret_cast<char*>(dcol) + (3 + (size_t)F) * P * sizeof(float);
      MGS_HIP(launch_zero_bytes(z0, (size_t)(end - z0), stream), "memset accumulators");
    } else {
      MGS_HIP(launch_zero_bytes(sc.acc8, 8 * P * sizeof(float), stream), "memset acc8");
      MGS_HIP(launch_zero_bytes(dcol, 3 * P * sizeof(float), stream), "memset dL_dcolors");
      if (F > 0) MGS_HIP(launch_zero_bytes(dL_dfeature, (size_t)F * P * sizeof(float), stream), "memset dL_dfeature");
    } }
  if (R != 0) {  // R < 0: count unknown to the host (asynchronous forward) -- empty ranges make the kernel a no-op
    const RenderArgs r = render_args(a, o, g);
    StageTimer t(ST_RENDER_BWD, stream);
    MGS_STAGE(launch_render_bwd_gm(r, b, im, cv, dL_dout_color, dL_dout_feature, sc.acc8, dcol, dL_dfeature, stream),
              "render backward", a->debug, stream);
  }
  BwdPreArgs p;
  p.V = 1; p.cov3D_per_view = 0; p.use_cam = 0;
  p.P = a->P; p.D = a->D; p.M = a->M; p.W = a->W; p.H = a->H;
  p.tanfovx = a->tanfovx; p.tanfovy = a->tanfovy;
  p.focal_y = a->H / (2.0f * a->tanfovy);
  p.focal_x = a->W / (2.0f * a->tanfovx);
  p.scale_modifier = a->scale_modifier;
  p.means3D = a->means3D; p.shs = a->shs; p.scales = a->scales; p.rotations = a->rotations;
  p.cov3D = a->cov3D_precomp ? a->cov3D_precomp : g.cov3D;
  p.viewmatrix = a->viewmatrix; p.projmatrix = a->projmatrix; p.campos = a->campos;
  p.radii = radii; p.clamped = g.clamped; p.acc8 = sc.acc8; p.dL_dcolor = dcol;
  p.dL_dmeans2D = dL_dmeans2D; p.dL_dconic = dL_dconic; p.dL_dopacity = dL_dopacity; p.dL_dmeans3D = dL_dmeans3D;
  p.dL_dcov3D = dL_dcov3D; p.dL_dsh = dL_dsh; p.dL_dscales = dL_dscales; p.dL_drot = dL_drotations;
  { StageTimer t(ST_PREPROCESS_BWD, stream);
    MGS_STAGE(launch_preprocess_bwd(p, stream), "preprocess backward", a->debug, stream); }
  return MGS_OK;
}


// ================================ multi-view batches ================================
// V views of ONE Gaussian set in one call.  The views are stacked into an atlas (each padded to whole tile rows), every
// (view, Gaussian) pair is a "virtual Gaussian" with id = view * P + Gaussian, and the binning and compositing kernels
// run unchanged on the atlas: V x the workgroups per launch, one set of launches per batch.  Per-Gaussian gradients
// are summed over the views on the device (atomics for colour/feature rows, registers in the backward preprocess).
namespace mgs {
struct Atlas { int V, tiles_x, tiles_yv, Hp, H, T; };
static Atlas atlas_of(int W, int H, int V) {
  Atlas at;
  at.V = V; at.tiles_x = (W + TILE - 1) / TILE; at.tiles_yv = (H + TILE - 1) / TILE;
  at.Hp = at.tiles_yv * TILE; at.H = V * at.Hp; at.T = at.tiles_x * at.tiles_yv * V;
  return at;
}
static int check_views(const MgsRasterArgs* a, int V, const MgsView* views, MgsRasterArgs* a1) {
  if (!a || !views || V < 1 || V > MAX_VIEWS) { set_error("views: need 1 <= V <= %d", MAX_VIEWS); return MGS_ERR_INVALID_ARG; }
  *a1 = *a;
  a1->viewmatrix = views[0].viewmatrix; a1->projmatrix = views[0].projmatrix; a1->campos = views[0].campos;
  a1->tanfovx = views[0].tanfovx; a1->tanfovy = views[0].tanfovy;
  int rc = check_common(a1);
  if (rc) return rc;
  for (int v = 0; v < V; v++)
    if (!views[v].viewmatrix || !views[v].projmatrix || !views[v].campos) { set_error("view %d: NULL matrix", v); return MGS_ERR_INVALID_ARG; }
  if (a->debug) { set_error("multi-view batches need debug 0"); return MGS_ERR_INVALID_ARG; }
  return MGS_OK;
}
static void fill_cams(ViewCam* cam, const MgsRasterArgs* a, int V, const MgsView* views) {
  for (int v = 0; v < V; v++) {
    cam[v].tanfovx = views[v].tanfovx; cam[v].tanfovy = views[v].tanfovy;
    cam[v].focal_y = a->H / (2.0f * views[v].tanfovy);
    cam[v].focal_x = a->W / (2.0f * views[v].tanfovx);
    cam[v].viewmatrix = views[v].viewmatrix; cam[v].projmatrix = views[v].projmatrix; cam[v].campos = views[v].campos;
  }
}
static RenderArgs views_render_args(const MgsRasterArgs* a, const Options& o, const Atlas& at, const GeomView& g) {
  RenderArgs r;
  const int F = a->include_feature ? a->F : 0;
  r.W = a->W; r.H = at.H; r.tiles_x = at.tiles_x; r.tiles_y = at.tiles_yv * at.V; r.F = F; r.include_feature = F > 0;
  r.fast_exp = o.fast_exp; r.exact_cull = o.exact_cull; r.gm_waves = o.gm_waves; r.dbg = 0;
  r.nwf = fwd_waves(F, r.tiles_x * r.tiles_y);
  r.V = at.V; r.Pg = a->P; r.Hv = a->H; r.Hp = at.Hp;
  r.colors_per_view = a->colors_precomp ? 0 : 1;
  r.bg = a->background;
  r.colors = a->colors_precomp ? a->colors_precomp : g.rgb;
  r.feats = a->language_feature;
  r.rec = g.rec;
  return r;
}
}  // namespace mgs

size_t mgs_views_geom_bytes(int P, int M, int W, int H, int V) {
  const Atlas at = atlas_of(W, H, V > 0 ? V : 1);
  size_t t; carve_geom(nullptr, P * at.V, M, at.T, at.V, &t); return t;
}
size_t mgs_views_img_bytes(int W, int H, int V) { const Atlas at = atlas_of(W, H, V > 0 ? V : 1); return mgs_img_bytes(W, at.H); }
size_t mgs_views_binning_bytes(int R, int W, int H, int F, int V) {
  return binning_bytes_T(R, 0, atlas_of(W, H, V > 0 ? V : 1).T, F);
}
size_t mgs_views_binning_bytes2(int R, int chunk_pool, int W, int H, int F, int V) {
  return binning_bytes_T(R, chunk_pool, atlas_of(W, H, V > 0 ? V : 1).T, F);
}
int mgs_views_chunk_pool_max(int R, int W, int H, int V) {
  return (int)chunk_pool_max(R > 0 ? (size_t)R : 1, atlas_of(W, H, V > 0 ? V : 1).T);
}
size_t mgs_views_backward_scratch_bytes(int P, int M, int F, int V) { return mgs_backward_scratch_bytes(P * (V > 0 ? V : 1), M, F); }

int mgs_rasterize_forward_views(const MgsRasterArgs* a, int32_t V, const MgsView* views, int32_t* radii, float* out_color,
                                float* out_feature, int32_t* num_rendered, uint64_t* host_status, mgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MgsRasterArgs a1;
  int rc = check_views(a, V, views, &a1);
  if (rc) return rc;
  if (!num_rendered || !host_status) { set_error("num_rendered / host_status is NULL"); return MGS_ERR_INVALID_ARG; }
  *num_rendered = 0;
  const size_t N = (size_t)a->W * a->H;
  const int F = a->include_feature ? a->F : 0;
  if (!out_color || (F > 0 && !out_feature)) { set_error("output image is NULL"); return MGS_ERR_INVALID_ARG; }
  if (a->P == 0) {
    MGS_HIP(launch_zero_bytes(out_color, (size_t)V * 3 * N * sizeof(float), stream), "memset out_color");
    if (F > 0) MGS_HIP(launch_zero_bytes(out_feature, (size_t)V * F * N * sizeof(float), stream), "memset out_feature");
    const uint64_t w = (uint64_t)(a->status_tag & 0xffffu) << 48;
    host_status[0] = w; host_status[1] = w; host_status[2] = w;
    return MGS_OK;
  }
  if (!radii || !a->opacities) { set_error("radii/opacities must be non-NULL"); return MGS_ERR_INVALID_ARG; }
  const Options o = options_of(a);
  const Atlas at = atlas_of(a->W, a->H, V);
  if (!a->geom || a->geom_bytes < mgs_views_geom_bytes(a->P, a->M, a->W, a->H, V) || !a->img ||
      a->img_bytes < mgs_views_img_bytes(a->W, a->H, V) || !a->binning) {
    set_error("views: geom/img/binning workspace missing or too small");
    return MGS_ERR_WORKSPACE;
  }
  const BinShape bs = bin_shape(a, at.T, F);
  if (bs.cap < 0) { set_error("binning workspace smaller than its fixed part"); return MGS_ERR_WORKSPACE; }
  GeomView g = carve_geom(a->geom, a->P * V, a->M, at.T, V, nullptr);
  ImgView im = carve_img(a->img, a->W, at.H, nullptr);
  g.flags = im.flags;
  ChunkView cv;
  BinView b = carve_binning(a->binning, bs.cap, at.T, F, bs.pool, &cv, nullptr);
  FwdPreArgs p;
  p.V = V; p.Pg = a->P; p.Hp = at.Hp; p.use_cam = 1;
  p.P = a->P * V; p.D = a->D; p.M = a->M; p.W = a->W; p.H = a->H;
  p.tiles_x = at.tiles_x; p.tiles_y = at.tiles_yv;
  p.tanfovx = p.tanfovy = p.focal_x = p.focal_y = 0.f;
  p.scale_modifier = a->scale_modifier;
  p.prefiltered = a->prefiltered; p.tight_bins = o.tight_bins;
  p.means3D = a->means3D; p.shs = a->shs; p.colors_precomp = a->colors_precomp; p.opacities = a->opacities;
  p.scales = a->scales; p.rotations = a->rotations; p.cov3D_precomp = a->cov3D_precomp;
  p.viewmatrix = p.projmatrix = p.campos = nullptr;
  fill_cams(p.cam, a, V, views);
  p.zero_ptr = nullptr; p.zero_f4 = 0;
  if (a->bwd_accum) {
    if ((reinterpret_cast<uintptr_t>(a->bwd_accum) & 15u) || (a->bwd_accum_bytes & 15u)) {
      set_error("bwd_accum must be 16-byte aligned and a multiple of 16 bytes");
      return MGS_ERR_INVALID_ARG;
    }
    p.zero_ptr = reinterpret_cast<float4*>(a->bwd_accum);
    p.zero_f4 = a->bwd_accum_bytes / 16;
  }
  const bool lds = lds_tables(o, at.T);
  im.direct_keys = lds ? direct_region(a, bs, at.T, F, (size_t)a->P * V, V, o) : nullptr;
  im.direct_stride = (uint32_t)a->P;
  p.direct_keys = im.direct_keys; p.direct_stride = im.direct_stride;
  const bool handshake = lds && !o.table_init;  // (see enqueue_preprocess)
  if (!handshake) MGS_HIP(launch_zero_bytes(im.flags, im.zero_bytes, stream), "zero flags + tile tables");
  p.tile_hist = im.tile_hist; p.blk_base = lds ? g.blk_base : nullptr; p.ref_count = im.ref_count;
  p.tables = im.flags; p.tables_words = (uint32_t)(im.zero_bytes / 4); p.ready = im.ready;
  p.nonce = handshake ? next_nonce() : 0ull;
  p.wg0_delay = (o.dbg & 1024) ? -1 : (o.dbg & 512) ? 100 : 0;
  im.nonce = p.nonce;
  { StageTimer t(ST_PREPROCESS, stream);
    MGS_HIP(launch_preprocess_fwd(p, g, radii, stream), "preprocess (views)"); }
  volatile uint64_t* hs = host_status;
  hs[0] = kStatusPending; hs[1] = kStatusPending; hs[2] = kStatusPending;
  const StatusSink status = {host_status, a->status_tag};
  for (int k = 0; k < 3; k++) {
    if (k == 2 && bucket_rank(o, at.T)) break;
    if (k == 0 && bucket_rank(o, at.T) && im.direct_keys) continue;  // (direct binning: no scatter launch)
    StageTimer t(ST_BIN_SCATTER + k, stream);
    MGS_HIP(launch_bin_segsort(k, lds, bucket_rank(o, at.T), g, b, im, a->P, V, bs.cap, at.tiles_x, at.tiles_yv * V, o.seg, o.dbg, status, stream),
            "binning (views)");
  }
  const RenderArgs r = views_render_args(a, o, at, g);
  { StageTimer t(ST_RENDER_FWD, stream);
    MGS_HIP(launch_render_fwd_dense(r, b, im, cv, out_color, out_feature, status, stream), "render forward (views)"); }
  if (a->async_forward) { *num_rendered = -1; return MGS_OK; }
  uint32_t R = 0, fl = 0, R_ref = 0;
  rc = wait_status(host_status, a->status_tag, stream, &R, &fl, &R_ref);  // (the preprocess only: see mgs_rasterize_forward)
  if (rc) return rc;
  rc = check_prefiltered(fl);
  if (rc) return rc;
  *num_rendered = sat_i32(R_ref);
  return (int)R > bs.cap ? MGS_NEED_CAPACITY : MGS_OK;
}

int mgs_forward_result_views(const MgsRasterArgs* a, int32_t V, const uint64_t* host_status, int32_t* num_rendered,
                             int32_t* chunks_used, int32_t* ref_rendered) {
  if (!a || V < 1) { set_error("forward_result_views: bad argument"); return MGS_ERR_INVALID_ARG; }
  return forward_result_T(a, atlas_of(a->W, a->H, V).T, host_status, num_rendered, chunks_used, ref_rendered);
}

int mgs_rasterize_backward_views(const MgsRasterArgs* a, int32_t V, const MgsView* views, int32_t R, const int32_t* radii,
                                 const float* dL_dout_color, const float* dL_dout_feature, float* dL_dmeans2D,
                                 float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_dfeature,
                                 float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations,
                                 void* scratch, size_t scratch_bytes, mgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  MgsRasterArgs a1;
  int rc = check_views(a, V, views, &a1);
  if (rc) return rc;
  if (a->P == 0) return MGS_OK;
  const int F = a->include_feature ? a->F : 0;
  if (!radii || !dL_dout_color || !dL_dmeans2D || !dL_dopacity || !dL_dcolors || !dL_dmeans3D || !dL_dcov3D ||
      !dL_dscales || !dL_drotations || (a->M > 0 && !dL_dsh) || (F > 0 && (!dL_dfeature || !dL_dout_feature))) {
    set_error("backward (views): a required pointer is NULL");
    return MGS_ERR_INVALID_ARG;
  }
  if (F > 0 && (reinterpret_cast<uintptr_t>(a->language_feature) & 15u)) {
    set_error("backward (views): language_feature must be 16-byte aligned");
    return MGS_ERR_INVALID_ARG;
  }
  const Options o = options_of(a);
  const Atlas at = atlas_of(a->W, a->H, V);
  if (!scratch || scratch_bytes < mgs_views_backward_scratch_bytes(a->P, a->M, F, V) || !a->geom ||
      a->geom_bytes < mgs_views_geom_bytes(a->P, a->M, a->W, a->H, V) || !a->img ||
      a->img_bytes < mgs_views_img_bytes(a->W, a->H, V) || !a->binning) {
    set_error("backward (views): workspace too small");
    return MGS_ERR_WORKSPACE;
  }
  const BinShape bs = bin_shape(a, at.T, F);
  if (bs.cap < 0) { set_error("backward (views): binning workspace smaller than its fixed part"); return MGS_ERR_WORKSPACE; }
  const size_t PV = (size_t)a->P * V, P = (size_t)a->P;
  GeomView g = carve_geom(a->geom, (int)PV, a->M, at.T, V, nullptr);
  ImgView im = carve_img(a->img, a->W, at.H, nullptr);
  ChunkView cv;
  BinView b = carve_binning(a->binning, bs.cap, at.T, F, bs.pool, &cv, nullptr);
  BwdScratch sc = carve_bwd(scratch, (int)PV, a->M, F, nullptr);
  const size_t ncol = a->colors_precomp ? P : PV;  // dL_dcolors rows: per Gaussian (precomputed colours) or per (view, Gaussian)
  if (!a->accum_prezeroed) {
    StageTimer t(ST_BWD_MEMSET, stream);
    MGS_HIP(launch_zero_bytes(sc.acc8, 8 * PV * sizeof(float), stream), "memset acc8");
    MGS_HIP(launch_zero_bytes(dL_dcolors, 3 * ncol * sizeof(float), stream), "memset dL_dcolors");
    if (F > 0) MGS_HIP(launch_zero_bytes(dL_dfeature, (size_t)F * P * sizeof(float), stream), "memset dL_dfeature");
  }
  if (R != 0) {
    const RenderArgs r = views_render_args(a, o, at, g);
    StageTimer t(ST_RENDER_BWD, stream);
    MGS_HIP(launch_render_bwd_gm(r, b, im, cv, dL_dout_color, dL_dout_feature, sc.acc8, dL_dcolors, dL_dfeature, stream),
            "render backward (views)");
  }
  BwdPreArgs p;
  p.V = V; p.cov3D_per_view = a->cov3D_precomp ? 0 : 1; p.use_cam = 1;
  p.P = a->P; p.D = a->D; p.M = a->M; p.W = a->W; p.H = a->H;
  p.tanfovx = p.tanfovy = p.focal_x = p.focal_y = 0.f;
  p.scale_modifier = a->scale_modifier;
  p.means3D = a->means3D; p.shs = a->shs; p.scales = a->scales; p.rotations = a->rotations;
  p.cov3D = a->cov3D_precomp ? a->cov3D_precomp : g.cov3D;
  p.viewmatrix = p.projmatrix = p.campos = nullptr;
  fill_cams(p.cam, a, V, views);
  p.radii = radii; p.clamped = g.clamped; p.acc8 = sc.acc8; p.dL_dcolor = dL_dcolors;
  p.dL_dmeans2D = dL_dmeans2D; p.dL_dconic = dL_dconic; p.dL_dopacity = dL_dopacity; p.dL_dmeans3D = dL_dmeans3D;
  p.dL_dcov3D = dL_dcov3D; p.dL_dsh = dL_dsh; p.dL_dscales = dL_dscales; p.dL_drot = dL_drotations;
  { StageTimer t(ST_PREPROCESS_BWD, stream);
    MGS_HIP(launch_preprocess_bwd(p, stream), "preprocess backward (views)"); }
  return MGS_OK;
}

int mgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     mgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P < 0) { set_error("P < 0"); return MGS_ERR_INVALID_ARG; }
  if (P == 0) return MGS_OK;
  if (!means3D || !viewmatrix || !projmatrix || !present) { set_error("mark_visible: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  MGS_HIP(launch_mark_visible(P, means3D, viewmatrix, projmatrix, present, stream), "mark_visible");
  return MGS_OK;
}

// Diagnostic (blocking): what the forward that last ran on these workspaces left behind, in the units its state is kept in:
//   incidences    (8x8 block, Gaussian) pairs of the chunks some pixel of the block visited (the fill may have listed more)
//   chunks        64-survivor chunks some pixel of their block visited
//   pixel_chunks  (pixel, chunk) pairs visited: what the per-chunk state (partial sums, T_end, T_mid, last_pos) costs
int mgs_forward_stats(const MgsRasterArgs* a, int32_t V, int64_t* incidences, int64_t* chunks, int64_t* pixel_chunks,
                      mgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a || !a->binning || V < 0 || V > MAX_VIEWS) { set_error("forward_stats: bad argument"); return MGS_ERR_INVALID_ARG; }
  const int F = a->include_feature ? a->F : 0;
  const int T = V > 0 ? atlas_of(a->W, a->H, V).T : num_tiles(a->W, a->H);
  const BinShape bs = bin_shape(a, T, F);
  if (bs.cap < 0) { set_error("forward_stats: binning workspace smaller than its fixed part"); return MGS_ERR_WORKSPACE; }
  ChunkView cv;
  (void)carve_binning(a->binning, bs.cap, T, F, bs.pool, &cv, nullptr);
  std::vector<uint2> ns((size_t)T * 4);
  std::vector<uint32_t> lc((size_t)T * 4 * 64);
  MGS_HIP(hipMemcpyAsync(ns.data(), cv.nsurv, ns.size() * sizeof(uint2), hipMemcpyDeviceToHost, stream), "forward_stats copy");
  MGS_HIP(hipMemcpyAsync(lc.data(), cv.last_chunk, lc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "forward_stats copy");
  MGS_HIP(hipStreamSynchronize(stream), "forward_stats sync");
  int64_t inc = 0, ch = 0, pc = 0;
  for (size_t b = 0; b < ns.size(); b++) {
    uint32_t vmax = 0;
    for (int p = 0; p < 64; p++) { const uint32_t v = lc[b * 64 + p]; pc += v; if (v > vmax) vmax = v; }
    ch += vmax;
    const int64_t listed = ns[b].x, visited = (int64_t)vmax * CHUNK;
    inc += listed < visited ? listed : visited;
  }
  if (incidences) *incidences = inc;
  if (chunks) *chunks = ch;
  if (pixel_chunks) *pixel_chunks = pc;
  return MGS_OK;
}

// Diagnostic: byte offsets of the per-Gaussian arrays the forward preprocess leaves in a geom workspace of mgs_geom_bytes(P,
// M, W, H) bytes (tests compare them bit for bit with the reference's GeometryState).
int mgs_debug_geom_layout(int P, int M, int W, int H, size_t* depths, size_t* rec, size_t* rgb, size_t* cov3D) {
  if (P < 0 || W <= 0 || H <= 0) { set_error("geom_layout: bad shape"); return MGS_ERR_INVALID_ARG; }
  char* const base = reinterpret_cast<char*>(ALIGN);  // (a non-null dummy base: carve_geom yields null pointers for nullptr)
  const GeomView g = carve_geom(base, P, M, num_tiles(W, H), 1, nullptr);
  if (depths) *depths = (size_t)(reinterpret_cast<char*>(g.depths) - base);
  if (rec) *rec = (size_t)(reinterpret_cast<char*>(g.rec) - base);
  if (rgb) *rgb = (size_t)(reinterpret_cast<char*>(g.rgb) - base);
  if (cov3D) *cov3D = (size_t)(reinterpret_cast<char*>(g.cov3D) - base);
  return MGS_OK;
}

int mgs_debug_binning_layout(const MgsRasterArgs* a, int32_t V, size_t* keys_unsorted, size_t* point_list, size_t* img_ranges,
                             int32_t* capacity) {
  if (!a || a->W <= 0 || a->H <= 0) { set_error("binning_layout: bad argument"); return MGS_ERR_INVALID_ARG; }
  const int F = a->include_feature ? a->F : 0;
  const int T = V > 0 ? atlas_of(a->W, a->H, V).T : num_tiles(a->W, a->H);
  const BinShape bs = bin_shape(a, T, F);
  if (bs.cap < 0) { set_error("binning_layout: binning workspace smaller than its fixed part"); return MGS_ERR_WORKSPACE; }
  char* const base = reinterpret_cast<char*>(ALIGN);
  const BinView b = carve_binning(base, bs.cap, T, F, bs.pool, nullptr, nullptr);
  const ImgView im = carve_img(base, a->W, V > 0 ? atlas_of(a->W, a->H, V).H : a->H, nullptr);
  if (keys_unsorted) *keys_unsorted = (size_t)(reinterpret_cast<char*>(b.keys_unsorted) - base);
  if (point_list) *point_list = (size_t)(reinterpret_cast<char*>(b.point_list) - base);
  if (img_ranges) *img_ranges = (size_t)(reinterpret_cast<char*>(im.ranges) - base);
  if (capacity) *capacity = bs.cap;
  return MGS_OK;
}

int mgs_debug_direct_keys(const MgsRasterArgs* a, int32_t V, size_t* keys, int32_t* stride) {
  if (!a || a->W <= 0 || a->H <= 0 || !keys || !stride) { set_error("direct_keys: bad argument"); return MGS_ERR_INVALID_ARG; }
  *keys = 0; *stride = 0;
  if (a->P <= 0 || !a->binning) return MGS_OK;
  const int F = a->include_feature ? a->F : 0;
  const int v = V > 0 ? V : 1;
  const int T = V > 0 ? atlas_of(a->W, a->H, V).T : num_tiles(a->W, a->H);
  const BinShape bs = bin_shape(a, T, F);
  if (bs.cap < 0) { set_error("direct_keys: binning workspace smaller than its fixed part"); return MGS_ERR_WORKSPACE; }
  const uint64_t* dk = direct_region(a, bs, T, F, (size_t)a->P * v, v, options_of(a));
  if (!dk) return MGS_OK;
  *keys = (size_t)(reinterpret_cast<const char*>(dk) - static_cast<const char*>(a->binning));
  *stride = a->P;
  return MGS_OK;
}

int mgs_profile_num_stages(void) { return ST_COUNT; }
const char* mgs_profile_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }

int mgs_profile_read(double* total_ms, int32_t* counts, int reset) {
  Profiler& p = profiler();
  std::lock_guard<std::mutex> lk(p.mu);
  for (int st = 0; st < ST_COUNT; st++) {
    double sum = 0;
    int n = 0;
    for (auto& pr : p.used[st]) {
      float ms = 0.f;
      if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        sum += ms;
        n++;
      }
    }
    if (total_ms) total_ms[st] = sum;
    if (counts) counts[st] = n;
    if (reset) {
      for (auto& pr : p.used[st]) { p.pool.push_back(pr.first); p.pool.push_back(pr.second); }
      p.used[st].clear();
    }
  }
  return MGS_OK;
}

int mgs_selftest(mgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int* d = nullptr;
  MGS_HIP(hipMalloc(&d, sizeof(int)), "hipMalloc");
  int h = 0;
  hipError_t e = launch_zero_bytes(d, sizeof(int), stream);
  if (e == hipSuccess) e = launch_selftest(d, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(d);
  MGS_HIP(e, "selftest");
  if (h != 0) { set_error("wave64 primitive self-test failed, mask 0x%x", h); return h; }
  return MGS_OK;
}

int mgs_calibration_kernel(int iters, float* sink, mgs_stream_t stream_) {
  if (iters < 1 || !sink) { set_error("calibration: iters >= 1 and a sink of 256 * 1024 floats"); return MGS_ERR_INVALID_ARG; }
  MGS_HIP(launch_calibration(iters, sink, (hipStream_t)stream_), "calibration kernel");
  return MGS_OK;
}

}  // extern "C"
