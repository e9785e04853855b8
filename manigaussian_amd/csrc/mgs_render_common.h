// mgs_render_common.h -- helpers shared by the render kernels (per-subtile and chunk-parallel variants).
#pragma once
#include "mgs_common.h"
#include "mgs_device.h"

namespace mgs {

// exp() of the render kernels.  FAST: v_exp_f32 of x log2(e) (two instructions; relative error ~2e-7 |x|).  Otherwise ocml's
// expf AS THIS TOOLCHAIN EVALUATES IT -- the extended-precision reduction x log2(e) = e + a (|a| <= 1/2), exp2(a), ldexp --
// without its two range clamps (x < -103.3 -> 0, x > 88.7 -> inf): nine instructions instead of fourteen, bit-identical to
// expf(x) for -103 < x < 88 (mgs_selftest compares them), and outside that range the callers discard the value anyway (a
// power > 0 is skipped, forward.cu:349; at x < -103 alpha = opacity * 1e-45 is far below the 1/255 skip threshold either way).
__device__ __forceinline__ float exp_ocml_unclamped(float x) {
#pragma clang fp contract(off)
  const float ph = x * 0x1.715476p+0f;                            // log2(e) rounded to float
  float pl = __builtin_fmaf(x, 0x1.715476p+0f, -ph);              // the product's rounding error
  const float e = __builtin_rintf(ph);
  pl = __builtin_fmaf(x, 0x1.4ae0bep-26f, pl);                    // + x * (log2(e) - float(log2(e)))
  const float a = (ph - e) + pl;
  return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}
// The same for two arguments at once: the multiply-adds as packed fp32 operations (v_pk_mul / v_pk_fma / v_pk_add: IEEE, the
// same roundings as the scalar forms); rint, exp2 and ldexp per component.  mgs_selftest compares it with expf bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 exp_ocml_unclamped2(f32x2 x) {
#pragma clang fp contract(off)
  const f32x2 c = {0x1.715476p+0f, 0x1.715476p+0f}, c2 = {0x1.4ae0bep-26f, 0x1.4ae0bep-26f};
  const f32x2 ph = x * c;
  f32x2 pl = __builtin_elementwise_fma(x, c, -ph);
  const f32x2 e = {__builtin_rintf(ph.x), __builtin_rintf(ph.y)};
  pl = __builtin_elementwise_fma(x, c2, pl);
  const f32x2 a = (ph - e) + pl;
  return f32x2{__builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(a.x), (int)e.x),
               __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(a.y), (int)e.y)};
}
// The exponent of one (pixel, Gaussian) pair, -1/2 (cx dx^2 + cz dy^2) - cy dx dy (forward.cu:345-347, backward.cu:456-458).
// EVERY kernel that takes the pair's two decisions (power > 0 -> skip, alpha < 1/255 -> skip) evaluates it through these
// functions: the forward and every form of the backward must round it identically, or a pair within an ulp of a threshold is
// blended by one and skipped by the other (the transmittances and suffix sums of the backward are then those of another
// blend).  The roundings are THE REFERENCE KERNELS' as this toolchain compiles them (round 6; the test-only hipcc build of its sources, read off
// the ISA of renderCUDA forward and backward at feature widths 3, 8 and 32 -- the same everywhere): the two squares are packed
// by the vectoriser and therefore NOT contracted,
//     t = ((dx cx) dx) + ((dy cz) dy)          v_pk_mul, v_pk_mul, v_add      (five roundings)
//     power = fma(t, -1/2, -((dx cy) dy))      v_mul, v_mul, v_fma
// Rounds 3-5 had t as fma(cx dx, dx, (cz dy) dy): up to a few ulps away, which moved an alpha across 1/255 in one pair of one
// scene in 1 200 (tests/tools/fuzz_parity.py seed 101 case 234: 2e-3 on one pixel) -- with the reference's own sequence alpha is
// the reference's bit for bit (bit-identical conic, mean and opacity: tests/test_gpu_parity.py; bit-identical exp: above).
// Written with explicit operations and contraction off: left to the compiler, the scalar and the packed expression are fused
// differently.  mgs_selftest compares the forms bit for bit, and the pair of that scene against the reference's bits.
__device__ __forceinline__ float gauss_power(float cx, float cy, float cz, float dx, float dy) {
#pragma clang fp contract(off)
  const float t = (cx * dx) * dx + (cz * dy) * dy;
  return __builtin_fmaf(-0.5f, t, -((cy * dx) * dy));
}
__device__ __forceinline__ f32x2 gauss_power2(float cx, float cy, float cz, f32x2 dx, float dy) {
#pragma clang fp contract(off)
  const float yy = (cz * dy) * dy;
  const f32x2 t = (cx * dx) * dx + f32x2{yy, yy};
  return __builtin_elementwise_fma(f32x2{-0.5f, -0.5f}, t, -((cy * dx) * dy));
}
// ... and with two Gaussians (conics) as well as two offsets: the forward's double steps
__device__ __forceinline__ f32x2 gauss_power2v(f32x2 cx, f32x2 cy, f32x2 cz, f32x2 dx, f32x2 dy) {
#pragma clang fp contract(off)
  const f32x2 t = (cx * dx) * dx + (cz * dy) * dy;
  return __builtin_elementwise_fma(f32x2{-0.5f, -0.5f}, t, -((cy * dx) * dy));
}
template <bool FAST>
__device__ __forceinline__ f32x2 exp2_(f32x2 x) {
  if constexpr (FAST) return f32x2{__expf(x.x), __expf(x.y)};
  else return exp_ocml_unclamped2(x);
}
template <bool FAST>
__device__ __forceinline__ float exp_(float x) {
  if constexpr (FAST) return __expf(x);
  else return exp_ocml_unclamped(x);
}

// block index -> (tile, sub-block).  Blocks b, b+8, b+16, b+24 (same XCD under the observed
// round-robin dispatch) work on the same tile, so the tile's instance list is fetched into one L2.
__device__ __forceinline__ void map_block(int b, int& tile, int& sub) {
  tile = (b / 32) * 8 + (b % 8);
  sub = (b / 8) % 4;
}

// conservative lane-parallel cull of one instance record against a pixel block [bxmin,bxmax]x[bymin,bymax]
__device__ __forceinline__ bool overlaps_block(const float4& g0, const float4& g1, float bxmin, float bxmax,
                                               float bymin, float bymax) {
  return g1.z >= 0.f && (g0.x + g1.z >= bxmin) && (g0.x - g1.z <= bxmax) && (g0.y + g1.w >= bymin) &&
         (g0.y - g1.w <= bymax);
}

// conservative EXACT-shape cull: does {q(d) <= tau} (tau = 2 ln(255 o), inflated) reach the pixel rectangle?
// bbox test first; then the minimum of the quadratic form over the rectangle (on the boundary unless the
// centre is inside).  Degenerate conics (marked by hx >= 1e5 in the preprocess) are never culled.
__device__ __forceinline__ bool reaches_block(const float4& g0, const float4& g1, float bxmin, float bxmax,
                                              float bymin, float bymax) {
  if (!overlaps_block(g0, g1, bxmin, bxmax, bymin, bymax)) return false;
  if (g1.z >= 1e5f) return true;
  const float dx0 = bxmin - g0.x, dx1 = bxmax - g0.x, dy0 = bymin - g0.y, dy1 = bymax - g0.y;
  if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;
  const float cx = g0.z, cy = g0.w, cz = g1.x;
  const float tau = 2.0f * __logf(fmaxf(g1.y * 255.0f, 1.0f)) * 1.001f + 2e-3f;
  const float icz = 1.0f / cz, icx = 1.0f / cx;
  float best = 3.0e38f;
  {
    const float dy = fminf(fmaxf(-cy * dx0 * icz, dy0), dy1);
    best = fminf(best, cx * dx0 * dx0 + 2.f * cy * dx0 * dy + cz * dy * dy);
  }
  {
    const float dy = fminf(fmaxf(-cy * dx1 * icz, dy0), dy1);
    best = fminf(best, cx * dx1 * dx1 + 2.f * cy * dx1 * dy + cz * dy * dy);
  }
  {
    const float dx = fminf(fmaxf(-cy * dy0 * icx, dx0), dx1);
    best = fminf(best, cx * dx * dx + 2.f * cy * dx * dy0 + cz * dy0 * dy0);
  }
  {
    const float dx = fminf(fmaxf(-cy * dy1 * icx, dx0), dx1);
    best = fminf(best, cx * dx * dx + 2.f * cy * dx * dy1 + cz * dy1 * dy1);
  }
  return !(best > tau);
}

struct PixBlk {
  int px, py;      // pixel of this lane in ATLAS coordinates (== image coordinates for a single view)
  int v;           // view of the block
  bool inside;
  float pxf, pyf, bxmin, bxmax, bymin, bymax;  // VIEW-LOCAL pixel / block coordinates (what the records are in): a
                                               // batched view computes bit for bit what a single-view call computes
  size_t pixl;     // y_local * W + x inside the view's image plane
  size_t pixa;     // atlas pixel index py * W + px (per-pixel workspace state)
};
__device__ __forceinline__ PixBlk pix_blk(const RenderArgs& r, int tile, int sub, int lane) {
  PixBlk p;
  const int tx = tile % r.tiles_x, ty = tile / r.tiles_x;
  const int bx0 = tx * TILE + (sub & 1) * SUB, by0 = ty * TILE + (sub >> 1) * SUB;
  p.px = bx0 + (lane & 7);
  p.py = by0 + (lane >> 3);
  int yl = p.py;
  p.v = 0;
  if (r.V > 1) { p.v = p.py / r.Hp; yl = p.py - p.v * r.Hp; }
  p.inside = p.px < r.W && yl < r.Hv;
  const int by0l = by0 - p.v * r.Hp;
  p.pxf = (float)p.px; p.pyf = (float)yl;
  p.bxmin = (float)bx0; p.bxmax = (float)min(bx0 + SUB - 1, r.W - 1);
  p.bymin = (float)by0l; p.bymax = (float)min(by0l + SUB - 1, r.Hv - 1);
  p.pixl = (size_t)yl * r.W + p.px;
  p.pixa = (size_t)p.py * r.W + p.px;
  return p;
}
// Gaussian of a (possibly virtual) instance id
__device__ __forceinline__ uint32_t gauss_of(const RenderArgs& r, uint32_t id) {
  return r.V > 1 ? id % (uint32_t)r.Pg : id;
}

template <bool EXACT>
__device__ __forceinline__ bool cull_ok(const float4& g0, const float4& g1, const PixBlk& p) {
  if constexpr (EXACT) return reaches_block(g0, g1, p.bxmin, p.bxmax, p.bymin, p.bymax);
  else return overlaps_block(g0, g1, p.bxmin, p.bxmax, p.bymin, p.bymax);
}

// wave-private LDS hand-off (all LDS traffic of one wave executes in order; only the compiler must not reorder)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Chunk records live in a pool (ChunkView): round r of block (tile, sub) owns the <= NWF consecutive records starting at
// round_base[round_entry(...)], NWF = waves of the render forward = chunks per round.  A block's rounds number at most
// len/512 + 1 (len = its tile's list length, a round consumes >= 512 survivors unless it is the last), so the entries
// range_x/512 + 2*tile + r are collision-free across tiles and bounded by R/512 + 2T.
constexpr uint32_t ROUND_GRANULE = 512;
__device__ __forceinline__ size_t round_entry(uint32_t range_x, int tile, int sub, uint32_t r) {
  return ((size_t)(range_x / ROUND_GRANULE) + 2 * (size_t)tile + r) * 4 + (size_t)sub;
}
// Feature widths compiled in.  Other widths are padded up by the host shim (zero channels change nothing).
#define MGS_FOR_EACH_F(X) X(0) X(3) X(4) X(8) X(16) X(32) X(64)

}  // namespace mgs
