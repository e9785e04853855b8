// mgs_preprocess.hip -- per-Gaussian kernels: forward preprocess (K2), fused backward preprocess
// (K9 + K10 in one pass), frustum marking (K1).  One thread per Gaussian, streaming, HBM-bound.
//
// What is computed follows the reference (RAST = third_party/gaussian-splatting/submodules/
// diff-gaussian-rasterization): forward.cu:156-257 (preprocessCUDA), :75-114 (computeCov2D),
// :119-153 (computeCov3D), :21-72 (SH -> RGB); backward.cu:144-274 (computeCov2DCUDA),
// :346-396 (preprocessCUDA bwd), :20-139 (SH bwd), :278-341 (cov3D bwd).  How it is computed is
// not: plain scalar algebra on registers (glm is not used), K9 and K10 fused so every per-Gaussian
// array is touched once, outputs written unconditionally so no pre-zeroing pass is needed.
#include "mgs_common.h"
#include <algorithm>
#include "mgs_device.h"

namespace mgs {

__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                        -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                        0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                        -0.5900435899266435f};

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const float* p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// The projection of the mean: shared by forward and backward (recompute instead of storing).
struct ViewCov {
  float tx, ty, tz;     // clamped view-space mean (forward.cu:83-88)
  float txtz, tytz;     // unclamped ratios (backward.cu:173-176)
  float a[3], b[3];     // rows of T = W*J that matter: T[0][r], T[1][r] in glm indexing
};

// computeCov2D (forward.cu:75-114).  The ROUNDING of every multiply-add below is the reference kernels' (hipcc, this
// toolchain), found by matching their conics bit for bit over 363 000 Gaussians seen through four cameras without structural
// zeros, both focal signs (scripts/diag/cov2d_ascent.py: a coordinate ascent over the fusion pattern of every sum; it
// converges to ONE pattern from any start): contraction is off inside these functions, fused operations are explicit.
__device__ __forceinline__ ViewCov view_cov(V3 mean, const float* __restrict__ vm, float fx, float fy,
                                            float tanx, float tany) {
#pragma clang fp contract(off)
  ViewCov o;
  // t = transformPoint4x3(mean, viewmatrix) as it rounds INSIDE computeCov2D: ((m1 y fused + m0 x) + m2 z) + m3 for all
  // three rows (p_view.z of in_frustum, the sort key, fuses m0 x instead: row_view_z)
  float tx = ((__builtin_fmaf(vm[4], mean.y, vm[0] * mean.x)) + vm[8] * mean.z) + vm[12];
  float ty = ((__builtin_fmaf(vm[5], mean.y, vm[1] * mean.x)) + vm[9] * mean.z) + vm[13];
  float tz = ((__builtin_fmaf(vm[6], mean.y, vm[2] * mean.x)) + vm[10] * mean.z) + vm[14];
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  o.txtz = tx / tz;
  o.tytz = ty / tz;
  tx = fminf(limx, fmaxf(-limx, o.txtz)) * tz;
  ty = fminf(limy, fmaxf(-limy, o.tytz)) * tz;
  o.tx = tx; o.ty = ty; o.tz = tz;
  const float j00 = fx / tz, j02 = -(fx * tx) / (tz * tz);
  const float j11 = fy / tz, j12 = -(fy * ty) / (tz * tz);
  // T = W * J (forward.cu:90-100): J's zeros drop out; rows 0 and 1 of a fuse the j02 product onto the rounded j00 product,
  // row 2 adds two rounded products; every row of b fuses the j12 product onto the rounded j11 product
  o.a[0] = __builtin_fmaf(vm[2], j02, vm[0] * j00);
  o.a[1] = __builtin_fmaf(vm[6], j02, vm[4] * j00);
  o.a[2] = vm[8] * j00 + vm[10] * j02;
#pragma unroll
  for (int r = 0; r < 3; r++) o.b[r] = __builtin_fmaf(vm[4 * r + 2], j12, vm[4 * r + 1] * j11);
  return o;
}

// cov2D entries (before the +0.3 low-pass) and V*a, V*b for the backward: cov = (T^T Vrk^T) T (forward.cu:107); cov[0][1],
// the entry the reference returns as cov.y, is (V b) . a.  Sums of three products p0 + p1 + p2: `mid` rounds p1 first
// (fma(p2, fma(p0, round(p1)))), `lead` rounds p0 first.
__device__ __forceinline__ void cov2d_from(const ViewCov& vc, const float* c6, float& c00, float& c01,
                                           float& c11, float* Va, float* Vb) {
#pragma clang fp contract(off)
  const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  auto mid = [](float p0, float q0, float p1, float q1, float p2, float q2) {
    return __builtin_fmaf(p2, q2, __builtin_fmaf(p0, q0, p1 * q1));
  };
  auto lead = [](float p0, float q0, float p1, float q1, float p2, float q2) {
    return __builtin_fmaf(p2, q2, __builtin_fmaf(p1, q1, p0 * q0));
  };
#pragma unroll
  for (int i = 0; i < 3; i++) Va[i] = mid(vc.a[0], V[0][i], vc.a[1], V[1][i], vc.a[2], V[2][i]);
  Vb[0] = lead(vc.b[0], V[0][0], vc.b[1], V[1][0], vc.b[2], V[2][0]);
  Vb[1] = lead(vc.b[0], V[0][1], vc.b[1], V[1][1], vc.b[2], V[2][1]);
  Vb[2] = mid(vc.b[0], V[0][2], vc.b[1], V[1][2], vc.b[2], V[2][2]);
  c00 = mid(Va[0], vc.a[0], Va[1], vc.a[1], Va[2], vc.a[2]);
  c01 = mid(Vb[0], vc.a[0], Vb[1], vc.a[1], Vb[2], vc.a[2]);
  c11 = mid(Vb[0], vc.b[0], Vb[1], vc.b[1], Vb[2], vc.b[2]);
}
// det of the low-passed 2-D covariance as the reference's build rounds it: c01^2 fused onto the rounded c00 c11
__device__ __forceinline__ float cov2d_det(float c00, float c01, float c11) {
#pragma clang fp contract(off)
  return __builtin_fmaf(-c01, c01, c00 * c11);
}

// Rstd[i][k]: rows as written in forward.cu:135-139 (glm column i), quaternion (r,x,y,z) NOT normalised.
__device__ __forceinline__ void quat_rows(const float* q, float R[3][3]) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// computeCov3D (forward.cu:119-153): Sigma = (S R)^T (S R) with glm's column-major products, the quaternion (r, x, y, z) NOT
// normalised.  What is computed is the reference's; HOW IT ROUNDS is pinned as well: float products do not associate and
// a compiler is free to fuse any multiply into a following add (fp-contract), so the same source text rounds one way inside
// the reference's glm templates and another way in scalar code -- round 4 measured cov3D a few ulps apart on 90 % of the
// Gaussians, and conics up to 85 ulps apart behind it (scripts/diag/preprocess_bits.py).  Below, every multiply-add is
// written out the way the reference's kernels evaluate it when built with this toolchain (found by matching their cov3D
// bit for bit over 92 000 Gaussians, scripts/diag/cov3d_search.py): contraction is switched off for the block and the fused
// operations are explicit fmaf calls.
__device__ __forceinline__ void cov3d_from_scale_rotation(const float* __restrict__ q, V3 s, float mod, float* c6) {
#pragma clang fp contract(off)
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  const float yy = y * y, zz = z * z, rz = r * z, ry = r * y, rx = r * x;
  // R[i][k] = glm column i, component k (forward.cu:135-139)
  float R[3][3];
  R[0][0] = 1.f - 2.f * (yy + zz);
  R[0][1] = 2.f * __builtin_fmaf(x, y, -rz);
  R[0][2] = 2.f * __builtin_fmaf(x, z, ry);
  R[1][0] = 2.f * __builtin_fmaf(r, z, x * y);
  R[1][1] = 1.f - 2.f * __builtin_fmaf(x, x, zz);
  R[1][2] = 2.f * __builtin_fmaf(y, z, -rx);
  R[2][0] = 2.f * __builtin_fmaf(x, z, -ry);
  R[2][1] = 2.f * __builtin_fmaf(y, z, rx);
  R[2][2] = 1.f - 2.f * __builtin_fmaf(x, x, yy);
  const float sc[3] = {mod * s.x, mod * s.y, mod * s.z};
  float m[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int k = 0; k < 3; k++) m[i][k] = sc[k] * R[i][k];
  // sum of three products p0 + p1 + p2, p_k = m[a][k] m[b][k]: which product is rounded first
  auto first1 = [&](int a_, int b_) {  // ((p1) + p0 fused) + p2 fused
    return __builtin_fmaf(m[a_][2], m[b_][2], __builtin_fmaf(m[a_][0], m[b_][0], m[a_][1] * m[b_][1]));
  };
  auto first0 = [&](int a_, int b_) {  // ((p0) + p1 fused) + p2 fused
    return __builtin_fmaf(m[a_][2], m[b_][2], __builtin_fmaf(m[a_][1], m[b_][1], m[a_][0] * m[b_][0]));
  };
  c6[0] = first1(0, 0);
  c6[1] = first0(0, 1);
  c6[2] = first1(0, 2);
  c6[3] = first1(1, 1);
  c6[4] = first1(1, 2);
  c6[5] = first0(2, 2);
}

// One row of a 4x4 transform applied to a point, m0 x + m1 y + m2 z + m3, with the rounding of the reference's build pinned
// the same way (auxiliary.h:58-77 transformPoint4x3 / 4x4: the same source text, yet the compiler fuses another multiply at
// every use site; scripts/diag/proj_search.py matched the pixel means bit for bit on a camera without structural zeros):
//   row_view_z  p_view.z of in_frustum = depths[]          ((m0 x fused + m1 y) + m2 z) + m3
//   row_hom_w   p_hom.w of the projection                  ((m1 y fused + m0 x) + m2 z) + m3
//   row_hom_xy  p_hom.x, p_hom.y of the projection         (m2 z fused + (m0 x fused + m1 y)) + m3
__device__ __forceinline__ float row_view_z(float m0, float m1, float m2, float m3, V3 p) {
#pragma clang fp contract(off)
  return ((__builtin_fmaf(m0, p.x, m1 * p.y)) + m2 * p.z) + m3;
}
__device__ __forceinline__ float row_hom_w(float m0, float m1, float m2, float m3, V3 p) {
#pragma clang fp contract(off)
  return ((__builtin_fmaf(m1, p.y, m0 * p.x)) + m2 * p.z) + m3;
}
__device__ __forceinline__ float row_hom_xy(float m0, float m1, float m2, float m3, V3 p) {
#pragma clang fp contract(off)
  return __builtin_fmaf(m2, p.z, __builtin_fmaf(m0, p.x, m1 * p.y)) + m3;
}

// auxiliary.h:41-44 is written with double literals: evaluate in double, round once.
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

__device__ __forceinline__ void get_rect(float px, float py, int rad, int gx, int gy, int& x0, int& y0, int& x1,
                                         int& y1) {
  // auxiliary.h:46-56: C-cast truncation toward zero, then clamp to the grid
  x0 = min(gx, max(0, (int)((px - rad) / TILE)));
  y0 = min(gy, max(0, (int)((py - rad) / TILE)));
  x1 = min(gx, max(0, (int)((px + rad + TILE - 1) / TILE)));
  y1 = min(gy, max(0, (int)((py + rad + TILE - 1) / TILE)));
}

// Number of tiles of the reference rect [x0,x1)x[y0,y1) that the alpha>=1/255 footprint bbox
// (centre p, half extents h; h.x < 0 => none) reaches.  Tiles are TILE px wide, pixel centres are
// integers, tile t covers pixels [t*TILE, t*TILE + TILE-1].
__device__ __forceinline__ void tight_rect(float px, float py, float hx, float hy, int& x0, int& y0, int& x1,
                                           int& y1) {
  if (hx < 0.f) { x1 = x0; y1 = y0; return; }
  // tile t is reached iff px+hx >= t*TILE and px-hx <= t*TILE+TILE-1
  const int tx0 = (int)ceilf((px - hx - (TILE - 1)) / TILE), tx1 = (int)floorf((px + hx) / TILE) + 1;
  const int ty0 = (int)ceilf((py - hy - (TILE - 1)) / TILE), ty1 = (int)floorf((py + hy) / TILE) + 1;
  x0 = max(x0, tx0); x1 = max(x0, min(x1, tx1));
  y0 = max(y0, ty0); y1 = max(y0, min(y1, ty1));
}

// HIST: build the per-tile instance histogram through an LDS histogram per workgroup, one RETURNING global atomic per
// non-empty (workgroup, tile) -- the workgroup's reservation inside the tile's slice, which the bin scatter reads back.
// !HIST (tile counts beyond the LDS tables): 256-thread workgroups, one plain atomic per instance.
template <bool HIST>
__global__ void __launch_bounds__(PRE_BLOCK) preprocess_fwd_kernel(FwdPreArgs a, GeomView g,
                                                                   int32_t* __restrict__ radii) {
  extern __shared__ uint32_t s_hist[];  // [tiles] when HIST: instances per tile
  __shared__ uint32_t s_total, s_total_ref, s_pref, s_fail;
  if (HIST && threadIdx.x == 0) { s_total = 0; s_total_ref = 0; s_pref = 0; s_fail = 0; }
  // HIST launches ONE EXTRA workgroup, index 0, that only zeroes the forward's small tables (flags | tile histogram | segment
  // bases) and publishes the launch's nonce; the working workgroups 1 .. n wait for the nonce right before their first atomic
  // on the tables, at the very end of the kernel -- ten microseconds later.  (Workgroups are dispatched in index order, so
  // workgroup 0 runs before anyone can wait for it.)  This replaces a zero-fill launch per forward; the bin scatter kernel
  // resets the word, so a replayed HIP graph -- same nonce, same buffer -- starts from "not ready" again.
  // a.nonce == 0 (MgsOptions.table_init = 1): an EARLIER LAUNCH of the stream zeroed the tables; workgroup 0 has nothing to do
  // and nobody waits for anybody -- correct by construction, one launch more per forward.
  if constexpr (HIST) {
    if (blockIdx.x == 0) {
      if (a.nonce == 0ull) return;
      if (a.wg0_delay < 0) return;  // (test hook, MgsOptions.dbg & 1024: the tables are never published -- the workers give up)
      for (int i = 0; i < a.wg0_delay; i++) __builtin_amdgcn_s_sleep(127);  // (test hook: the workers must wait this out)
      for (uint32_t i = threadIdx.x; i < a.tables_words; i += blockDim.x) a.tables[i] = 0u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(a.ready, a.nonce, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  // HIST launches a.zero_blocks more workgroups at the END of the grid that only zero the later backward's accumulator block
  // (17 MB at BASELINE configs[2]).  The working workgroups used to issue those stores first thing "on the side" -- but loads
  // and stores share one in-order counter, so their first wait for a load was also a wait for every store before it; and at
  // 100 000 Gaussians 98 working workgroups leave 158 CUs idle for the job.
  const int nwork = HIST ? (int)gridDim.x - 1 - a.zero_blocks : (int)gridDim.x;
  if constexpr (HIST) {
    if ((int)blockIdx.x > nwork) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const size_t zb = (size_t)((int)blockIdx.x - nwork - 1);
      for (size_t i = zb * blockDim.x + threadIdx.x; i < a.zero_f4; i += (size_t)a.zero_blocks * blockDim.x) a.zero_ptr[i] = z;
      return;
    }
  }
  const int blk = HIST ? (int)blockIdx.x - 1 : (int)blockIdx.x;  // working workgroup
  // Workgroups never straddle views: view = blockIdx / (workgroups per view), so the camera is wave-uniform.
  // Single view (V == 1): gi == idx and everything below is the plain per-Gaussian preprocess.
  const int bpv = (a.Pg + (int)blockDim.x - 1) / (int)blockDim.x;
  const int v = blk / bpv;
  const int gi = (blk - v * bpv) * (int)blockDim.x + (int)threadIdx.x;  // Gaussian
  const int idx = v * a.Pg + gi;                                                    // (virtual) instance owner
  const int Tv = a.tiles_x * a.tiles_y, T = Tv * a.V;
  if (a.zero_ptr && (!HIST || a.zero_blocks == 0)) {  // (tables in memory: on the side)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < a.zero_f4; i += (size_t)nwork * blockDim.x)
      a.zero_ptr[i] = z;
  }
  const bool batch = a.use_cam != 0;
  const float tanfovx = batch ? a.cam[v].tanfovx : a.tanfovx, tanfovy = batch ? a.cam[v].tanfovy : a.tanfovy;
  const float focal_x = batch ? a.cam[v].focal_x : a.focal_x, focal_y = batch ? a.cam[v].focal_y : a.focal_y;
  const int S = T;  // sort slices = tiles
  // direct binning (FwdPreArgs::direct_keys): two more tables behind the histogram -- this workgroup's reserved offset inside
  // every tile slice, and its write cursor there
  uint32_t* const s_base = s_hist + S;
  uint32_t* const s_cur = s_hist + 2 * S;
  const bool direct = HIST && a.direct_keys != nullptr;  // (grid-uniform)
  if constexpr (HIST) {
    for (int t = threadIdx.x; t < S; t += blockDim.x) s_hist[t] = 0;
    if (direct) for (int t = threadIdx.x; t < S; t += blockDim.x) s_cur[t] = 0;
    __syncthreads();
  }
  int32_t radius_out = 0;
  float depth_out = 0.f;
  uint32_t touched = 0;
  uint32_t touched_ref = 0;  // tiles of the reference's 3-sigma rect (what ITS num_rendered counts, rasterizer_impl.cu:280-284)
  int rx0 = 0, ry0 = 0, rx1 = 0, ry1 = 0;
  // thread 0 looks at the hand-shake word NOW (no wait: the value is examined at the end of the kernel, microseconds later,
  // when workgroup 0's store has long arrived -- a second look is only needed if this one came too early)
  unsigned long long seen = 0ull;
  if (HIST && threadIdx.x == 0 && a.nonce != 0ull) seen = __hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (gi < a.Pg) {
  const float* __restrict__ vm = batch ? a.cam[v].viewmatrix : a.viewmatrix;
  const float* __restrict__ pm = batch ? a.cam[v].projmatrix : a.projmatrix;
  const float* __restrict__ campos = batch ? a.cam[v].campos : a.campos;
  const V3 p = ld3(a.means3D, gi);
  const float view_z = row_view_z(vm[2], vm[6], vm[10], vm[14], p);
  if (view_z <= 0.2f) {  // auxiliary.h:154 near cull
    if (a.prefiltered) {
      if constexpr (HIST) atomicOr(&s_pref, 1u);  // (handed to the flags word at the end, after the tables are ready)
      else atomicOr(&g.flags[FLAG_PREFILTERED], 1u);
    }
  } else {
    const float hw = row_hom_w(pm[3], pm[7], pm[11], pm[15], p);
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float ndc_x = row_hom_xy(pm[0], pm[4], pm[8], pm[12], p) * p_w;
    const float ndc_y = row_hom_xy(pm[1], pm[5], pm[9], pm[13], p) * p_w;
    float c6[6];
    if (a.cov3D_precomp) {
#pragma unroll
      for (int i = 0; i < 6; i++) c6[i] = a.cov3D_precomp[6 * (size_t)gi + i];
    } else {
      cov3d_from_scale_rotation(a.rotations + 4 * (size_t)gi, ld3(a.scales, gi), a.scale_modifier, c6);
#pragma unroll
      for (int i = 0; i < 6; i++) g.cov3D[6 * (size_t)idx + i] = c6[i];
    }
    const ViewCov vc = view_cov(p, vm, focal_x, focal_y, tanfovx, tanfovy);
    float c00, c01, c11, Va[3], Vb[3];
    cov2d_from(vc, c6, c00, c01, c11, Va, Vb);
    c00 += 0.3f;
    c11 += 0.3f;
    const float det = cov2d_det(c00, c01, c11);
    if (det != 0.0f) {
      const float det_inv = 1.f / det;
      const float conx = c11 * det_inv, cony = -c01 * det_inv, conz = c00 * det_inv;
      const float mid = 0.5f * (c00 + c11);
      const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
      const float my_radius = ceilf(3.f * sqrtf(fmaxf(mid + disc, mid - disc)));
      const float px = ndc2pix(ndc_x, a.W), py = ndc2pix(ndc_y, a.H);
      int x0, y0, x1, y1;
      get_rect(px, py, (int)my_radius, a.tiles_x, a.tiles_y, x0, y0, x1, y1);
      touched_ref = (uint32_t)((x1 - x0) * (y1 - y0));
      if ((x1 - x0) * (y1 - y0) != 0) {
        const float opacity = a.opacities[gi];
        // bbox of {alpha >= 1/255} = {q(d) <= 2 ln(255 o)}: half extents sqrt(tau * cov_xx), sqrt(tau * cov_yy)
        // (cov = conic^-1 = the low-passed cov2D).  Inflated so it is conservative w.r.t. float rounding
        // of the per-pixel test; the per-pixel test itself stays exact.
        float hx = -1.f, hy = -1.f;
        if (opacity * 255.0f >= 0.999f) {
          const float tau = 2.0f * logf(fmaxf(opacity * 255.0f, 1.0f)) * 1.0001f + 1e-3f;
          hx = fminf(sqrtf(tau * fmaxf(c00, 0.f)) * 1.0001f + 0.01f, 1e6f);
          hy = fminf(sqrtf(tau * fmaxf(c11, 0.f)) * 1.0001f + 0.01f, 1e6f);
          if (!(det > 0.f) || !(hx == hx) || !(hy == hy)) { hx = 1e6f; hy = 1e6f; }  // degenerate: never cull
        }
        if (a.tight_bins) tight_rect(px, py, hx, hy, x0, y0, x1, y1);
        if (a.colors_precomp == nullptr) {
          // forward.cu:21-72
          const V3 cam = {campos[0], campos[1], campos[2]};
          V3 dir = p - cam;
          const float len = sqrtf(dot(dir, dir));
          dir = {dir.x / len, dir.y / len, dir.z / len};
          const float* __restrict__ sh = a.shs + (size_t)gi * a.M * 3;
          auto SH = [&](int k) { return v3(sh[3 * k], sh[3 * k + 1], sh[3 * k + 2]); };
          V3 res = SH_C0 * SH(0);
          if (a.D > 0) {
            const float x = dir.x, y = dir.y, z = dir.z;
            res = res - (SH_C1 * y) * SH(1) + (SH_C1 * z) * SH(2) - (SH_C1 * x) * SH(3);
            if (a.D > 1) {
              const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
              res = res + (SH_C2[0] * xy) * SH(4) + (SH_C2[1] * yz) * SH(5) +
                    (SH_C2[2] * (2.0f * zz - xx - yy)) * SH(6) + (SH_C2[3] * xz) * SH(7) +
                    (SH_C2[4] * (xx - yy)) * SH(8);
              if (a.D > 2) {
                res = res + (SH_C3[0] * y * (3.0f * xx - yy)) * SH(9) + (SH_C3[1] * xy * z) * SH(10) +
                      (SH_C3[2] * y * (4.0f * zz - xx - yy)) * SH(11) +
                      (SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * SH(12) +
                      (SH_C3[4] * x * (4.0f * zz - xx - yy)) * SH(13) + (SH_C3[5] * z * (xx - yy)) * SH(14) +
                      (SH_C3[6] * x * (xx - 3.0f * yy)) * SH(15);
              }
            }
          }
          res = res + v3(0.5f, 0.5f, 0.5f);
          g.clamped[idx] = (uint8_t)((res.x < 0 ? 1 : 0) | (res.y < 0 ? 2 : 0) | (res.z < 0 ? 4 : 0));
          g.rgb[3 * (size_t)idx + 0] = fmaxf(res.x, 0.f);
          g.rgb[3 * (size_t)idx + 1] = fmaxf(res.y, 0.f);
          g.rgb[3 * (size_t)idx + 2] = fmaxf(res.z, 0.f);
        }
        g.depths[idx] = view_z;
        depth_out = view_z;
        g.rec[2 * (size_t)idx] = make_float4(px, py, conx, cony);  // view-local pixel coordinates
        g.rec[2 * (size_t)idx + 1] = make_float4(conz, opacity, hx, hy);
        radius_out = (int32_t)my_radius;
        touched = (uint32_t)((y1 - y0) * (x1 - x0));
        rx0 = x0; ry0 = y0 + v * a.tiles_y; rx1 = x1; ry1 = y1 + v * a.tiles_y;  // atlas tile rows of the view
      }
    }
  }
  radii[idx] = radius_out;
  if (touched == 0) { rx0 = ry0 = rx1 = ry1 = 0; }
  g.rect[idx] = make_uint2((uint32_t)rx0 | ((uint32_t)rx1 << 16), (uint32_t)ry0 | ((uint32_t)ry1 << 16));
  }  // idx < P
  if constexpr (!HIST) {
    // Tables in memory (more tiles than the LDS tables hold, or bin_mode 0): one atomic per instance on the tile histogram
    // -- which the caller zeroed together with the flags --, the counts wave-reduced first.
    for (int y = ry0; y < ry1; y++)
      for (int x = rx0; x < rx1; x++) atomicAdd(&a.tile_hist[y * a.tiles_x + x], 1u);
    uint32_t ts = touched, tr = touched_ref;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      ts += (uint32_t)__shfl_xor((int)ts, d, 64);
      tr += (uint32_t)__shfl_xor((int)tr, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      if (ts) atomicAdd(&g.flags[FLAG_NUM_RENDERED], ts);
      if (tr) atomicAdd(a.ref_count, tr);
    }
  }
  if constexpr (HIST) {
    for (int y = ry0; y < ry1; y++)
      for (int x = rx0; x < rx1; x++) atomicAdd(&s_hist[y * a.tiles_x + x], 1u);
    {  // the workgroup's instance counts: summed over the wave first (1 024 atomics on one LDS word serialise)
      uint32_t ts = touched, tr = touched_ref;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        ts += (uint32_t)__shfl_xor((int)ts, d, 64);
        tr += (uint32_t)__shfl_xor((int)tr, d, 64);
      }
      if ((threadIdx.x & 63) == 0) {
        if (ts) atomicAdd(&s_total, ts);
        if (tr) atomicAdd(&s_total_ref, tr);
      }
    }
    if (threadIdx.x == 0) {  // the tables are zero once workgroup 0 has published this launch's nonce
      // (workgroup 0 is dispatched first and needs ~2 us.  The wait is BOUNDED: if the nonce has not come after about a
      //  second -- workgroup 0 held by a debugger, a device that lost the store -- this workgroup gives up WITHOUT touching
      //  the tables and leaves the launch's nonce in ready[1]; the bin scatter kernel, next in the chain, then publishes
      //  "nothing binned" and reports the failure through the status word, so that the host raises MGS_ERR_HIP for THIS
      //  call instead of losing the HIP context to a trap)
      for (uint32_t polls = 0; a.nonce != 0ull && seen != a.nonce; polls++) {
        if (polls == (1u << 23)) {
          s_fail = 1u;
          __hip_atomic_store(a.ready + 1, a.nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
        seen = __hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (s_fail) return;
    if (threadIdx.x == 0 && s_total) atomicAdd(&g.flags[FLAG_NUM_RENDERED], s_total);
    if (threadIdx.x == 0 && s_total_ref && a.ref_count) atomicAdd(a.ref_count, s_total_ref);
    if (threadIdx.x == 0 && s_pref) atomicOr(&g.flags[FLAG_PREFILTERED], 1u);
    // reserve this workgroup's slots inside every slice it contributes to; the bin scatter (same
    // blockDim partition of the Gaussians) reads the offsets back
    uint32_t* __restrict__ row = a.blk_base + (size_t)blk * S;
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
      const uint32_t c = s_hist[t];
      if (c) {
        const uint32_t base = atomicAdd(&a.tile_hist[t], c);
        row[t] = base;  // (kept in direct mode too: a later re-binning into ANOTHER workspace -- the capacity retry -- scatters)
        if (direct) s_base[t] = base;
      }
    }
    if (direct) {
      // ... and write the keys right away: tile t's slice starts at t * direct_stride whatever the other tiles hold (a tile
      // holds each Gaussian of its view at most once: the slice cannot overflow), so no prefix over the tiles is needed and
      // the bin scatter kernel's work is done here, on registers that still hold the rect and the depth
      __syncthreads();
      const uint64_t key = ((uint64_t)__float_as_uint(depth_out) << 32) | (uint32_t)idx;
      for (int y = ry0; y < ry1; y++)
        for (int x = rx0; x < rx1; x++) {
          const int t = y * a.tiles_x + x;
          a.direct_keys[(size_t)t * a.direct_stride + s_base[t] + atomicAdd(&s_cur[t], 1u)] = key;
        }
    }
  }
}

// Zero-fill of the small per-forward tables (flags | tile histogram | segment bases) as an ordinary kernel node: a captured
// hipMemsetAsync node was observed to run out of order with the kernels around it when a HIP graph is replayed after other
// work (ROCm 7.2), which corrupts the histogram; a kernel launch is ordered like every other launch of the chain.
__global__ void __launch_bounds__(256) zero_words_kernel(uint32_t* __restrict__ p, size_t head, size_t n16, size_t tail) {
  // [head words][n16 x 16 bytes][tail words]: the body is 16-byte aligned
  uint4* __restrict__ body = reinterpret_cast<uint4*>(p + head);
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
  for (size_t i = gid; i < n16; i += gsz) body[i] = make_uint4(0u, 0u, 0u, 0u);
  if (gid < head) p[gid] = 0u;
  if (gid < tail) p[head + 4 * n16 + gid] = 0u;
}
hipError_t launch_zero_bytes(void* p, size_t bytes, hipStream_t s) {  // p 4-byte aligned, bytes a multiple of 4
  if (bytes == 0) return hipSuccess;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  size_t words = bytes / 4;
  size_t head = ((16 - (a & 15u)) & 15u) / 4;
  if (head > words) head = words;
  const size_t n16 = (words - head) / 4, tail = (words - head) % 4;
  size_t blocks = (n16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), head, n16, tail);
  return hipGetLastError();
}

hipError_t launch_preprocess_fwd(const FwdPreArgs& a, const GeomView& g, int32_t* radii, hipStream_t s) {
  if (a.P <= 0) return hipSuccess;
  if (a.blk_base) {
    FwdPreArgs b = a;
    // one zeroing workgroup per 16 K float4 (16 stores per thread), at most 128
    b.zero_blocks = b.zero_ptr ? (int)std::min<size_t>(128, (b.zero_f4 + 16383) / 16384) : 0;
    const int pb = pre_block((size_t)a.Pg);
    hipLaunchKernelGGL(preprocess_fwd_kernel<true>, dim3(a.V * ((a.Pg + pb - 1) / pb) + 1 + b.zero_blocks),
                       dim3(pb), sizeof(uint32_t) * (size_t)a.tiles_x * a.tiles_y * a.V * (a.direct_keys ? 3 : 1), s, b, g, radii);
  } else
    hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3(a.V * ((a.Pg + 255) / 256)), dim3(256), 0, s, a, g, radii);
  return hipGetLastError();
}

// ---- frustum marking (rasterizer_impl.cu:54-66) ---------------------------------------------------
__global__ void mark_visible_kernel(int P, const float* __restrict__ means, const float* __restrict__ vm,
                                    uint8_t* __restrict__ present) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P) return;
  const V3 p = ld3(means, idx);
  const float z = row_view_z(vm[2], vm[6], vm[10], vm[14], p);  // p_view.z as in_frustum rounds it (auxiliary.h:139-164)
  present[idx] = (z <= 0.2f) ? 0 : 1;
}
hipError_t launch_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present,
                               hipStream_t s) {
  (void)proj;  // the projected point is computed but unused by the reference test (auxiliary.h:148-154)
  if (P <= 0) return hipSuccess;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
  return hipGetLastError();
}

// ---- fused backward preprocess: K9 (conic -> cov3D, mean) + K10 (projection, SH, scale/rot) --------
__global__ void __launch_bounds__(256) preprocess_bwd_kernel(BwdPreArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.P) return;
  const size_t i = (size_t)idx;
  float g_mean[3] = {0, 0, 0}, g_cov[6] = {0, 0, 0, 0, 0, 0}, g_scale[3] = {0, 0, 0}, g_rot[4] = {0, 0, 0, 0};
  float g_op = 0.f;
  const int n_sh = a.M;
  bool sh_written = false;  // the first visible view stores the SH gradient, later views add to it
  const V3 m = ld3(a.means3D, idx);
  // One pass per view (a single-view call has V == 1): the render-backward sums, the clamp flags, the per-view colour
  // gradient and the 2D outputs are [V][P][.]; everything per Gaussian is summed over the views in registers.
  for (int vw = 0; vw < a.V; vw++) {
    const size_t vi = (size_t)vw * a.P + i;  // instance owner (virtual id)
    const bool vis = a.radii[vi] > 0;
    float acc[8];
    {
      const float4 u = reinterpret_cast<const float4*>(a.acc8)[2 * vi];
      const float4 v = reinterpret_cast<const float4*>(a.acc8)[2 * vi + 1];
      acc[0] = u.x; acc[1] = u.y; acc[2] = u.z; acc[3] = u.w; acc[4] = v.x; acc[5] = v.y;
    }
    // render-backward sums: a non-visible Gaussian is in no tile list, so they are already zero.
    a.dL_dmeans2D[3 * vi + 0] = acc[0];
    a.dL_dmeans2D[3 * vi + 1] = acc[1];
    a.dL_dmeans2D[3 * vi + 2] = 0.f;
    g_op += acc[5];
    if (a.dL_dconic) {
      a.dL_dconic[4 * vi + 0] = acc[2]; a.dL_dconic[4 * vi + 1] = acc[3];
      a.dL_dconic[4 * vi + 2] = 0.f;    a.dL_dconic[4 * vi + 3] = acc[4];
    }
    if (!vis) continue;
    const bool batch = a.use_cam != 0;
    const float* __restrict__ vm = batch ? a.cam[vw].viewmatrix : a.viewmatrix;
    const float* __restrict__ proj = batch ? a.cam[vw].projmatrix : a.projmatrix;
    const float* __restrict__ campos = batch ? a.cam[vw].campos : a.campos;
    const float tanfovx = batch ? a.cam[vw].tanfovx : a.tanfovx, tanfovy = batch ? a.cam[vw].tanfovy : a.tanfovy;
    const float focal_x = batch ? a.cam[vw].focal_x : a.focal_x, focal_y = batch ? a.cam[vw].focal_y : a.focal_y;
    float gc[6] = {0, 0, 0, 0, 0, 0};  // this view's dL_dcov3D
    float c6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) c6[k] = a.cov3D[6 * (a.cov3D_per_view ? vi : i) + k];
    // ---- K9: backward.cu:144-274 ----
    const ViewCov vc = view_cov(m, vm, focal_x, focal_y, tanfovx, tanfovy);
    float c00, c01, c11, Va[3], Vb[3];
    cov2d_from(vc, c6, c00, c01, c11, Va, Vb);
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float x_grad_mul = (vc.txtz < -limx || vc.txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (vc.tytz < -limy || vc.tytz > limy) ? 0.f : 1.f;
    const float ca = c00 + 0.3f, cb = c01, cc = c11 + 0.3f;
    const float dcx = acc[2], dcy = acc[3], dcz = acc[4];
    const float denom = ca * cc - cb * cb;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    if (denom2inv != 0) {
      dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
      dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
      dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
      const float* A = vc.a; const float* B = vc.b;
      gc[0] = A[0] * A[0] * dL_da + A[0] * B[0] * dL_db + B[0] * B[0] * dL_dc;
      gc[3] = A[1] * A[1] * dL_da + A[1] * B[1] * dL_db + B[1] * B[1] * dL_dc;
      gc[5] = A[2] * A[2] * dL_da + A[2] * B[2] * dL_db + B[2] * B[2] * dL_dc;
      gc[1] = 2 * A[0] * A[1] * dL_da + (A[0] * B[1] + A[1] * B[0]) * dL_db + 2 * B[0] * B[1] * dL_dc;
      gc[2] = 2 * A[0] * A[2] * dL_da + (A[0] * B[2] + A[2] * B[0]) * dL_db + 2 * B[0] * B[2] * dL_dc;
      gc[4] = 2 * A[2] * A[1] * dL_da + (A[1] * B[2] + A[2] * B[1]) * dL_db + 2 * B[1] * B[2] * dL_dc;
    }
    float dT0[3], dT1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      dT0[j] = 2 * Va[j] * dL_da + Vb[j] * dL_db;
      dT1[j] = 2 * Vb[j] * dL_dc + Va[j] * dL_db;
    }
    const float dJ00 = vm[0] * dT0[0] + vm[4] * dT0[1] + vm[8] * dT0[2];
    const float dJ02 = vm[2] * dT0[0] + vm[6] * dT0[1] + vm[10] * dT0[2];
    const float dJ11 = vm[1] * dT1[0] + vm[5] * dT1[1] + vm[9] * dT1[2];
    const float dJ12 = vm[2] * dT1[0] + vm[6] * dT1[1] + vm[10] * dT1[2];
    const float tz = 1.f / vc.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    const float hx = focal_x, hy = focal_y;
    const float dtx = x_grad_mul * -hx * tz2 * dJ02;
    const float dty = y_grad_mul * -hy * tz2 * dJ12;
    const float dtz = -hx * tz2 * dJ00 - hy * tz2 * dJ11 + (2 * hx * vc.tx) * tz3 * dJ02 + (2 * hy * vc.ty) * tz3 * dJ12;
    g_mean[0] += vm[0] * dtx + vm[1] * dty + vm[2] * dtz;   // transformVec4x3Transpose
    g_mean[1] += vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
    g_mean[2] += vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
#pragma unroll
    for (int kk = 0; kk < 6; kk++) g_cov[kk] += gc[kk];
    // ---- K10: projection path, backward.cu:369-387 ----
    const float hw = proj[3] * m.x + proj[7] * m.y + proj[11] * m.z + proj[15];
    const float m_w = 1.0f / (hw + 0.0000001f);
    const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    const float gx = acc[0], gy = acc[1];
    g_mean[0] += (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
    g_mean[1] += (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
    g_mean[2] += (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
    // ---- SH backward, backward.cu:20-139 ----
    if (a.shs) {
      const V3 cam = {campos[0], campos[1], campos[2]};
      const V3 dir_orig = m - cam;
      const float len = sqrtf(dot(dir_orig, dir_orig));
      const V3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
      const float* __restrict__ sh = a.shs + i * n_sh * 3;
      auto SH = [&](int k) { return v3(sh[3 * k], sh[3 * k + 1], sh[3 * k + 2]); };
      const uint8_t cl = a.clamped[vi];
      V3 dRGB = ld3(a.dL_dcolor, vi);
      dRGB.x *= (cl & 1) ? 0.f : 1.f;
      dRGB.y *= (cl & 2) ? 0.f : 1.f;
      dRGB.z *= (cl & 4) ? 0.f : 1.f;
      float* __restrict__ o = a.dL_dsh + i * n_sh * 3;
      auto ST = [&](int k, float w) {
        if (sh_written) { o[3 * k] += w * dRGB.x; o[3 * k + 1] += w * dRGB.y; o[3 * k + 2] += w * dRGB.z; }
        else { o[3 * k] = w * dRGB.x; o[3 * k + 1] = w * dRGB.y; o[3 * k + 2] = w * dRGB.z; }
      };
      V3 dx = {0, 0, 0}, dy = {0, 0, 0}, dz = {0, 0, 0};
      const float x = dir.x, y = dir.y, z = dir.z;
      ST(0, SH_C0);
      int written = 1;
      if (a.D > 0) {
        ST(1, -SH_C1 * y); ST(2, SH_C1 * z); ST(3, -SH_C1 * x);
        written = 4;
        dx = (-SH_C1) * SH(3); dy = (-SH_C1) * SH(1); dz = SH_C1 * SH(2);
        if (a.D > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          ST(4, SH_C2[0] * xy); ST(5, SH_C2[1] * yz); ST(6, SH_C2[2] * (2.f * zz - xx - yy));
          ST(7, SH_C2[3] * xz); ST(8, SH_C2[4] * (xx - yy));
          written = 9;
          dx = dx + (SH_C2[0] * y) * SH(4) + (SH_C2[2] * 2.f * -x) * SH(6) + (SH_C2[3] * z) * SH(7) + (SH_C2[4] * 2.f * x) * SH(8);
          dy = dy + (SH_C2[0] * x) * SH(4) + (SH_C2[1] * z) * SH(5) + (SH_C2[2] * 2.f * -y) * SH(6) + (SH_C2[4] * 2.f * -y) * SH(8);
          dz = dz + (SH_C2[1] * y) * SH(5) + (SH_C2[2] * 2.f * 2.f * z) * SH(6) + (SH_C2[3] * x) * SH(7);
          if (a.D > 2) {
            ST(9, SH_C3[0] * y * (3.f * xx - yy)); ST(10, SH_C3[1] * xy * z);
            ST(11, SH_C3[2] * y * (4.f * zz - xx - yy)); ST(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
            ST(13, SH_C3[4] * x * (4.f * zz - xx - yy)); ST(14, SH_C3[5] * z * (xx - yy));
            ST(15, SH_C3[6] * x * (xx - 3.f * yy));
            written = 16;
            dx = dx + (SH_C3[0] * 3.f * 2.f * xy) * SH(9) + (SH_C3[1] * yz) * SH(10) + (SH_C3[2] * -2.f * xy) * SH(11) +
                 (SH_C3[3] * -3.f * 2.f * xz) * SH(12) + (SH_C3[4] * (-3.f * xx + 4.f * zz - yy)) * SH(13) +
                 (SH_C3[5] * 2.f * xz) * SH(14) + (SH_C3[6] * 3.f * (xx - yy)) * SH(15);
            dy = dy + (SH_C3[0] * 3.f * (xx - yy)) * SH(9) + (SH_C3[1] * xz) * SH(10) +
                 (SH_C3[2] * (-3.f * yy + 4.f * zz - xx)) * SH(11) + (SH_C3[3] * -3.f * 2.f * yz) * SH(12) +
                 (SH_C3[4] * -2.f * xy) * SH(13) + (SH_C3[5] * -2.f * yz) * SH(14) + (SH_C3[6] * -3.f * 2.f * xy) * SH(15);
            dz = dz + (SH_C3[1] * xy) * SH(10) + (SH_C3[2] * 4.f * 2.f * yz) * SH(11) +
                 (SH_C3[3] * 3.f * (2.f * zz - xx - yy)) * SH(12) + (SH_C3[4] * 4.f * 2.f * xz) * SH(13) +
                 (SH_C3[5] * (xx - yy)) * SH(14);
          }
        }
      }
      for (int k = written; k < n_sh; k++) ST(k, 0.f);  // coefficients above the active degree keep zero
      sh_written = true;
      const V3 dL_ddir = {dot(dx, dRGB), dot(dy, dRGB), dot(dz, dRGB)};
      // dnormvdv, auxiliary.h:107-117
      const V3 v = dir_orig;
      const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
      const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
      g_mean[0] += ((+sum2 - v.x * v.x) * dL_ddir.x - v.y * v.x * dL_ddir.y - v.z * v.x * dL_ddir.z) * invsum32;
      g_mean[1] += (-v.x * v.y * dL_ddir.x + (sum2 - v.y * v.y) * dL_ddir.y - v.z * v.y * dL_ddir.z) * invsum32;
      g_mean[2] += (-v.x * v.z * dL_ddir.x - v.y * v.z * dL_ddir.y + (sum2 - v.z * v.z) * dL_ddir.z) * invsum32;
    }
  }  // views
  if (a.dL_dsh && !sh_written)
    for (int k = 0; k < 3 * n_sh; k++) a.dL_dsh[i * 3 * n_sh + k] = 0.f;
  a.dL_dopacity[i] = g_op;
  {
    // ---- cov3D backward, backward.cu:278-341 (linear in dL_dcov3D: applied once to the sum over the views) ----
    if (a.scales) {
      float R[3][3];
      const float* q = a.rotations + 4 * i;
      quat_rows(q, R);
      const V3 s0 = ld3(a.scales, idx);
      const float sc[3] = {a.scale_modifier * s0.x, a.scale_modifier * s0.y, a.scale_modifier * s0.z};
      const float dS[3][3] = {{g_cov[0], 0.5f * g_cov[1], 0.5f * g_cov[2]},
                              {0.5f * g_cov[1], g_cov[3], 0.5f * g_cov[4]},
                              {0.5f * g_cov[2], 0.5f * g_cov[4], g_cov[5]}};
      // Q[c][r] = dL_dMt[c][r] = 2 * s_c * sum_k R[k][c] * dS[r][k]
      float Q[3][3];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
          Q[c][r] = 2.0f * (sc[c] * R[0][c] * dS[r][0] + sc[c] * R[1][c] * dS[r][1] + sc[c] * R[2][c] * dS[r][2]);
#pragma unroll
      for (int c = 0; c < 3; c++) g_scale[c] = R[0][c] * Q[c][0] + R[1][c] * Q[c][1] + R[2][c] * Q[c][2];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) Q[c][r] *= sc[c];
      const float r = q[0], x = q[1], y = q[2], z = q[3];
      g_rot[0] = 2 * z * (Q[0][1] - Q[1][0]) + 2 * y * (Q[2][0] - Q[0][2]) + 2 * x * (Q[1][2] - Q[2][1]);
      g_rot[1] = 2 * y * (Q[1][0] + Q[0][1]) + 2 * z * (Q[2][0] + Q[0][2]) + 2 * r * (Q[1][2] - Q[2][1]) - 4 * x * (Q[2][2] + Q[1][1]);
      g_rot[2] = 2 * x * (Q[1][0] + Q[0][1]) + 2 * r * (Q[2][0] - Q[0][2]) + 2 * z * (Q[1][2] + Q[2][1]) - 4 * y * (Q[2][2] + Q[0][0]);
      g_rot[3] = 2 * r * (Q[0][1] - Q[1][0]) + 2 * x * (Q[2][0] + Q[0][2]) + 2 * y * (Q[1][2] + Q[2][1]) - 4 * z * (Q[1][1] + Q[0][0]);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) a.dL_dmeans3D[3 * i + k] = g_mean[k];
#pragma unroll
  for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * i + k] = g_cov[k];
  if (a.dL_dscales) {
#pragma unroll
    for (int k = 0; k < 3; k++) a.dL_dscales[3 * i + k] = g_scale[k];
  }
  if (a.dL_drot) {
#pragma unroll
    for (int k = 0; k < 4; k++) a.dL_drot[4 * i + k] = g_rot[k];
  }
}

hipError_t launch_preprocess_bwd(const BwdPreArgs& a, hipStream_t s) {
  if (a.P <= 0) return hipSuccess;
  // one wave per workgroup for small sets: 256 single-wave workgroups reach every CU at ManiGaussian's 16 384 Gaussians
  // (8.4 -> 7.8 us by the stage timers; no difference from 100 000 Gaussians up)
  const int bb = a.P <= 32768 ? 64 : 256;
  hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((a.P + bb - 1) / bb), dim3(bb), 0, s, a);
  return hipGetLastError();
}

}  // namespace mgs
