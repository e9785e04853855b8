/*
 * mgs_oracle.c -- CPU restatement ("Oracle B") of the reference Gaussian rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (manigaussian_amd/, the C-ABI
 * library, diff_gaussian_rasterization/) may import, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Parity status: PINNED against outputs of the reference itself.  The reference holds no golden
 * vectors, known-answer tests or fixtures for this path, so they were made: the
 * reference's own forward.cu / backward.cu / rasterizer_impl.cu are compiled unmodified with
 * hipcc (oracle/Makefile target _ref, through oracle/refshim for the CUDA-runtime / CUB /
 * cooperative-groups names and the un-vendored glm), run on an MI355X, and their outputs are
 * committed as tests/golden/ref/ (tests/golden/make_golden_ref.py).
 * tests/test_oracle.py::test_oracle_b_matches_reference_kernels checks this file against them
 * (num_rendered and radii bit-exact, images 1e-5, gradients 1e-3 of the max).  Independent pins
 * remain: (i) the autograd oracle (oracle/oracle_a.py), (ii) closed-form cases, (iii) float64
 * finite differences -- see tests/test_oracle.py.
 *
 * Every function cites the reference lines it follows.  Paths are relative to
 *   RAST = /root/reference/third_party/gaussian-splatting/submodules/diff-gaussian-rasterization
 *
 * glm is absent from the reference tree; its mat3/vec3 algebra is restated here with the
 * same conventions: mat3 is COLUMN-major, constructor arguments fill columns, m[c][r].
 *
 * Arithmetic is IEEE float32 like the reference.  The one deliberate difference: the
 * per-Gaussian sums that the reference builds with float atomicAdd in arbitrary order
 * (RAST/cuda_rasterizer/backward.cu:541-590) are accumulated here in float64 and rounded
 * once, so the oracle is deterministic and is the best estimate of that unordered sum.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16 /* RAST/cuda_rasterizer/config.h:17 */
#define BLOCK_Y 16 /* RAST/cuda_rasterizer/config.h:18 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NUM_CHANNELS 3 /* RAST/cuda_rasterizer/config.h:15 */
#define MAX_F 64

/* RAST/cuda_rasterizer/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                              -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                              0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                              -0.5900435899266435f};

typedef struct { float x, y, z; } vec3;
typedef struct { float m[3][3]; } mat3; /* m[col][row], glm convention */

static inline vec3 v3(float x, float y, float z) { vec3 v = {x, y, z}; return v; }
static inline vec3 v3add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v3sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v3scale(float s, vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
static inline float v3dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* glm::mat3(a,b,c, d,e,f, g,h,i): columns (a,b,c) (d,e,f) (g,h,i) */
static inline mat3 mat3_cols(float a, float b, float c, float d, float e, float f, float g, float h,
                             float i) {
  mat3 r;
  r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
  r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
  r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
  return r;
}
/* glm operator*: R[c][r] = sum_k A[k][r] * B[c][k] */
static inline mat3 mat3_mul(mat3 A, mat3 B) {
  mat3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++)
      R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
  return R;
}
static inline mat3 mat3_T(mat3 A) {
  mat3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
  return R;
}
static inline mat3 mat3_scale(float s, mat3 A) {
  mat3 R;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) R.m[c][r] = s * A.m[c][r];
  return R;
}

/* RAST/cuda_rasterizer/auxiliary.h:41-44 (double arithmetic as written there: 1.0, 0.5 literals) */
static inline float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* RAST/cuda_rasterizer/auxiliary.h:46-56 -- C-cast truncation, clamp to the tile grid */
static inline void getRect(float px, float py, int max_radius, int gx, int gy, uint32_t* rmin,
                           uint32_t* rmax) {
  rmin[0] = (uint32_t)imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
  rmin[1] = (uint32_t)imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
  rmax[0] = (uint32_t)imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
  rmax[1] = (uint32_t)imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* RAST/cuda_rasterizer/auxiliary.h:58-66 */
static inline vec3 transformPoint4x3(vec3 p, const float* m) {
  return v3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
/* RAST/cuda_rasterizer/auxiliary.h:68-77 */
static inline void transformPoint4x4(vec3 p, const float* m, float out[4]) {
  out[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
  out[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
  out[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
  out[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}
/* RAST/cuda_rasterizer/auxiliary.h:89-97 */
static inline vec3 transformVec4x3Transpose(vec3 p, const float* m) {
  return v3(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
/* RAST/cuda_rasterizer/auxiliary.h:107-117 */
static inline vec3 dnormvdv(vec3 v, vec3 dv) {
  float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  vec3 r;
  r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
  r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
  r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
  return r;
}

/* RAST/cuda_rasterizer/auxiliary.h:139-164.  prefiltered violation = device __trap() in the
 * reference; here it sets *trap and the caller reports failure. */
static inline int in_frustum(int idx, const float* orig_points, const float* viewmatrix,
                             const float* projmatrix, int prefiltered, vec3* p_view, int* trap) {
  vec3 p_orig = v3(orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]);
  (void)projmatrix; /* p_proj is computed but unused by the test, auxiliary.h:148-151 */
  *p_view = transformPoint4x3(p_orig, viewmatrix);
  if (p_view->z <= 0.2f) {
    if (prefiltered) *trap = 1;
    return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* State kept between forward and backward (the reference's geom/binning/img buffers,          */
/* RAST/cuda_rasterizer/rasterizer_impl.h:30-63)                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct OrcState {
  int P, D, M, F, W, H, R, include_feature;
  int gx, gy;
  float tanfovx, tanfovy, scale_modifier;
  float bg[3], view[16], proj[16], campos[3];
  const float *means3D, *shs, *colors_precomp, *lang, *scales, *rotations, *cov3D_precomp; /* borrowed */
  /* GeometryState */
  float* depths;
  uint8_t* clamped; /* 3P */
  int* radii;
  float* means2D;        /* 2P */
  float* cov3D;          /* 6P */
  float* conic_opacity;  /* 4P */
  float* rgb;            /* 3P */
  uint32_t* tiles_touched;
  uint32_t* point_offsets;
  /* BinningState */
  uint64_t* keys;
  uint32_t* point_list;
  /* ImageState */
  float* final_T;
  uint32_t* n_contrib;
  uint32_t* ranges; /* 2T */
} OrcState;

/* RAST/cuda_rasterizer/forward.cu:21-72 */
static vec3 computeColorFromSH_fwd(int idx, int deg, int max_coeffs, const float* means, vec3 campos,
                                   const float* shs, uint8_t* clamped) {
  vec3 pos = v3(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
  vec3 dir = v3sub(pos, campos);
  float len = sqrtf(v3dot(dir, dir));
  dir = v3(dir.x / len, dir.y / len, dir.z / len);
  const vec3* sh = ((const vec3*)shs) + (size_t)idx * max_coeffs;
  vec3 result = v3scale(SH_C0, sh[0]);
  if (deg > 0) {
    float x = dir.x, y = dir.y, z = dir.z;
    result = v3add(v3sub(v3add(v3sub(result, v3scale(SH_C1 * y, sh[1])), v3scale(SH_C1 * z, sh[2])),
                         v3scale(SH_C1 * x, sh[3])),
                   v3(0, 0, 0));
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z;
      float xy = x * y, yz = y * z, xz = x * z;
      result = v3add(result, v3scale(SH_C2[0] * xy, sh[4]));
      result = v3add(result, v3scale(SH_C2[1] * yz, sh[5]));
      result = v3add(result, v3scale(SH_C2[2] * (2.0f * zz - xx - yy), sh[6]));
      result = v3add(result, v3scale(SH_C2[3] * xz, sh[7]));
      result = v3add(result, v3scale(SH_C2[4] * (xx - yy), sh[8]));
      if (deg > 2) {
        result = v3add(result, v3scale(SH_C3[0] * y * (3.0f * xx - yy), sh[9]));
        result = v3add(result, v3scale(SH_C3[1] * xy * z, sh[10]));
        result = v3add(result, v3scale(SH_C3[2] * y * (4.0f * zz - xx - yy), sh[11]));
        result = v3add(result, v3scale(SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), sh[12]));
        result = v3add(result, v3scale(SH_C3[4] * x * (4.0f * zz - xx - yy), sh[13]));
        result = v3add(result, v3scale(SH_C3[5] * z * (xx - yy), sh[14]));
        result = v3add(result, v3scale(SH_C3[6] * x * (xx - 3.0f * yy), sh[15]));
      }
    }
  }
  result = v3add(result, v3(0.5f, 0.5f, 0.5f));
  clamped[3 * idx + 0] = (result.x < 0);
  clamped[3 * idx + 1] = (result.y < 0);
  clamped[3 * idx + 2] = (result.z < 0);
  return v3(fmaxf(result.x, 0.0f), fmaxf(result.y, 0.0f), fmaxf(result.z, 0.0f));
}

/* RAST/cuda_rasterizer/forward.cu:75-114 (forward) -- shared with backward.cu:162-195 recompute.
 * Returns T and Vrk through pointers when asked. */
static void cov2D_core(vec3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                       const float* cov3D, const float* viewmatrix, vec3* t_out, mat3* T_out,
                       mat3* Vrk_out, mat3* W_out, mat3* cov_out, float* txtz_o, float* tytz_o) {
  vec3 t = transformPoint4x3(mean, viewmatrix);
  const float limx = 1.3f * tan_fovx;
  const float limy = 1.3f * tan_fovy;
  const float txtz = t.x / t.z;
  const float tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  mat3 J = mat3_cols(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z,
                     -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
  mat3 W = mat3_cols(viewmatrix[0], viewmatrix[4], viewmatrix[8], viewmatrix[1], viewmatrix[5],
                     viewmatrix[9], viewmatrix[2], viewmatrix[6], viewmatrix[10]);
  mat3 T = mat3_mul(W, J);
  mat3 Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4],
                       cov3D[5]);
  mat3 cov = mat3_mul(mat3_mul(mat3_T(T), mat3_T(Vrk)), T);
  *t_out = t; *T_out = T; *Vrk_out = Vrk; *W_out = W; *cov_out = cov;
  *txtz_o = txtz; *tytz_o = tytz;
}

/* RAST/cuda_rasterizer/forward.cu:119-153.  Quaternion used UN-normalised, order (r,x,y,z). */
static void computeCov3D_fwd(vec3 scale, float mod, const float* rot, float* cov3D) {
  mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
  S.m[0][0] = mod * scale.x;
  S.m[1][1] = mod * scale.y;
  S.m[2][2] = mod * scale.z;
  float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                     2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                     2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
  mat3 Mm = mat3_mul(S, R);
  mat3 Sigma = mat3_mul(mat3_T(Mm), Mm);
  cov3D[0] = Sigma.m[0][0];
  cov3D[1] = Sigma.m[0][1];
  cov3D[2] = Sigma.m[0][2];
  cov3D[3] = Sigma.m[1][1];
  cov3D[4] = Sigma.m[1][2];
  cov3D[5] = Sigma.m[2][2];
}

/* RAST/cuda_rasterizer/forward.cu:156-257 (preprocessCUDA) */
static int preprocess_fwd(OrcState* s, const float* opacities, int prefiltered) {
  const int P = s->P;
  const float focal_y = s->H / (2.0f * s->tanfovy); /* rasterizer_impl.cu:225-226 */
  const float focal_x = s->W / (2.0f * s->tanfovx);
  int trap = 0;
#pragma omp parallel for schedule(static) reduction(| : trap)
  for (int idx = 0; idx < P; idx++) {
    s->radii[idx] = 0;
    s->tiles_touched[idx] = 0;
    vec3 p_view;
    int tr = 0;
    if (!in_frustum(idx, s->means3D, s->view, s->proj, prefiltered, &p_view, &tr)) {
      trap |= tr;
      continue;
    }
    vec3 p_orig = v3(s->means3D[3 * idx], s->means3D[3 * idx + 1], s->means3D[3 * idx + 2]);
    float p_hom[4];
    transformPoint4x4(p_orig, s->proj, p_hom);
    float p_w = 1.0f / (p_hom[3] + 0.0000001f);
    vec3 p_proj = v3(p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w);

    const float* cov3D;
    if (s->cov3D_precomp) {
      cov3D = s->cov3D_precomp + (size_t)idx * 6;
    } else {
      computeCov3D_fwd(v3(s->scales[3 * idx], s->scales[3 * idx + 1], s->scales[3 * idx + 2]),
                       s->scale_modifier, s->rotations + 4 * (size_t)idx, s->cov3D + (size_t)idx * 6);
      cov3D = s->cov3D + (size_t)idx * 6;
    }
    vec3 t; mat3 T, Vrk, Wm, cov2; float txtz, tytz;
    cov2D_core(p_orig, focal_x, focal_y, s->tanfovx, s->tanfovy, cov3D, s->view, &t, &T, &Vrk, &Wm,
               &cov2, &txtz, &tytz);
    cov2.m[0][0] += 0.3f; /* forward.cu:111-112 low-pass */
    cov2.m[1][1] += 0.3f;
    float cx = cov2.m[0][0], cy = cov2.m[0][1], cz = cov2.m[1][1];

    float det = (cx * cz - cy * cy);
    if (det == 0.0f) continue;
    float det_inv = 1.f / det;
    float conic_x = cz * det_inv, conic_y = -cy * det_inv, conic_z = cx * det_inv;

    float mid = 0.5f * (cx + cz);
    float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    float pix = ndc2Pix(p_proj.x, s->W), piy = ndc2Pix(p_proj.y, s->H);
    uint32_t rmin[2], rmax[2];
    getRect(pix, piy, (int)my_radius, s->gx, s->gy, rmin, rmax);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;

    if (!s->colors_precomp) {
      vec3 c = computeColorFromSH_fwd(idx, s->D, s->M, s->means3D,
                                      v3(s->campos[0], s->campos[1], s->campos[2]), s->shs, s->clamped);
      s->rgb[idx * NUM_CHANNELS + 0] = c.x;
      s->rgb[idx * NUM_CHANNELS + 1] = c.y;
      s->rgb[idx * NUM_CHANNELS + 2] = c.z;
    }
    s->depths[idx] = p_view.z;
    s->radii[idx] = (int)my_radius;
    s->means2D[2 * idx] = pix;
    s->means2D[2 * idx + 1] = piy;
    s->conic_opacity[4 * idx + 0] = conic_x;
    s->conic_opacity[4 * idx + 1] = conic_y;
    s->conic_opacity[4 * idx + 2] = conic_z;
    s->conic_opacity[4 * idx + 3] = opacities[idx];
    s->tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
  }
  return trap;
}

/* Stable LSD radix sort of (u64 key, u32 value) pairs -- stands in for
 * cub::DeviceRadixSort::SortPairs (RAST/cuda_rasterizer/rasterizer_impl.cu:306-311), which is a
 * stable LSD radix sort over key bits [0, 32+bit).  Bits above 32+bit are zero, so sorting all
 * 64 bits gives the same order. */
static void radix_sort_pairs(uint64_t* keys, uint32_t* vals, size_t n) {
  if (n == 0) return;
  uint64_t* k2 = (uint64_t*)malloc(n * sizeof(uint64_t));
  uint32_t* v2 = (uint32_t*)malloc(n * sizeof(uint32_t));
  uint64_t *ks = keys, *kd = k2;
  uint32_t *vs = vals, *vd = v2;
  for (int pass = 0; pass < 8; pass++) {
    size_t hist[256];
    memset(hist, 0, sizeof(hist));
    int shift = pass * 8;
    for (size_t i = 0; i < n; i++) hist[(ks[i] >> shift) & 0xff]++;
    if (hist[(ks[0] >> shift) & 0xff] == n) continue; /* all in one bucket: order unchanged */
    size_t sum = 0;
    for (int b = 0; b < 256; b++) { size_t c = hist[b]; hist[b] = sum; sum += c; }
    for (size_t i = 0; i < n; i++) {
      size_t d = hist[(ks[i] >> shift) & 0xff]++;
      kd[d] = ks[i];
      vd[d] = vs[i];
    }
    uint64_t* tk = ks; ks = kd; kd = tk;
    uint32_t* tv = vs; vs = vd; vd = tv;
  }
  if (ks != keys) {
    memcpy(keys, ks, n * sizeof(uint64_t));
    memcpy(vals, vs, n * sizeof(uint32_t));
  }
  free(k2);
  free(v2);
}

/* RAST/cuda_rasterizer/rasterizer_impl.cu:280-320: InclusiveSum, duplicateWithKeys (:70-111),
 * SortPairs, identifyTileRanges (:116-138). */
static void binning(OrcState* s) {
  const int P = s->P;
  uint32_t acc = 0;
  for (int i = 0; i < P; i++) { acc += s->tiles_touched[i]; s->point_offsets[i] = acc; }
  s->R = (int)acc;
  const int T = s->gx * s->gy;
  s->ranges = (uint32_t*)calloc((size_t)2 * T, sizeof(uint32_t)); /* cudaMemset 0, :313 */
  s->keys = (uint64_t*)malloc(((size_t)s->R + 1) * sizeof(uint64_t));
  s->point_list = (uint32_t*)malloc(((size_t)s->R + 1) * sizeof(uint32_t));
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    if (s->radii[idx] > 0) {
      uint32_t off = (idx == 0) ? 0 : s->point_offsets[idx - 1];
      uint32_t rmin[2], rmax[2];
      getRect(s->means2D[2 * idx], s->means2D[2 * idx + 1], s->radii[idx], s->gx, s->gy, rmin, rmax);
      uint32_t dbits;
      memcpy(&dbits, &s->depths[idx], 4);
      for (uint32_t y = rmin[1]; y < rmax[1]; y++)
        for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
          uint64_t key = (uint64_t)(y * (uint32_t)s->gx + x);
          key <<= 32;
          key |= dbits;
          s->keys[off] = key;
          s->point_list[off] = (uint32_t)idx;
          off++;
        }
    }
  }
  radix_sort_pairs(s->keys, s->point_list, (size_t)s->R);
  const int L = s->R;
  for (int idx = 0; idx < L; idx++) {
    uint32_t currtile = (uint32_t)(s->keys[idx] >> 32);
    if (idx == 0)
      s->ranges[2 * currtile] = 0;
    else {
      uint32_t prevtile = (uint32_t)(s->keys[idx - 1] >> 32);
      if (currtile != prevtile) {
        s->ranges[2 * prevtile + 1] = (uint32_t)idx;
        s->ranges[2 * currtile] = (uint32_t)idx;
      }
    }
    if (idx == L - 1) s->ranges[2 * currtile + 1] = (uint32_t)L;
  }
}

/* RAST/cuda_rasterizer/forward.cu:262-398 (renderCUDA fwd).  One tile at a time, one pixel at a
 * time: the reference's 256-entry batches and block-level votes do not change any pixel's
 * result, only when the block stops fetching. */
static void render_fwd(OrcState* s, float* out_color, float* out_feat) {
  const int W = s->W, H = s->H, F = s->F;
  const float* features = s->colors_precomp ? s->colors_precomp : s->rgb;
  const int T = s->gx * s->gy;
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < T; tile++) {
    const int ty = tile / s->gx, tx = tile % s->gx;
    const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ly++)
      for (int lx = 0; lx < BLOCK_X; lx++) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const uint32_t pix_id = (uint32_t)W * py + px;
        const float pixfx = (float)px, pixfy = (float)py;
        float Tt = 1.0f;
        uint32_t contributor = 0, last_contributor = 0;
        float C[NUM_CHANNELS] = {0, 0, 0};
        float Fv[MAX_F];
        for (int ch = 0; ch < F; ch++) Fv[ch] = 0.f;
        for (uint32_t k = r0; k < r1; k++) {
          contributor++;
          const uint32_t id = s->point_list[k];
          const float dx = s->means2D[2 * id] - pixfx, dy = s->means2D[2 * id + 1] - pixfy;
          const float* co = s->conic_opacity + 4 * (size_t)id;
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float alpha = fminf(0.99f, co[3] * expf(power));
          if (alpha < 1.0f / 255.0f) continue;
          const float test_T = Tt * (1 - alpha);
          if (test_T < 0.0001f) break; /* done = true; the Gaussian is NOT blended (:356-360) */
          for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] += features[id * NUM_CHANNELS + ch] * alpha * Tt;
          if (s->include_feature)
            for (int ch = 0; ch < F; ch++) Fv[ch] += s->lang[(size_t)id * F + ch] * alpha * Tt;
          Tt = test_T;
          last_contributor = contributor;
        }
        s->final_T[pix_id] = Tt;
        s->n_contrib[pix_id] = last_contributor;
        for (int ch = 0; ch < NUM_CHANNELS; ch++)
          out_color[(size_t)ch * H * W + pix_id] = C[ch] + Tt * s->bg[ch];
        if (s->include_feature) /* features get NO background term (:393) */
          for (int ch = 0; ch < F; ch++) out_feat[(size_t)ch * H * W + pix_id] = Fv[ch];
      }
  }
}

/* Test aid, not part of the reference: marks pixels whose forward walk passes within rel_eps of one of
 * the reference's three hard decisions (power > 0, alpha < 1/255, T*(1-alpha) < 1e-4; forward.cu:345-360).
 * At such a pixel two correct implementations that differ by one ulp in exp() can legitimately blend a
 * different set of Gaussians (a step of up to ~alpha*T*c = 4e-3), so parity tests compare them with a
 * looser bound and require them to be rare. */
void orc_fragile_mask(const OrcState* s, float rel_eps, uint8_t* mask, uint8_t* gmask /*[P] or NULL*/) {
  const int W = s->W, H = s->H;
  const int T = s->gx * s->gy;
  memset(mask, 0, (size_t)W * H);
  if (gmask) memset(gmask, 0, (size_t)s->P);
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < T; tile++) {
    const int ty = tile / s->gx, tx = tile % s->gx;
    const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ly++)
      for (int lx = 0; lx < BLOCK_X; lx++) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        float Tt = 1.0f;
        uint8_t frag = 0;
        for (uint32_t k = r0; k < r1; k++) {
          const uint32_t id = s->point_list[k];
          const float dx = s->means2D[2 * id] - (float)px, dy = s->means2D[2 * id + 1] - (float)py;
          const float* co = s->conic_opacity + 4 * (size_t)id;
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          uint8_t f = 0;
          if (fabsf(power) < 1e-5f) f = 1;
          const float araw = co[3] * expf(fminf(power, 0.f));
          const float alpha = fminf(0.99f, araw);
          if (fabsf(alpha * 255.0f - 1.0f) < rel_eps) f = 1;
          const float test_T = Tt * (1 - alpha);
          if (alpha >= 1.0f / 255.0f && fabsf(test_T * 10000.0f - 1.0f) < rel_eps) f = 1;
          if (f) { frag = 1; if (gmask) gmask[id] = 1; } /* benign race: every writer stores 1 */
          if (power > 0.0f) continue;
          if (alpha < 1.0f / 255.0f) continue;
          if (test_T < 0.0001f) break;
          Tt = test_T;
        }
        mask[(size_t)W * py + px] = frag;
      }
  }
}

static void orc_free_internal(OrcState* s) {
  free(s->depths); free(s->clamped); free(s->radii); free(s->means2D); free(s->cov3D);
  free(s->conic_opacity); free(s->rgb); free(s->tiles_touched); free(s->point_offsets);
  free(s->keys); free(s->point_list); free(s->final_T); free(s->n_contrib); free(s->ranges);
}

void orc_free(OrcState* s) {
  if (!s) return;
  orc_free_internal(s);
  free(s);
}

void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* RAST/cuda_rasterizer/rasterizer_impl.cu:198-355 (Rasterizer::forward) with the output
 * conventions of RAST/rasterize_points.cu:35-128 (zero-filled outputs, P==0 short-circuit).
 * Pointers that the reference receives as nullptr (empty tensors) are passed as NULL.
 * Returns NULL if the prefiltered trap fires. */
OrcState* orc_forward(int P, int D, int M, int F, const float* bg, int W, int H, const float* means3D,
                      const float* shs, const float* colors_precomp, const float* lang,
                      const float* opacities, const float* scales, float scale_modifier,
                      const float* rotations, const float* cov3D_precomp, const float* view,
                      const float* proj, const float* campos, float tanfovx, float tanfovy,
                      int prefiltered, int include_feature, float* out_color, float* out_feat,
                      int* radii_out, int* num_rendered) {
  if (F > MAX_F) return NULL;
  OrcState* s = (OrcState*)calloc(1, sizeof(OrcState));
  s->P = P; s->D = D; s->M = M; s->F = F; s->W = W; s->H = H;
  s->include_feature = include_feature;
  s->gx = (W + BLOCK_X - 1) / BLOCK_X;
  s->gy = (H + BLOCK_Y - 1) / BLOCK_Y;
  s->tanfovx = tanfovx; s->tanfovy = tanfovy; s->scale_modifier = scale_modifier;
  memcpy(s->bg, bg, 12); memcpy(s->view, view, 64); memcpy(s->proj, proj, 64);
  memcpy(s->campos, campos, 12);
  s->means3D = means3D; s->shs = shs; s->colors_precomp = colors_precomp; s->lang = lang;
  s->scales = scales; s->rotations = rotations; s->cov3D_precomp = cov3D_precomp;
  const size_t N = (size_t)W * H;
  memset(out_color, 0, NUM_CHANNELS * N * sizeof(float)); /* torch::full(0), rasterize_points.cu:70 */
  if (include_feature) memset(out_feat, 0, (size_t)F * N * sizeof(float));
  size_t Pa = (size_t)(P > 0 ? P : 1);
  s->depths = (float*)calloc(Pa, 4);
  s->clamped = (uint8_t*)calloc(Pa * 3, 1);
  s->radii = (int*)calloc(Pa, 4);
  s->means2D = (float*)calloc(Pa * 2, 4);
  s->cov3D = (float*)calloc(Pa * 6, 4);
  s->conic_opacity = (float*)calloc(Pa * 4, 4);
  s->rgb = (float*)calloc(Pa * 3, 4);
  s->tiles_touched = (uint32_t*)calloc(Pa, 4);
  s->point_offsets = (uint32_t*)calloc(Pa, 4);
  s->final_T = (float*)calloc(N ? N : 1, 4);
  s->n_contrib = (uint32_t*)calloc(N ? N : 1, 4);
  *num_rendered = 0;
  if (P == 0) { /* rasterize_points.cu:92 */
    s->ranges = (uint32_t*)calloc((size_t)2 * s->gx * s->gy + 2, 4);
    return s;
  }
  if (NUM_CHANNELS != 3 && colors_precomp == NULL) { orc_free(s); return NULL; }
  if (preprocess_fwd(s, opacities, prefiltered)) { orc_free(s); return NULL; }
  binning(s);
  render_fwd(s, out_color, out_feat);
  if (radii_out) memcpy(radii_out, s->radii, (size_t)P * 4);
  *num_rendered = s->R;
  return s;
}

/* ---------------------------------------- backward ---------------------------------------- */

/* RAST/cuda_rasterizer/backward.cu:399-593 (renderCUDA bwd).  Per-Gaussian sums in float64
 * (see file header). */
static void render_bwd(const OrcState* s, const float* dL_dpixels, const float* dL_dpixels_F,
                       double* a_mean2D /*2P*/, double* a_conic /*3P: x,y,w*/, double* a_opacity,
                       double* a_colors /*3P*/, double* a_feat /*F*P*/) {
  const int W = s->W, H = s->H, F = s->F;
  const float* colors = s->colors_precomp ? s->colors_precomp : s->rgb;
  const int T = s->gx * s->gy;
  const float ddelx_dx = (float)(0.5 * W); /* :476-477 */
  const float ddely_dy = (float)(0.5 * H);
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < T; tile++) {
    const int ty = tile / s->gx, tx = tile % s->gx;
    const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
    for (int ly = 0; ly < BLOCK_Y; ly++)
      for (int lx = 0; lx < BLOCK_X; lx++) {
        const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
        if (!(px < W && py < H)) continue;
        const uint32_t pix_id = (uint32_t)W * py + px;
        const float pixfx = (float)px, pixfy = (float)py;
        const float T_final = s->final_T[pix_id];
        float Tt = T_final;
        uint32_t contributor = r1 - r0;
        const uint32_t last_contributor = s->n_contrib[pix_id];
        float accum_rec[NUM_CHANNELS] = {0, 0, 0};
        float dL_dpixel[NUM_CHANNELS];
        for (int i = 0; i < NUM_CHANNELS; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
        float last_alpha = 0;
        float last_color[NUM_CHANNELS] = {0, 0, 0};
        float accum_rec_F[MAX_F], dL_dpixel_F[MAX_F], last_F[MAX_F];
        for (int i = 0; i < F; i++) { accum_rec_F[i] = 0; dL_dpixel_F[i] = 0; last_F[i] = 0; }
        if (s->include_feature)
          for (int i = 0; i < F; i++) dL_dpixel_F[i] = dL_dpixels_F[(size_t)i * H * W + pix_id];

        for (uint32_t k = r1; k-- > r0;) { /* back to front, :487 */
          contributor--;
          if (contributor >= last_contributor) continue;
          const uint32_t id = s->point_list[k];
          const float dx = s->means2D[2 * id] - pixfx, dy = s->means2D[2 * id + 1] - pixfy;
          const float* co = s->conic_opacity + 4 * (size_t)id;
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float G = expf(power);
          const float alpha = fminf(0.99f, co[3] * G);
          if (alpha < 1.0f / 255.0f) continue;

          Tt = Tt / (1.f - alpha);
          const float dchannel_dcolor = alpha * Tt;
          float dL_dalpha = 0.0f;
          for (int ch = 0; ch < NUM_CHANNELS; ch++) {
            const float c = colors[id * NUM_CHANNELS + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = c;
            const float dL_dchannel = dL_dpixel[ch];
            dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
            double v = (double)(dchannel_dcolor * dL_dchannel);
#pragma omp atomic
            a_colors[(size_t)id * NUM_CHANNELS + ch] += v;
          }
          if (s->include_feature) {
            for (int ch = 0; ch < F; ch++) {
              const float f = s->lang[(size_t)id * F + ch];
              accum_rec_F[ch] = last_alpha * last_F[ch] + (1.f - last_alpha) * accum_rec_F[ch];
              last_F[ch] = f;
              const float dL_dchannel_F = dL_dpixel_F[ch];
              dL_dalpha += (f - accum_rec_F[ch]) * dL_dchannel_F;
              double v = (double)(dchannel_dcolor * dL_dchannel_F);
#pragma omp atomic
              a_feat[(size_t)id * F + ch] += v;
            }
          }
          dL_dalpha *= Tt;
          last_alpha = alpha;
          float bg_dot_dpixel = 0;
          for (int i = 0; i < NUM_CHANNELS; i++) bg_dot_dpixel += s->bg[i] * dL_dpixel[i];
          dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

          const float dL_dG = co[3] * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * co[0] - gdy * co[1];
          const float dG_ddely = -gdy * co[2] - gdx * co[1];
          double v0 = (double)(dL_dG * dG_ddelx * ddelx_dx), v1 = (double)(dL_dG * dG_ddely * ddely_dy);
          double c0 = (double)(-0.5f * gdx * dx * dL_dG), c1 = (double)(-0.5f * gdx * dy * dL_dG),
                 c2 = (double)(-0.5f * gdy * dy * dL_dG);
          double o = (double)(G * dL_dalpha);
#pragma omp atomic
          a_mean2D[2 * (size_t)id] += v0;
#pragma omp atomic
          a_mean2D[2 * (size_t)id + 1] += v1;
#pragma omp atomic
          a_conic[3 * (size_t)id] += c0;
#pragma omp atomic
          a_conic[3 * (size_t)id + 1] += c1;
#pragma omp atomic
          a_conic[3 * (size_t)id + 2] += c2;
#pragma omp atomic
          a_opacity[id] += o;
        }
      }
  }
}

/* RAST/cuda_rasterizer/backward.cu:144-274 (computeCov2DCUDA) */
static void cov2D_bwd(const OrcState* s, int idx, const float* cov3D, float h_x, float h_y,
                      const float* dL_dconics, float* dL_dmeans, float* dL_dcov) {
  vec3 mean = v3(s->means3D[3 * idx], s->means3D[3 * idx + 1], s->means3D[3 * idx + 2]);
  float dLc_x = dL_dconics[4 * idx], dLc_y = dL_dconics[4 * idx + 1], dLc_z = dL_dconics[4 * idx + 3];
  vec3 t; mat3 T, Vrk, Wm, cov2D; float txtz, tytz;
  cov2D_core(mean, h_x, h_y, s->tanfovx, s->tanfovy, cov3D, s->view, &t, &T, &Vrk, &Wm, &cov2D, &txtz,
             &tytz);
  const float limx = 1.3f * s->tanfovx, limy = 1.3f * s->tanfovy;
  const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;

  float a = cov2D.m[0][0] += 0.3f;
  float b = cov2D.m[0][1];
  float c = cov2D.m[1][1] += 0.3f;
  float denom = a * c - b * b;
  float dL_da = 0, dL_db = 0, dL_dc = 0;
  float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  float* dc = dL_dcov + 6 * (size_t)idx;
  if (denom2inv != 0) {
    dL_da = denom2inv * (-c * c * dLc_x + 2 * b * c * dLc_y + (denom - a * c) * dLc_z);
    dL_dc = denom2inv * (-a * a * dLc_z + 2 * a * b * dLc_y + (denom - a * c) * dLc_x);
    dL_db = denom2inv * 2 * (b * c * dLc_x - (denom + 2 * b * b) * dLc_y + a * b * dLc_z);
    dc[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
    dc[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
    dc[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
    dc[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db +
            2 * T.m[1][0] * T.m[1][1] * dL_dc;
    dc[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db +
            2 * T.m[1][0] * T.m[1][2] * dL_dc;
    dc[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db +
            2 * T.m[1][1] * T.m[1][2] * dL_dc;
  } else {
    for (int i = 0; i < 6; i++) dc[i] = 0;
  }
  float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                  (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
  float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                  (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
  float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                  (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
  float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                  (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
  float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                  (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
  float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                  (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;
  float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
  float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
  float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
  float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
  float tz = 1.f / t.z;
  float tz2 = tz * tz;
  float tz3 = tz2 * tz;
  float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
  float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
  float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
                 (2 * h_y * t.y) * tz3 * dL_dJ12;
  vec3 dL_dmean = transformVec4x3Transpose(v3(dL_dtx, dL_dty, dL_dtz), s->view);
  dL_dmeans[3 * idx + 0] = dL_dmean.x; /* ASSIGNMENT, backward.cu:273 */
  dL_dmeans[3 * idx + 1] = dL_dmean.y;
  dL_dmeans[3 * idx + 2] = dL_dmean.z;
}

/* RAST/cuda_rasterizer/backward.cu:20-139 (computeColorFromSH bwd) */
static void sh_bwd(const OrcState* s, int idx, const float* dL_dcolor, float* dL_dmeans, float* dL_dshs) {
  const int deg = s->D, max_coeffs = s->M;
  vec3 pos = v3(s->means3D[3 * idx], s->means3D[3 * idx + 1], s->means3D[3 * idx + 2]);
  vec3 dir_orig = v3sub(pos, v3(s->campos[0], s->campos[1], s->campos[2]));
  float len = sqrtf(v3dot(dir_orig, dir_orig));
  vec3 dir = v3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
  const vec3* sh = ((const vec3*)s->shs) + (size_t)idx * max_coeffs;
  vec3 dL_dRGB = v3(dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]);
  dL_dRGB.x *= s->clamped[3 * idx + 0] ? 0 : 1;
  dL_dRGB.y *= s->clamped[3 * idx + 1] ? 0 : 1;
  dL_dRGB.z *= s->clamped[3 * idx + 2] ? 0 : 1;
  vec3 dRGBdx = v3(0, 0, 0), dRGBdy = v3(0, 0, 0), dRGBdz = v3(0, 0, 0);
  float x = dir.x, y = dir.y, z = dir.z;
  vec3* dL_dsh = ((vec3*)dL_dshs) + (size_t)idx * max_coeffs;
  dL_dsh[0] = v3scale(SH_C0, dL_dRGB);
  if (deg > 0) {
    dL_dsh[1] = v3scale(-SH_C1 * y, dL_dRGB);
    dL_dsh[2] = v3scale(SH_C1 * z, dL_dRGB);
    dL_dsh[3] = v3scale(-SH_C1 * x, dL_dRGB);
    dRGBdx = v3scale(-SH_C1, sh[3]);
    dRGBdy = v3scale(-SH_C1, sh[1]);
    dRGBdz = v3scale(SH_C1, sh[2]);
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z;
      float xy = x * y, yz = y * z, xz = x * z;
      dL_dsh[4] = v3scale(SH_C2[0] * xy, dL_dRGB);
      dL_dsh[5] = v3scale(SH_C2[1] * yz, dL_dRGB);
      dL_dsh[6] = v3scale(SH_C2[2] * (2.f * zz - xx - yy), dL_dRGB);
      dL_dsh[7] = v3scale(SH_C2[3] * xz, dL_dRGB);
      dL_dsh[8] = v3scale(SH_C2[4] * (xx - yy), dL_dRGB);
      dRGBdx = v3add(dRGBdx, v3add(v3add(v3scale(SH_C2[0] * y, sh[4]), v3scale(SH_C2[2] * 2.f * -x, sh[6])),
                                   v3add(v3scale(SH_C2[3] * z, sh[7]), v3scale(SH_C2[4] * 2.f * x, sh[8]))));
      dRGBdy = v3add(dRGBdy, v3add(v3add(v3scale(SH_C2[0] * x, sh[4]), v3scale(SH_C2[1] * z, sh[5])),
                                   v3add(v3scale(SH_C2[2] * 2.f * -y, sh[6]), v3scale(SH_C2[4] * 2.f * -y, sh[8]))));
      dRGBdz = v3add(dRGBdz, v3add(v3add(v3scale(SH_C2[1] * y, sh[5]), v3scale(SH_C2[2] * 2.f * 2.f * z, sh[6])),
                                   v3scale(SH_C2[3] * x, sh[7])));
      if (deg > 2) {
        dL_dsh[9] = v3scale(SH_C3[0] * y * (3.f * xx - yy), dL_dRGB);
        dL_dsh[10] = v3scale(SH_C3[1] * xy * z, dL_dRGB);
        dL_dsh[11] = v3scale(SH_C3[2] * y * (4.f * zz - xx - yy), dL_dRGB);
        dL_dsh[12] = v3scale(SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), dL_dRGB);
        dL_dsh[13] = v3scale(SH_C3[4] * x * (4.f * zz - xx - yy), dL_dRGB);
        dL_dsh[14] = v3scale(SH_C3[5] * z * (xx - yy), dL_dRGB);
        dL_dsh[15] = v3scale(SH_C3[6] * x * (xx - 3.f * yy), dL_dRGB);
        vec3 ax = v3(0, 0, 0), ay = v3(0, 0, 0), az = v3(0, 0, 0);
        ax = v3add(ax, v3scale(SH_C3[0] * 3.f * 2.f * xy, sh[9]));
        ax = v3add(ax, v3scale(SH_C3[1] * yz, sh[10]));
        ax = v3add(ax, v3scale(SH_C3[2] * -2.f * xy, sh[11]));
        ax = v3add(ax, v3scale(SH_C3[3] * -3.f * 2.f * xz, sh[12]));
        ax = v3add(ax, v3scale(SH_C3[4] * (-3.f * xx + 4.f * zz - yy), sh[13]));
        ax = v3add(ax, v3scale(SH_C3[5] * 2.f * xz, sh[14]));
        ax = v3add(ax, v3scale(SH_C3[6] * 3.f * (xx - yy), sh[15]));
        ay = v3add(ay, v3scale(SH_C3[0] * 3.f * (xx - yy), sh[9]));
        ay = v3add(ay, v3scale(SH_C3[1] * xz, sh[10]));
        ay = v3add(ay, v3scale(SH_C3[2] * (-3.f * yy + 4.f * zz - xx), sh[11]));
        ay = v3add(ay, v3scale(SH_C3[3] * -3.f * 2.f * yz, sh[12]));
        ay = v3add(ay, v3scale(SH_C3[4] * -2.f * xy, sh[13]));
        ay = v3add(ay, v3scale(SH_C3[5] * -2.f * yz, sh[14]));
        ay = v3add(ay, v3scale(SH_C3[6] * -3.f * 2.f * xy, sh[15]));
        az = v3add(az, v3scale(SH_C3[1] * xy, sh[10]));
        az = v3add(az, v3scale(SH_C3[2] * 4.f * 2.f * yz, sh[11]));
        az = v3add(az, v3scale(SH_C3[3] * 3.f * (2.f * zz - xx - yy), sh[12]));
        az = v3add(az, v3scale(SH_C3[4] * 4.f * 2.f * xz, sh[13]));
        az = v3add(az, v3scale(SH_C3[5] * (xx - yy), sh[14]));
        dRGBdx = v3add(dRGBdx, ax);
        dRGBdy = v3add(dRGBdy, ay);
        dRGBdz = v3add(dRGBdz, az);
      }
    }
  }
  vec3 dL_ddir = v3(v3dot(dRGBdx, dL_dRGB), v3dot(dRGBdy, dL_dRGB), v3dot(dRGBdz, dL_dRGB));
  vec3 dL_dmean = dnormvdv(dir_orig, dL_ddir);
  dL_dmeans[3 * idx + 0] += dL_dmean.x;
  dL_dmeans[3 * idx + 1] += dL_dmean.y;
  dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

/* RAST/cuda_rasterizer/backward.cu:278-341 (computeCov3D bwd) */
static void cov3D_bwd(int idx, vec3 scale, float mod, const float* rot, const float* dL_dcov3Ds,
                      float* dL_dscales, float* dL_drots) {
  float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
  mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                     2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                     2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
  mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
  vec3 sv = v3scale(mod, scale);
  S.m[0][0] = sv.x; S.m[1][1] = sv.y; S.m[2][2] = sv.z;
  mat3 Mm = mat3_mul(S, R);
  const float* d = dL_dcov3Ds + 6 * (size_t)idx;
  mat3 dL_dSigma = mat3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2],
                             0.5f * d[4], d[5]);
  mat3 dL_dM = mat3_mul(mat3_scale(2.0f, Mm), dL_dSigma);
  mat3 Rt = mat3_T(R);
  mat3 dL_dMt = mat3_T(dL_dM);
  dL_dscales[3 * idx + 0] = Rt.m[0][0] * dL_dMt.m[0][0] + Rt.m[0][1] * dL_dMt.m[0][1] + Rt.m[0][2] * dL_dMt.m[0][2];
  dL_dscales[3 * idx + 1] = Rt.m[1][0] * dL_dMt.m[1][0] + Rt.m[1][1] * dL_dMt.m[1][1] + Rt.m[1][2] * dL_dMt.m[1][2];
  dL_dscales[3 * idx + 2] = Rt.m[2][0] * dL_dMt.m[2][0] + Rt.m[2][1] * dL_dMt.m[2][1] + Rt.m[2][2] * dL_dMt.m[2][2];
  for (int k = 0; k < 3; k++) {
    dL_dMt.m[0][k] *= sv.x;
    dL_dMt.m[1][k] *= sv.y;
    dL_dMt.m[2][k] *= sv.z;
  }
  float (*q)[3] = dL_dMt.m;
  dL_drots[4 * idx + 0] = 2 * z * (q[0][1] - q[1][0]) + 2 * y * (q[2][0] - q[0][2]) + 2 * x * (q[1][2] - q[2][1]);
  dL_drots[4 * idx + 1] = 2 * y * (q[1][0] + q[0][1]) + 2 * z * (q[2][0] + q[0][2]) + 2 * r * (q[1][2] - q[2][1]) -
                          4 * x * (q[2][2] + q[1][1]);
  dL_drots[4 * idx + 2] = 2 * x * (q[1][0] + q[0][1]) + 2 * r * (q[2][0] - q[0][2]) + 2 * z * (q[1][2] + q[2][1]) -
                          4 * y * (q[2][2] + q[0][0]);
  dL_drots[4 * idx + 3] = 2 * r * (q[0][1] - q[1][0]) + 2 * x * (q[2][0] + q[0][2]) + 2 * y * (q[1][2] + q[2][1]) -
                          4 * z * (q[1][1] + q[0][0]);
  /* no normalisation Jacobian: backward.cu:340 */
}

/* RAST/cuda_rasterizer/rasterizer_impl.cu:359-463 (Rasterizer::backward) +
 * RAST/rasterize_points.cu:130-225 (all ten gradient tensors zero-initialised, shapes:
 * dL_dmean2D [P,3], dL_dconic [P,2,2], dL_dopacity [P,1], dL_dcolor [P,3], dL_dfeat [P,F],
 * dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3], dL_dscale [P,3], dL_drot [P,4]). */
void orc_backward(const OrcState* s, const float* dL_dpix, const float* dL_dpix_F, float* dL_dmean2D,
                  float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dfeat,
                  float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot) {
  const int P = s->P, F = s->F;
  memset(dL_dmean2D, 0, (size_t)P * 3 * 4);
  memset(dL_dconic, 0, (size_t)P * 4 * 4);
  memset(dL_dopacity, 0, (size_t)P * 4);
  memset(dL_dcolor, 0, (size_t)P * 3 * 4);
  if (s->include_feature) memset(dL_dfeat, 0, (size_t)P * F * 4);
  memset(dL_dmean3D, 0, (size_t)P * 3 * 4);
  memset(dL_dcov3D, 0, (size_t)P * 6 * 4);
  if (s->M > 0) memset(dL_dsh, 0, (size_t)P * s->M * 3 * 4);
  memset(dL_dscale, 0, (size_t)P * 3 * 4);
  memset(dL_drot, 0, (size_t)P * 4 * 4);
  if (P == 0) return;

  double* a_mean2D = (double*)calloc((size_t)P * 2, 8);
  double* a_conic = (double*)calloc((size_t)P * 3, 8);
  double* a_opacity = (double*)calloc((size_t)P, 8);
  double* a_colors = (double*)calloc((size_t)P * 3, 8);
  double* a_feat = (double*)calloc((size_t)P * (F > 0 ? F : 1), 8);
  render_bwd(s, dL_dpix, dL_dpix_F, a_mean2D, a_conic, a_opacity, a_colors, a_feat);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    dL_dmean2D[3 * i] = (float)a_mean2D[2 * i];
    dL_dmean2D[3 * i + 1] = (float)a_mean2D[2 * i + 1];
    dL_dconic[4 * i] = (float)a_conic[3 * i];
    dL_dconic[4 * i + 1] = (float)a_conic[3 * i + 1];
    dL_dconic[4 * i + 3] = (float)a_conic[3 * i + 2]; /* .w; .z never written, backward.cu:585-587 */
    dL_dopacity[i] = (float)a_opacity[i];
    for (int c = 0; c < 3; c++) dL_dcolor[3 * i + c] = (float)a_colors[3 * i + c];
    if (s->include_feature)
      for (int c = 0; c < F; c++) dL_dfeat[(size_t)i * F + c] = (float)a_feat[(size_t)i * F + c];
  }
  free(a_mean2D); free(a_conic); free(a_opacity); free(a_colors); free(a_feat);

  const float focal_y = s->H / (2.0f * s->tanfovy);
  const float focal_x = s->W / (2.0f * s->tanfovx);
  const float* cov3D_ptr = s->cov3D_precomp ? s->cov3D_precomp : s->cov3D;
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    if (!(s->radii[idx] > 0)) continue;
    /* K9: backward.cu:623-635 */
    cov2D_bwd(s, idx, cov3D_ptr + 6 * (size_t)idx, focal_x, focal_y, dL_dconic, dL_dmean3D, dL_dcov3D);
    /* K10: backward.cu:346-396 */
    vec3 m = v3(s->means3D[3 * idx], s->means3D[3 * idx + 1], s->means3D[3 * idx + 2]);
    const float* proj = s->proj;
    float m_hom[4];
    transformPoint4x4(m, proj, m_hom);
    float m_w = 1.0f / (m_hom[3] + 0.0000001f);
    float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
    dL_dmean3D[3 * idx + 0] += (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
    dL_dmean3D[3 * idx + 1] += (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
    dL_dmean3D[3 * idx + 2] += (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
    if (s->shs) sh_bwd(s, idx, dL_dcolor, dL_dmean3D, dL_dsh);
    if (s->scales)
      cov3D_bwd(idx, v3(s->scales[3 * idx], s->scales[3 * idx + 1], s->scales[3 * idx + 2]),
                s->scale_modifier, s->rotations + 4 * (size_t)idx, dL_dcov3D, dL_dscale, dL_drot);
  }
}

/* RAST/cuda_rasterizer/rasterizer_impl.cu:54-66,141-153 (checkFrustum / markVisible) */
void orc_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present) {
  for (int idx = 0; idx < P; idx++) {
    vec3 pv; int trap = 0;
    present[idx] = (uint8_t)in_frustum(idx, means3D, view, proj, 0, &pv, &trap);
  }
}

/* RAST/cuda_rasterizer/rasterizer_impl.cu:35-50 */
uint32_t orc_get_higher_msb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4;
  uint32_t step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

/* ---- accessors for stage-level parity tests ---- */
int orc_num_rendered(const OrcState* s) { return s->R; }
const float* orc_depths(const OrcState* s) { return s->depths; }
const float* orc_means2D(const OrcState* s) { return s->means2D; }
const float* orc_conic_opacity(const OrcState* s) { return s->conic_opacity; }
const float* orc_rgb(const OrcState* s) { return s->rgb; }
const float* orc_cov3D(const OrcState* s) { return s->cov3D; }
const uint8_t* orc_clamped(const OrcState* s) { return s->clamped; }
const uint32_t* orc_tiles_touched(const OrcState* s) { return s->tiles_touched; }
const uint32_t* orc_point_list(const OrcState* s) { return s->point_list; }
const uint64_t* orc_keys(const OrcState* s) { return s->keys; }
const uint32_t* orc_ranges(const OrcState* s) { return s->ranges; }
const float* orc_final_T(const OrcState* s) { return s->final_T; }
const uint32_t* orc_n_contrib(const OrcState* s) { return s->n_contrib; }

/* ------------------------------------------------------------------------------------------ */
/* Deformation-field apply epilogue: agents/manigaussian_bc/models_embed.py:297-304           */
/*   next.xyz = xyz.detach() + dxyz ; next.rot = F.normalize(rot.detach() + drot, dim=-1)      */
/* F.normalize: v / max(||v||_2, eps), eps = 1e-12 (torch.nn.functional.normalize default).    */
/* ------------------------------------------------------------------------------------------ */
void orc_deform_apply_fwd(int N, const float* xyz, const float* rot, const float* delta /*[N,7]*/,
                          float* xyz_out, float* rot_out) {
  for (int i = 0; i < N; i++) {
    for (int c = 0; c < 3; c++) xyz_out[3 * i + c] = xyz[3 * i + c] + delta[7 * i + c];
    float q[4], n2 = 0;
    for (int c = 0; c < 4; c++) { q[c] = rot[4 * i + c] + delta[7 * i + 3 + c]; n2 += q[c] * q[c]; }
    float n = fmaxf(sqrtf(n2), 1e-12f);
    for (int c = 0; c < 4; c++) rot_out[4 * i + c] = q[c] / n;
  }
}
/* gradient flows only into delta (xyz, rot are detached) */
void orc_deform_apply_bwd(int N, const float* rot, const float* delta, const float* g_xyz, const float* g_rot,
                          float* g_delta) {
  for (int i = 0; i < N; i++) {
    for (int c = 0; c < 3; c++) g_delta[7 * i + c] = g_xyz[3 * i + c];
    float q[4], n2 = 0, dot = 0;
    for (int c = 0; c < 4; c++) { q[c] = rot[4 * i + c] + delta[7 * i + 3 + c]; n2 += q[c] * q[c]; }
    float n = sqrtf(n2);
    if (n > 1e-12f) {
      for (int c = 0; c < 4; c++) dot += q[c] * g_rot[4 * i + c];
      for (int c = 0; c < 4; c++) g_delta[7 * i + 3 + c] = g_rot[4 * i + c] / n - q[c] * dot / (n * n * n);
    } else {
      for (int c = 0; c < 4; c++) g_delta[7 * i + 3 + c] = g_rot[4 * i + c] / 1e-12f;
    }
  }
}
