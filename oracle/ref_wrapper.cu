// oracle/ref_wrapper.cu -- TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE's own rasterizer
// (CudaRasterizer::Rasterizer::forward / ::backward, RAST/cuda_rasterizer/rasterizer.h:20-92), compiled unmodified from
// /root/reference with hipcc through oracle/refshim (see oracle/Makefile, target _ref).  Host arrays in, host arrays
// out.  The stock reference build fixes NUM_CHANNELS = NUM_CHANNELS_language_feature = 3 (RAST/cuda_rasterizer/
// config.h:15-16); the _f32 variant of the library is the same sources with the feature width set to 32 the way the
// reference prescribes (a recompile with another NUM_CHANNELS_language_feature).
#include <functional>
#include <vector>
#include "cuda_runtime.h"
#include "rasterizer.h"
#include "rasterizer_impl.h"
#include "config.h"

namespace {
struct DevBuf {
  char* p = nullptr; size_t n = 0;
  char* resize(size_t bytes) { if (bytes > n) { if (p) hipFree(p); hipMalloc(&p, bytes); n = bytes; } return p; }
  ~DevBuf() { if (p) hipFree(p); }
};

struct Job {
  static constexpr int F = NUM_CHANNELS_language_feature;
  std::vector<void*> pool;
  int P, D, M, W, H, inc, R = 0;
  size_t N;
  float scale_modifier, tanfovx, tanfovy;
  float *bg, *means, *sh, *col, *feat, *op, *sc, *rot, *cov, *vm, *pm, *cam, *oc, *of, *gpx = nullptr, *gfx = nullptr;
  int* radii;
  float *g_m2d, *g_conic, *g_op, *g_col, *g_feat, *g_m3d, *g_cov, *g_sh, *g_sc, *g_rot;
  DevBuf geom, binning, img;

  template <typename T> T* up(const T* h, size_t count) {
    if (!h || count == 0) return nullptr;
    T* d = nullptr;
    hipMalloc(&d, count * sizeof(T));
    hipMemcpy(d, h, count * sizeof(T), hipMemcpyHostToDevice);
    pool.push_back(d);
    return d;
  }
  template <typename T> T* zeros(size_t count) {
    T* d = nullptr;
    hipMalloc(&d, (count ? count : 1) * sizeof(T));
    hipMemset(d, 0, (count ? count : 1) * sizeof(T));
    pool.push_back(d);
    return d;
  }
  ~Job() { for (void* p : pool) hipFree(p); }

  int forward() {
    std::function<char*(size_t)> fg = [&](size_t n) { return geom.resize(n); };
    std::function<char*(size_t)> fb = [&](size_t n) { return binning.resize(n); };
    std::function<char*(size_t)> fi = [&](size_t n) { return img.resize(n); };
    // rasterize_points.cu:66-68 (torch::full 0 for the images and radii)
    hipMemsetAsync(oc, 0, 3 * N * sizeof(float)); hipMemsetAsync(of, 0, F * N * sizeof(float));
    hipMemsetAsync(radii, 0, (P ? P : 1) * sizeof(int));
    R = 0;
    if (P > 0)
      R = CudaRasterizer::Rasterizer::forward(fg, fb, fi, P, D, M, bg, W, H, means, sh, col, feat, op, sc, scale_modifier,
                                              rot, cov, vm, pm, cam, tanfovx, tanfovy, false, oc, of, radii, false, inc != 0);
    return R;
  }
  void alloc_grads() {
    g_m2d = zeros<float>(3 * (size_t)P); g_conic = zeros<float>(4 * (size_t)P); g_op = zeros<float>(P);
    g_col = zeros<float>(3 * (size_t)P); g_feat = zeros<float>((size_t)F * P); g_m3d = zeros<float>(3 * (size_t)P);
    g_cov = zeros<float>(6 * (size_t)P); g_sh = zeros<float>(3 * (size_t)M * P); g_sc = zeros<float>(3 * (size_t)P);
    g_rot = zeros<float>(4 * (size_t)P);
  }
  void backward() {
    // rasterize_points.cu:167-184 (torch::zeros for every gradient)
    auto z = [](float* p, size_t n) { hipMemsetAsync(p, 0, (n ? n : 1) * sizeof(float)); };
    z(g_m2d, 3 * (size_t)P); z(g_conic, 4 * (size_t)P); z(g_op, P); z(g_col, 3 * (size_t)P); z(g_feat, (size_t)F * P);
    z(g_m3d, 3 * (size_t)P); z(g_cov, 6 * (size_t)P); z(g_sh, 3 * (size_t)M * P); z(g_sc, 3 * (size_t)P);
    z(g_rot, 4 * (size_t)P);
    if (P > 0)
      CudaRasterizer::Rasterizer::backward(P, D, M, R, bg, W, H, means, sh, col, feat, sc, scale_modifier, rot, cov, vm, pm,
                                           cam, tanfovx, tanfovy, radii, geom.p, binning.p, img.p, gpx, gfx, g_m2d, g_conic,
                                           g_op, g_col, g_feat, g_m3d, g_cov, g_sh, g_sc, g_rot, false, inc != 0);
  }
};

template <typename T>
void down(T* h, const T* d, size_t count) { if (h && count) hipMemcpy(h, d, count * sizeof(T), hipMemcpyDeviceToHost); }

void setup(Job& j, int P, int D, int M, int W, int H, const float* bg, const float* means3D, const float* shs,
           const float* colors_precomp, const float* language_feature, const float* opacities, const float* scales,
           float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
           const float* projmatrix, const float* campos, float tanfovx, float tanfovy, int include_feature,
           const float* dL_dcolor_px, const float* dL_dfeat_px) {
  const int F = Job::F;
  j.P = P; j.D = D; j.M = M; j.W = W; j.H = H; j.inc = include_feature; j.N = (size_t)W * H;
  j.scale_modifier = scale_modifier; j.tanfovx = tanfovx; j.tanfovy = tanfovy;
  j.bg = j.up(bg, 3); j.means = j.up(means3D, 3 * (size_t)P); j.sh = j.up(shs, 3 * (size_t)M * P);
  j.col = j.up(colors_precomp, 3 * (size_t)P); j.feat = j.up(language_feature, (size_t)F * P); j.op = j.up(opacities, P);
  j.sc = j.up(scales, 3 * (size_t)P); j.rot = j.up(rotations, 4 * (size_t)P); j.cov = j.up(cov3D_precomp, 6 * (size_t)P);
  j.vm = j.up(viewmatrix, 16); j.pm = j.up(projmatrix, 16); j.cam = j.up(campos, 3);
  j.oc = j.zeros<float>(3 * j.N); j.of = j.zeros<float>(F * j.N); j.radii = j.zeros<int>(P);
  if (dL_dcolor_px) {
    j.gpx = j.up(dL_dcolor_px, 3 * j.N);
    j.gfx = dL_dfeat_px ? j.up(dL_dfeat_px, F * j.N) : j.zeros<float>(F * j.N);
    j.alloc_grads();
  }
}
}  // namespace

extern "C" int ref_num_feature_channels() { return NUM_CHANNELS_language_feature; }

#define REF_INPUTS                                                                                                       \
  int P, int D, int M, int W, int H, const float *bg, const float *means3D, const float *shs,                            \
      const float *colors_precomp, const float *language_feature, const float *opacities, const float *scales,           \
      float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *viewmatrix,                 \
      const float *projmatrix, const float *campos, float tanfovx, float tanfovy, int include_feature,                   \
      const float *dL_dcolor_px, const float *dL_dfeat_px
#define REF_INPUT_NAMES                                                                                                  \
  P, D, M, W, H, bg, means3D, shs, colors_precomp, language_feature, opacities, scales, scale_modifier, rotations,      \
      cov3D_precomp, viewmatrix, projmatrix, campos, tanfovx, tanfovy, include_feature, dL_dcolor_px, dL_dfeat_px

// One forward + (if dL_dcolor_px) one backward.  Returns num_rendered (< 0 on error).
extern "C" int ref_forward_backward(REF_INPUTS,
                                    /* outputs (host) */
                                    float* out_color, float* out_feat, int* radii, float* dL_dmeans2D, float* dL_dopacity,
                                    float* dL_dcolors, float* dL_dfeature, float* dL_dmeans3D, float* dL_dcov3D,
                                    float* dL_dsh, float* dL_dscales, float* dL_drotations) {
  try {
    const int F = Job::F;
    Job j;
    setup(j, REF_INPUT_NAMES);
    int R = j.forward();
    hipDeviceSynchronize();
    down(out_color, j.oc, 3 * j.N); down(out_feat, j.of, F * j.N); down(radii, j.radii, P);
    if (dL_dcolor_px && P > 0) {
      j.backward();
      hipDeviceSynchronize();
      down(dL_dmeans2D, j.g_m2d, 3 * (size_t)P); down(dL_dopacity, j.g_op, P); down(dL_dcolors, j.g_col, 3 * (size_t)P);
      down(dL_dfeature, j.g_feat, (size_t)F * P); down(dL_dmeans3D, j.g_m3d, 3 * (size_t)P);
      down(dL_dcov3D, j.g_cov, 6 * (size_t)P); down(dL_dsh, j.g_sh, 3 * (size_t)M * P);
      down(dL_dscales, j.g_sc, 3 * (size_t)P); down(dL_drotations, j.g_rot, 4 * (size_t)P);
    }
    return hipGetLastError() == hipSuccess ? R : -2;
  } catch (const std::exception& e) {
    fprintf(stderr, "ref_forward_backward: %s\n", e.what());
    return -1;
  }
}

// One forward; returns what the reference's preprocess left in its GeometryState (RAST/cuda_rasterizer/rasterizer_impl.h:
// 30-46): means2D [P,2], conic_opacity [P,4], depths [P], rgb [P,3], cov3D [P,6] (+ radii).  Rows of Gaussians the preprocess
// returned early for hold whatever the buffer held (compare only where radii > 0).  Returns num_rendered.
extern "C" int ref_forward_geometry(REF_INPUTS, int* radii, float* means2D, float* conic_opacity, float* depths, float* rgb,
                                    float* cov3D) {
  try {
    Job j;
    setup(j, REF_INPUT_NAMES);
    int R = j.forward();
    hipDeviceSynchronize();
    if (P > 0) {
      char* chunk = j.geom.p;
      CudaRasterizer::GeometryState g = CudaRasterizer::GeometryState::fromChunk(chunk, P);
      down(radii, j.radii, P);
      down(means2D, reinterpret_cast<const float*>(g.means2D), 2 * (size_t)P);
      down(conic_opacity, reinterpret_cast<const float*>(g.conic_opacity), 4 * (size_t)P);
      down(depths, g.depths, P); down(rgb, g.rgb, 3 * (size_t)P); down(cov3D, g.cov3D, 6 * (size_t)P);
    }
    return hipGetLastError() == hipSuccess ? R : -2;
  } catch (const std::exception& e) {
    fprintf(stderr, "ref_forward_geometry: %s\n", e.what());
    return -1;
  }
}

// Timing of the reference kernels on this GPU, inputs resident: `warmup` untimed then `iters` timed forward+backward
// passes (image/radii/gradient zero-fills included, as rasterize_points.cu does them per call).  ms_total = hipEvent time
// around the `iters` passes run back to back; ms_fwd / ms_bwd = a second set of `iters` passes timed per half, summed.
extern "C" int ref_bench(REF_INPUTS, int warmup, int iters, float* ms_total, float* ms_fwd, float* ms_bwd) {
  try {
    Job j;
    setup(j, REF_INPUT_NAMES);
    if (!dL_dcolor_px) return -3;
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (int i = 0; i < warmup; i++) { j.forward(); j.backward(); }
    hipDeviceSynchronize();
    float tf = 0.f, tb = 0.f, tt = 0.f;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) { j.forward(); j.backward(); }   // no host sync beyond the reference's own
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&tt, a, b);
    for (int i = 0; i < iters; i++) {
      hipEventRecord(e0);
      j.forward();
      hipEventRecord(e1);
      j.backward();
      hipEventRecord(e2);
      hipEventSynchronize(e2);
      float x;
      hipEventElapsedTime(&x, e0, e1); tf += x;
      hipEventElapsedTime(&x, e1, e2); tb += x;
    }
    *ms_total = tt; *ms_fwd = tf; *ms_bwd = tb;
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2); hipEventDestroy(a); hipEventDestroy(b);
    return hipGetLastError() == hipSuccess ? j.R : -2;
  } catch (const std::exception& e) {
    fprintf(stderr, "ref_bench: %s\n", e.what());
    return -1;
  }
}
