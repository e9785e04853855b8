// oracle/ref_wrapper.cu -- TEST INFRASTRUCTURE ONLY.  C entry point around the REFERENCE's own rasterizer
// (CudaRasterizer::Rasterizer::forward / ::backward, RAST/cuda_rasterizer/rasterizer.h:20-92), compiled unmodified from
// /root/reference with hipcc through oracle/refshim (see oracle/Makefile, target _ref).  Host arrays in, host arrays
// out; one forward + one backward per call.  The reference build fixes NUM_CHANNELS = NUM_CHANNELS_language_feature = 3
// (RAST/cuda_rasterizer/config.h:15-16).
#include <functional>
#include <vector>
#include "cuda_runtime.h"
#include "rasterizer.h"
#include "config.h"

namespace {
struct DevBuf {
  char* p = nullptr; size_t n = 0;
  char* resize(size_t bytes) { if (bytes > n) { if (p) hipFree(p); hipMalloc(&p, bytes); n = bytes; } return p; }
  ~DevBuf() { if (p) hipFree(p); }
};
template <typename T>
T* up(const T* h, size_t count, std::vector<void*>& pool) {
  if (!h || count == 0) return nullptr;
  T* d = nullptr;
  hipMalloc(&d, count * sizeof(T));
  hipMemcpy(d, h, count * sizeof(T), hipMemcpyHostToDevice);
  pool.push_back(d);
  return d;
}
template <typename T>
T* zeros(size_t count, std::vector<void*>& pool) {
  T* d = nullptr;
  hipMalloc(&d, (count ? count : 1) * sizeof(T));
  hipMemset(d, 0, (count ? count : 1) * sizeof(T));
  pool.push_back(d);
  return d;
}
template <typename T>
void down(T* h, const T* d, size_t count) { if (h && count) hipMemcpy(h, d, count * sizeof(T), hipMemcpyDeviceToHost); }
}  // namespace

extern "C" int ref_num_feature_channels() { return NUM_CHANNELS_language_feature; }

// returns num_rendered (< 0 on error)
extern "C" int ref_forward_backward(
    int P, int D, int M, int W, int H, const float* bg, const float* means3D, const float* shs, const float* colors_precomp,
    const float* language_feature, const float* opacities, const float* scales, float scale_modifier, const float* rotations,
    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, float tanfovx,
    float tanfovy, int include_feature, const float* dL_dcolor_px, const float* dL_dfeat_px,
    /* outputs (host) */
    float* out_color, float* out_feat, int* radii, float* dL_dmeans2D, float* dL_dopacity, float* dL_dcolors,
    float* dL_dfeature, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations) {
  try {
    const int F = NUM_CHANNELS_language_feature;
    std::vector<void*> pool;
    const size_t N = (size_t)W * H;
    float* d_bg = up(bg, 3, pool);
    float* d_means = up(means3D, 3 * (size_t)P, pool);
    float* d_sh = up(shs, 3 * (size_t)M * P, pool);
    float* d_col = up(colors_precomp, 3 * (size_t)P, pool);
    float* d_feat = up(language_feature, (size_t)F * P, pool);
    float* d_op = up(opacities, P, pool);
    float* d_sc = up(scales, 3 * (size_t)P, pool);
    float* d_rot = up(rotations, 4 * (size_t)P, pool);
    float* d_cov = up(cov3D_precomp, 6 * (size_t)P, pool);
    float* d_vm = up(viewmatrix, 16, pool);
    float* d_pm = up(projmatrix, 16, pool);
    float* d_cam = up(campos, 3, pool);
    float* d_oc = zeros<float>(3 * N, pool);
    float* d_of = zeros<float>(F * N, pool);
    int* d_radii = zeros<int>(P, pool);
    DevBuf geom, binning, img;
    std::function<char*(size_t)> fg = [&](size_t n) { return geom.resize(n); };
    std::function<char*(size_t)> fb = [&](size_t n) { return binning.resize(n); };
    std::function<char*(size_t)> fi = [&](size_t n) { return img.resize(n); };
    int R = 0;
    if (P > 0)
      R = CudaRasterizer::Rasterizer::forward(fg, fb, fi, P, D, M, d_bg, W, H, d_means, d_sh, d_col, d_feat, d_op, d_sc,
                                              scale_modifier, d_rot, d_cov, d_vm, d_pm, d_cam, tanfovx, tanfovy, false, d_oc,
                                              d_of, d_radii, false, include_feature != 0);
    hipDeviceSynchronize();
    down(out_color, d_oc, 3 * N);
    down(out_feat, d_of, F * N);
    down(radii, d_radii, P);
    if (dL_dcolor_px && P > 0) {
      float* d_gpx = up(dL_dcolor_px, 3 * N, pool);
      float* d_gfx = dL_dfeat_px ? up(dL_dfeat_px, F * N, pool) : zeros<float>(F * N, pool);
      float* g_m2d = zeros<float>(3 * (size_t)P, pool);
      float* g_conic = zeros<float>(4 * (size_t)P, pool);
      float* g_op = zeros<float>(P, pool);
      float* g_col = zeros<float>(3 * (size_t)P, pool);
      float* g_feat = zeros<float>((size_t)F * P, pool);
      float* g_m3d = zeros<float>(3 * (size_t)P, pool);
      float* g_cov = zeros<float>(6 * (size_t)P, pool);
      float* g_sh = zeros<float>(3 * (size_t)M * P, pool);
      float* g_sc = zeros<float>(3 * (size_t)P, pool);
      float* g_rot = zeros<float>(4 * (size_t)P, pool);
      CudaRasterizer::Rasterizer::backward(P, D, M, R, d_bg, W, H, d_means, d_sh, d_col, d_feat, d_sc, scale_modifier, d_rot,
                                           d_cov, d_vm, d_pm, d_cam, tanfovx, tanfovy, d_radii, geom.p, binning.p, img.p,
                                           d_gpx, d_gfx, g_m2d, g_conic, g_op, g_col, g_feat, g_m3d, g_cov, g_sh, g_sc, g_rot,
                                           false, include_feature != 0);
      hipDeviceSynchronize();
      down(dL_dmeans2D, g_m2d, 3 * (size_t)P); down(dL_dopacity, g_op, P); down(dL_dcolors, g_col, 3 * (size_t)P);
      down(dL_dfeature, g_feat, (size_t)F * P); down(dL_dmeans3D, g_m3d, 3 * (size_t)P); down(dL_dcov3D, g_cov, 6 * (size_t)P);
      down(dL_dsh, g_sh, 3 * (size_t)M * P); down(dL_dscales, g_sc, 3 * (size_t)P); down(dL_drotations, g_rot, 4 * (size_t)P);
    }
    for (void* p : pool) hipFree(p);
    return hipGetLastError() == hipSuccess ? R : -2;
  } catch (const std::exception& e) {
    fprintf(stderr, "ref_forward_backward: %s\n", e.what());
    return -1;
  }
}
