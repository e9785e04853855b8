// mgs_regress.hip -- Gaussian-regressor epilogue: the step right before the rasterizer.
// Reference: agents/manigaussian_bc/models_embed.py:233-253 (split of the 26-vector, exp + clamp_max(0.05) scale,
// sigmoid opacity, normalised rotation, SH [N,4,3] assembly, xyz + delta) and gaussian_renderer/__init__.py:66-68
// (feature / (|feature| + 1e-12)).  The reference runs ~10 small torch kernels + 3 cats here; this is ONE streaming
// pass forward and one backward (HBM-bound: 116 B read, 128 B written per point).
//
// raw row layout (models_embed.py:121,139-141): xyz 3 | opacity 1 | scale 3 | rot 4 | f_dc 3 | feature 3 | f_rest 9.
#include "mgs_common.h"

namespace mgs {

constexpr int RAW = 26;
constexpr float SCALE_MAX = 0.05f;      // models_embed.py:246
constexpr float NORM_EPS = 1e-12f;      // F.normalize eps / MIN_DENOMINATOR

__global__ void __launch_bounds__(256) regress_epilogue_fwd_kernel(int N, const float* __restrict__ raw,
                                                                   const float* __restrict__ xyz_in,
                                                                   float* __restrict__ xyz, float* __restrict__ opacity,
                                                                   float* __restrict__ scale, float* __restrict__ rot,
                                                                   float* __restrict__ sh, float* __restrict__ feature,
                                                                   float* __restrict__ feature_n) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* r = raw + (size_t)RAW * n;
  float v[RAW];
#pragma unroll
  for (int i = 0; i < RAW; i++) v[i] = r[i];
  const size_t i3 = 3 * (size_t)n;
#pragma unroll
  for (int c = 0; c < 3; c++) xyz[i3 + c] = xyz_in[i3 + c] + v[c];
  opacity[n] = 1.0f / (1.0f + expf(-v[3]));
#pragma unroll
  for (int c = 0; c < 3; c++) scale[i3 + c] = fminf(expf(v[4 + c]), SCALE_MAX);
  {
    const float nr = sqrtf(v[7] * v[7] + v[8] * v[8] + v[9] * v[9] + v[10] * v[10]);
    const float inv = 1.0f / fmaxf(nr, NORM_EPS);
#pragma unroll
    for (int c = 0; c < 4; c++) rot[4 * (size_t)n + c] = v[7 + c] * inv;
  }
  float* s = sh + 12 * (size_t)n;  // [4][3]: f_dc then the three f_rest coefficients
#pragma unroll
  for (int c = 0; c < 3; c++) s[c] = v[11 + c];
#pragma unroll
  for (int c = 0; c < 9; c++) s[3 + c] = v[17 + c];
  const float nf = sqrtf(v[14] * v[14] + v[15] * v[15] + v[16] * v[16]);
  const float invf = 1.0f / (nf + NORM_EPS);
#pragma unroll
  for (int c = 0; c < 3; c++) { feature[i3 + c] = v[14 + c]; feature_n[i3 + c] = v[14 + c] * invf; }
}

// g_feature (un-normalised output) and g_feature_n may be NULL (no gradient through that output).
__global__ void __launch_bounds__(256) regress_epilogue_bwd_kernel(int N, const float* __restrict__ raw,
                                                                   const float* __restrict__ g_xyz,
                                                                   const float* __restrict__ g_opacity,
                                                                   const float* __restrict__ g_scale,
                                                                   const float* __restrict__ g_rot,
                                                                   const float* __restrict__ g_sh,
                                                                   const float* __restrict__ g_feature,
                                                                   const float* __restrict__ g_feature_n,
                                                                   float* __restrict__ g_raw) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* r = raw + (size_t)RAW * n;
  float v[RAW], g[RAW];
#pragma unroll
  for (int i = 0; i < RAW; i++) v[i] = r[i];
  const size_t i3 = 3 * (size_t)n;
#pragma unroll
  for (int c = 0; c < 3; c++) g[c] = g_xyz ? g_xyz[i3 + c] : 0.f;
  {
    const float s = 1.0f / (1.0f + expf(-v[3]));
    g[3] = g_opacity ? g_opacity[n] * s * (1.0f - s) : 0.f;
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float e = expf(v[4 + c]);
    g[4 + c] = (g_scale && e <= SCALE_MAX) ? g_scale[i3 + c] * e : 0.f;  // clamp_max passes the gradient where it is inactive
  }
  {
    const float nr = sqrtf(v[7] * v[7] + v[8] * v[8] + v[9] * v[9] + v[10] * v[10]);
    float go[4] = {0, 0, 0, 0};
    if (g_rot) {
#pragma unroll
      for (int c = 0; c < 4; c++) go[c] = g_rot[4 * (size_t)n + c];
    }
    if (nr > NORM_EPS) {
      const float inv = 1.0f / nr;
      const float d = (v[7] * go[0] + v[8] * go[1] + v[9] * go[2] + v[10] * go[3]) * inv * inv;
#pragma unroll
      for (int c = 0; c < 4; c++) g[7 + c] = (go[c] - v[7 + c] * d) * inv;
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) g[7 + c] = go[c] * (1.0f / NORM_EPS);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) g[11 + c] = g_sh ? g_sh[12 * (size_t)n + c] : 0.f;
#pragma unroll
  for (int c = 0; c < 9; c++) g[17 + c] = g_sh ? g_sh[12 * (size_t)n + 3 + c] : 0.f;
  {
    const float nf = sqrtf(v[14] * v[14] + v[15] * v[15] + v[16] * v[16]);
    const float dnm = nf + NORM_EPS;
    float gn[3] = {0, 0, 0};
    if (g_feature_n) {
#pragma unroll
      for (int c = 0; c < 3; c++) gn[c] = g_feature_n[i3 + c];
    }
    // out = f / (|f| + eps):  d out_k / d f_c = delta_kc / dnm - f_k f_c / (|f| dnm^2)   (|f| = 0: second term 0)
    const float dot = v[14] * gn[0] + v[15] * gn[1] + v[16] * gn[2];
    const float k = nf > 0.f ? dot / (nf * dnm * dnm) : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) g[14 + c] = gn[c] / dnm - v[14 + c] * k + (g_feature ? g_feature[i3 + c] : 0.f);
  }
  float* o = g_raw + (size_t)RAW * n;
#pragma unroll
  for (int i = 0; i < RAW; i++) o[i] = g[i];
}

}  // namespace mgs

using namespace mgs;

extern "C" {

int mgs_regress_epilogue_forward(int N, const float* raw, const float* xyz_in, float* xyz, float* opacity, float* scale,
                                 float* rot, float* sh, float* feature, float* feature_n, mgs_stream_t stream) {
  if (N < 0) { set_error("regress_epilogue_fwd: N < 0"); return MGS_ERR_INVALID_ARG; }
  if (N == 0) return MGS_OK;
  if (!raw || !xyz_in || !xyz || !opacity || !scale || !rot || !sh || !feature || !feature_n) {
    set_error("regress_epilogue_fwd: NULL pointer");
    return MGS_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(regress_epilogue_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, raw, xyz_in,
                     xyz, opacity, scale, rot, sh, feature, feature_n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("regress_epilogue_fwd: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

int mgs_regress_epilogue_backward(int N, const float* raw, const float* g_xyz, const float* g_opacity, const float* g_scale,
                                  const float* g_rot, const float* g_sh, const float* g_feature, const float* g_feature_n,
                                  float* g_raw, mgs_stream_t stream) {
  if (N < 0) { set_error("regress_epilogue_bwd: N < 0"); return MGS_ERR_INVALID_ARG; }
  if (N == 0) return MGS_OK;
  if (!raw || !g_raw) { set_error("regress_epilogue_bwd: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  hipLaunchKernelGGL(regress_epilogue_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, raw, g_xyz,
                     g_opacity, g_scale, g_rot, g_sh, g_feature, g_feature_n, g_raw);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("regress_epilogue_bwd: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

}  // extern "C"
