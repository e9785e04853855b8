// mgs_deform.hip -- deformation-field glue around the (torch/hipBLASLt) ResnetFC GEMMs:
// fused input assembly (one pass instead of 9 intermediate tensors + two cats + a repeat) and the
// apply epilogue, forward and backward.  Reference: agents/manigaussian_bc/models_embed.py:255-304.
// Pure streaming kernels, HBM-bound: 4 B read + 4 B written per assembled element.
#include "mgs_common.h"

namespace mgs {

// column layout of one assembled row (models_embed.py:258-287):
//   [0,DL) point_latent | xyz 3 | f_dc 3 + f_rest 9 (= sh[n] flattened, 12) | rot 4 | scale 3 | opacity 1 |
//   (feature 3) | z_feature DZ | action DA
__global__ void __launch_bounds__(256) deform_assemble_fwd_kernel(
    size_t total, int stride, int DL, int DZ, int DA, int has_feat, const float* __restrict__ point_latent,
    const float* __restrict__ xyz, const float* __restrict__ sh, const float* __restrict__ rot,
    const float* __restrict__ scale, const float* __restrict__ opacity, const float* __restrict__ feature,
    const float* __restrict__ z_feature, const float* __restrict__ action, float* __restrict__ out) {
  const int o_xyz = DL, o_sh = DL + 3, o_rot = DL + 15, o_scale = DL + 19, o_op = DL + 22;
  const int o_feat = DL + 23, o_z = o_feat + (has_feat ? 3 : 0), o_act = o_z + DZ;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / stride;
    const int c = (int)(i - n * stride);
    float v;
    if (c < o_xyz) v = point_latent[n * DL + c];
    else if (c < o_sh) v = xyz[n * 3 + (c - o_xyz)];
    else if (c < o_rot) v = sh[n * 12 + (c - o_sh)];
    else if (c < o_scale) v = rot[n * 4 + (c - o_rot)];
    else if (c < o_op) v = scale[n * 3 + (c - o_scale)];
    else if (c < o_feat) v = opacity[n];
    else if (c < o_z) v = feature[n * 3 + (c - o_feat)];
    else if (c < o_act) v = z_feature[n * DZ + (c - o_z)];
    else v = action[c - o_act];
    out[i] = v;
  }
}

__global__ void __launch_bounds__(256) deform_assemble_bwd_kernel(size_t N, int stride, int DL, int DZ, int o_z,
                                                                  const float* __restrict__ g_out,
                                                                  float* __restrict__ g_latent,
                                                                  float* __restrict__ g_z) {
  const int w = DL + DZ;
  const size_t total = N * (size_t)w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / w;
    const int c = (int)(i - n * w);
    if (c < DL) g_latent[n * DL + c] = g_out[n * stride + c];
    else g_z[n * DZ + (c - DL)] = g_out[n * stride + o_z + (c - DL)];
  }
}

__global__ void __launch_bounds__(256) deform_apply_fwd_kernel(int N, const float* __restrict__ xyz,
                                                               const float* __restrict__ rot,
                                                               const float* __restrict__ delta,
                                                               float* __restrict__ xyz_out, float* __restrict__ rot_out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* d = delta + 7 * (size_t)n;
#pragma unroll
  for (int c = 0; c < 3; c++) xyz_out[3 * (size_t)n + c] = xyz[3 * (size_t)n + c] + d[c];
  float q[4], n2 = 0.f;
#pragma unroll
  for (int c = 0; c < 4; c++) { q[c] = rot[4 * (size_t)n + c] + d[3 + c]; n2 += q[c] * q[c]; }
  const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);  // F.normalize eps
#pragma unroll
  for (int c = 0; c < 4; c++) rot_out[4 * (size_t)n + c] = q[c] * inv;
}

__global__ void __launch_bounds__(256) deform_apply_bwd_kernel(int N, const float* __restrict__ rot,
                                                               const float* __restrict__ delta,
                                                               const float* __restrict__ g_xyz,
                                                               const float* __restrict__ g_rot,
                                                               float* __restrict__ g_delta) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* d = delta + 7 * (size_t)n;
  float* o = g_delta + 7 * (size_t)n;
#pragma unroll
  for (int c = 0; c < 3; c++) o[c] = g_xyz[3 * (size_t)n + c];
  float q[4], g[4], n2 = 0.f, dt = 0.f;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    q[c] = rot[4 * (size_t)n + c] + d[3 + c];
    g[c] = g_rot[4 * (size_t)n + c];
    n2 += q[c] * q[c];
    dt += q[c] * g[c];
  }
  const float nn = sqrtf(n2);
  if (nn > 1e-12f) {
    const float inv = 1.0f / nn, inv3 = inv * inv * inv;
#pragma unroll
    for (int c = 0; c < 4; c++) o[3 + c] = g[c] * inv - q[c] * dt * inv3;
  } else {
#pragma unroll
    for (int c = 0; c < 4; c++) o[3 + c] = g[c] * 1e12f;
  }
}

}  // namespace mgs

using namespace mgs;

static int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g ? g : 1));
}

extern "C" {

int mgs_deform_assemble_forward(int N, int DL, int DZ, int DA, const float* point_latent, const float* xyz,
                                const float* sh, const float* rot, const float* scale, const float* opacity,
                                const float* feature, const float* z_feature, const float* action, float* out,
                                mgs_stream_t stream) {
  if (N < 0 || DL < 0 || DZ < 0 || DA < 0) { set_error("deform_assemble: negative size"); return MGS_ERR_INVALID_ARG; }
  if (N == 0) return MGS_OK;
  if ((DL && !point_latent) || !xyz || !sh || !rot || !scale || !opacity || (DZ && !z_feature) || (DA && !action) || !out) {
    set_error("deform_assemble: NULL pointer");
    return MGS_ERR_INVALID_ARG;
  }
  const int stride = DL + 23 + (feature ? 3 : 0) + DZ + DA;
  const size_t total = (size_t)N * stride;
  hipLaunchKernelGGL(deform_assemble_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, total, stride,
                     DL, DZ, DA, feature ? 1 : 0, point_latent, xyz, sh, rot, scale, opacity, feature, z_feature, action,
                     out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("deform_assemble_fwd: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

int mgs_deform_assemble_backward(int N, int DL, int DZ, int DA, int has_feature, const float* g_out,
                                 float* g_point_latent, float* g_z_feature, mgs_stream_t stream) {
  if (N < 0 || DL < 0 || DZ < 0 || DA < 0) { set_error("deform_assemble_bwd: negative size"); return MGS_ERR_INVALID_ARG; }
  if (N == 0 || DL + DZ == 0) return MGS_OK;
  if (!g_out || (DL && !g_point_latent) || (DZ && !g_z_feature)) { set_error("deform_assemble_bwd: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  const int stride = DL + 23 + (has_feature ? 3 : 0) + DZ + DA;
  const int o_z = DL + 23 + (has_feature ? 3 : 0);
  const size_t total = (size_t)N * (DL + DZ);
  hipLaunchKernelGGL(deform_assemble_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (size_t)N,
                     stride, DL, DZ, o_z, g_out, g_point_latent, g_z_feature);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("deform_assemble_bwd: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

int mgs_deform_apply_forward(int N, const float* xyz, const float* rot, const float* delta, float* xyz_out,
                             float* rot_out, mgs_stream_t stream) {
  if (N < 0) { set_error("deform_apply: N < 0"); return MGS_ERR_INVALID_ARG; }
  if (N == 0) return MGS_OK;
  if (!xyz || !rot || !delta || !xyz_out || !rot_out) { set_error("deform_apply: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  hipLaunchKernelGGL(deform_apply_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, xyz, rot,
                     delta, xyz_out, rot_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("deform_apply_fwd: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

int mgs_deform_apply_backward(int N, const float* rot, const float* delta, const float* g_xyz_out,
                              const float* g_rot_out, float* g_delta, mgs_stream_t stream) {
  if (N < 0) { set_error("deform_apply_bwd: N < 0"); return MGS_ERR_INVALID_ARG; }
  if (N == 0) return MGS_OK;
  if (!rot || !delta || !g_xyz_out || !g_rot_out || !g_delta) { set_error("deform_apply_bwd: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  hipLaunchKernelGGL(deform_apply_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, rot, delta,
                     g_xyz_out, g_rot_out, g_delta);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("deform_apply_bwd: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

}  // extern "C"
