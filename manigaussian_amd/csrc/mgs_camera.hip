// mgs_camera.hip -- camera calibration for the rasterizer, replacing the per-step host round trip of
// NeuralRenderer.get_novel_calib (MG/neural_rendering.py:205-248: .cpu().numpy(), np.linalg.inv, seven small H2D copies).
//   extr  = inv(c2w)                                                   (:221; the saved extrinsic is cam2world)
//   R     = extr[:3,:3]^T (float32), T = extr[:3,3] (float32)          (:224-225)
//   Fov   = 2 atan(pixels / (2 focal))                                 (MG/graphics_utils.py:51-52; NEGATIVE for PyRep's
//                                                                       negative focal lengths -- kept)
//   W2V   = float32(inv(C2W')), C2W' = inv([R^T T; 0 1]) with centre (c + trans) * scale   (graphics_utils.py:17-29)
//   world_view_transform = W2V^T, proj = getProjectionMatrix(znear, zfar, K, h, w)^T        (graphics_utils.py:31-48)
//   full_proj_transform = world_view_transform @ proj, camera_center = inv(world_view_transform)[3,:3]
// One routine, compiled for host and device: the host entry point is what a data-loader cache calls once per camera file
// (it also yields tan(Fov/2) as host floats, which GaussianRasterizationSettings needs without a device sync); the kernel
// serves intrinsics/extrinsics that already live on the device.  Arithmetic is float64 with float32 rounding where the
// reference rounds (extr -> R, T; the final matrices), so results agree with the numpy/torch original to float32 rounding.
#include <math.h>

#include "mgs_common.h"

namespace mgs {

struct CalibArgs { int V, W, H; float znear, zfar, tx, ty, tz, scale; };

// general 4x4 inverse by cofactors (row-major); returns false if singular
__host__ __device__ inline bool inv4(const double* m, double* o) {
  double inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == 0.0) return false;
  const double r = 1.0 / det;
  for (int i = 0; i < 16; i++) o[i] = inv[i] * r;
  return true;
}

// One camera.  c2w [16], K [9] row-major float32.  Outputs row-major float32; any may be NULL.
__host__ __device__ inline bool calib_one(const CalibArgs& a, const float* c2w, const float* K, float* wvt, float* fpt,
                                          float* centre, float* fov, float* tanfov) {
  double m[16], e[16];
  for (int i = 0; i < 16; i++) m[i] = (double)c2w[i];
  if (!inv4(m, e)) return false;
  // R = float32(extr[:3,:3])^T, T = float32(extr[:3,3]); Rt = [R^T T; 0 1] = float32-rounded extr with a clean last row
  double Rt[16];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) Rt[4 * r + c] = (double)(float)e[4 * r + c];
  Rt[12] = Rt[13] = Rt[14] = 0.0; Rt[15] = 1.0;
  double C2W[16], W2V[16];
  if (!inv4(Rt, C2W)) return false;
  C2W[3] = (C2W[3] + (double)a.tx) * (double)a.scale;
  C2W[7] = (C2W[7] + (double)a.ty) * (double)a.scale;
  C2W[11] = (C2W[11] + (double)a.tz) * (double)a.scale;
  if (!inv4(C2W, W2V)) return false;
  float wv[16];  // world_view_transform = float32(W2V)^T
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) wv[4 * r + c] = (float)W2V[4 * c + r];
  // getProjectionMatrix (graphics_utils.py:31-48), then transposed
  const double fx = (double)K[0], fy = (double)K[4], cx = (double)K[2], cy = (double)K[5];
  const double zn = (double)a.znear, zf = (double)a.zfar;
  const double near_fx = zn / fx, near_fy = zn / fy;
  const double left = -((double)a.W - cx) * near_fx, right = cx * near_fx;
  const double bottom = (cy - (double)a.H) * near_fy, top = cy * near_fy;
  float P[16];
  for (int i = 0; i < 16; i++) P[i] = 0.f;
  P[0] = (float)(2.0 * zn / (right - left));
  P[5] = (float)(2.0 * zn / (top - bottom));
  P[2] = (float)((right + left) / (right - left));
  P[6] = (float)((top + bottom) / (top - bottom));
  P[14] = 1.f;
  P[10] = (float)(zf / (zf - zn));
  P[11] = (float)(-(zf * zn) / (zf - zn));
  if (wvt) for (int i = 0; i < 16; i++) wvt[i] = wv[i];
  if (fpt)  // full = wv @ P^T : full[r][c] = sum_k wv[r][k] * P[c][k]
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        double s = 0.0;
        for (int k = 0; k < 4; k++) s += (double)wv[4 * r + k] * (double)P[4 * c + k];
        fpt[4 * r + c] = (float)s;
      }
  if (centre) {  // inverse(world_view_transform)[3, :3]
    double w[16], wi[16];
    for (int i = 0; i < 16; i++) w[i] = (double)wv[i];
    if (!inv4(w, wi)) return false;
    centre[0] = (float)wi[12]; centre[1] = (float)wi[13]; centre[2] = (float)wi[14];
  }
  const double fovx = 2.0 * atan((double)a.W / (2.0 * fx)), fovy = 2.0 * atan((double)a.H / (2.0 * fy));
  if (fov) { fov[0] = (float)fovx; fov[1] = (float)fovy; }
  // math.tan(FovX * 0.5) on the float32 value, as render() evaluates it (MG/gaussian_renderer/__init__.py:35-36)
  if (tanfov) { tanfov[0] = (float)tan((double)(float)fovx * 0.5); tanfov[1] = (float)tan((double)(float)fovy * 0.5); }
  return true;
}

__global__ void novel_calib_kernel(CalibArgs a, const float* __restrict__ c2w, const float* __restrict__ K,
                                   float* __restrict__ wvt, float* __restrict__ fpt, float* __restrict__ centre,
                                   float* __restrict__ fov, float* __restrict__ tanfov, int* __restrict__ bad) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= a.V) return;
  const bool ok = calib_one(a, c2w + 16 * v, K + 9 * v, wvt ? wvt + 16 * v : nullptr, fpt ? fpt + 16 * v : nullptr,
                            centre ? centre + 3 * v : nullptr, fov ? fov + 2 * v : nullptr,
                            tanfov ? tanfov + 2 * v : nullptr);
  if (!ok && bad) atomicOr(bad, 1);
}

}  // namespace mgs

using namespace mgs;

static int fill_calib(CalibArgs& a, int V, int W, int H, float znear, float zfar, float tx, float ty, float tz, float scale,
                      const void* c2w, const void* K) {
  if (V < 0 || W <= 0 || H <= 0 || !(zfar > znear) || !(znear > 0.f) || (V > 0 && (!c2w || !K))) {
    set_error("novel_calib: bad arguments (V=%d W=%d H=%d znear=%g zfar=%g) or NULL input", V, W, H, znear, zfar);
    return MGS_ERR_INVALID_ARG;
  }
  a.V = V; a.W = W; a.H = H; a.znear = znear; a.zfar = zfar; a.tx = tx; a.ty = ty; a.tz = tz; a.scale = scale;
  return MGS_OK;
}

extern "C" {

int mgs_novel_calib(int V, const float* c2w, const float* K, int W, int H, float znear, float zfar, float trans_x,
                    float trans_y, float trans_z, float scale, float* world_view_transform, float* full_proj_transform,
                    float* camera_center, float* fov, float* tanfov, int32_t* singular, mgs_stream_t stream) {
  CalibArgs a;
  int rc = fill_calib(a, V, W, H, znear, zfar, trans_x, trans_y, trans_z, scale, c2w, K);
  if (rc || V == 0) return rc;
  hipLaunchKernelGGL(novel_calib_kernel, dim3((V + 63) / 64), dim3(64), 0, (hipStream_t)stream, a, c2w, K,
                     world_view_transform, full_proj_transform, camera_center, fov, tanfov, (int*)singular);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("novel_calib: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

int mgs_novel_calib_host(int V, const float* c2w, const float* K, int W, int H, float znear, float zfar, float trans_x,
                         float trans_y, float trans_z, float scale, float* world_view_transform,
                         float* full_proj_transform, float* camera_center, float* fov, float* tanfov) {
  CalibArgs a;
  int rc = fill_calib(a, V, W, H, znear, zfar, trans_x, trans_y, trans_z, scale, c2w, K);
  if (rc) return rc;
  for (int v = 0; v < V; v++)
    if (!calib_one(a, c2w + 16 * v, K + 9 * v, world_view_transform ? world_view_transform + 16 * v : nullptr,
                   full_proj_transform ? full_proj_transform + 16 * v : nullptr,
                   camera_center ? camera_center + 3 * v : nullptr, fov ? fov + 2 * v : nullptr,
                   tanfov ? tanfov + 2 * v : nullptr)) {
      set_error("novel_calib: camera %d has a singular cam2world matrix", v);
      return MGS_ERR_INVALID_ARG;
    }
  return MGS_OK;
}

}  // extern "C"
