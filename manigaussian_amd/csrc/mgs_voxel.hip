// mgs_voxel.hip -- per-point latent for the Gaussian regressor / deformation field:
//   canon = (xyz - bb_min) / (bb_max - bb_min)                      (MG/models_embed.py:147-165 world_to_canonical)
//   latent[n, 0:C]     = trilinear sample of voxel_feat [C,D,H,W] at canon      (:167-188, F.grid_sample(align_corners=True,
//                                                                                 mode='bilinear', padding zeros): x -> W, y -> H, z -> D)
//   latent[n, C:C+3+6K] = NeRF positional encoding of canon                      (MG/utils.py:133-169: x, then for each frequency
//                                                                                 f_k = pi * 2^k: sin(f_k x) [3], cos(f_k x) [3])
// One pass instead of clone/sub/div + grid_sample + squeeze/permute/reshape + repeat/addcmul/sin/cat + cat.
// The volume is NCDHW (that is what the Perceiver decoder produces), so the 8 corner reads of one (point, channel) are the
// only locality there is; the kernel is bound by 32-B sector gathers from a C x 4 MB volume, not by arithmetic.
#include "mgs_common.h"

namespace mgs {

struct VoxelArgs {
  int N, C, D, H, W, K;  // K = number of PE frequencies
  float bmin[3], inv_ext[3];
  float freq0;           // freq_factor (pi)
};

struct Corner8 { int x0, y0, z0; float fx, fy, fz; };
__device__ __forceinline__ Corner8 corners(const VoxelArgs& a, float cx, float cy, float cz) {
  Corner8 c;
  const float ix = cx * (float)(a.W - 1), iy = cy * (float)(a.H - 1), iz = cz * (float)(a.D - 1);  // align_corners=True
  const float flx = floorf(ix), fly = floorf(iy), flz = floorf(iz);
  c.x0 = (int)flx; c.y0 = (int)fly; c.z0 = (int)flz;
  c.fx = ix - flx; c.fy = iy - fly; c.fz = iz - flz;
  return c;
}

// thread = (point, channel), channel fastest: the [N, C+PE] output row is written coalesced
__global__ void __launch_bounds__(256) voxel_sample_pe_fwd_kernel(VoxelArgs a, const float* __restrict__ voxel,
                                                                   const float* __restrict__ xyz, float* __restrict__ out) {
  const int PE = 3 + 6 * a.K, ROW = a.C + PE;
  const size_t total = (size_t)a.N * ROW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / ROW;
    const int c = (int)(i - n * ROW);
    const float cx = (xyz[3 * n] - a.bmin[0]) * a.inv_ext[0];
    const float cy = (xyz[3 * n + 1] - a.bmin[1]) * a.inv_ext[1];
    const float cz = (xyz[3 * n + 2] - a.bmin[2]) * a.inv_ext[2];
    float v;
    if (c < a.C) {
      const Corner8 k = corners(a, cx, cy, cz);
      const float* __restrict__ plane = voxel + (size_t)c * a.D * a.H * a.W;
      v = 0.f;
#pragma unroll
      for (int dz = 0; dz < 2; dz++)
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
          for (int dx = 0; dx < 2; dx++) {
            const int x = k.x0 + dx, y = k.y0 + dy, z = k.z0 + dz;
            const float w = (dx ? k.fx : 1.f - k.fx) * (dy ? k.fy : 1.f - k.fy) * (dz ? k.fz : 1.f - k.fz);
            if (x >= 0 && x < a.W && y >= 0 && y < a.H && z >= 0 && z < a.D) v += w * plane[((size_t)z * a.H + y) * a.W + x];
          }
    } else {
      const int e = c - a.C;  // 0..2: x y z; then blocks of 3: sin(f_0 .), cos(f_0 .), sin(f_1 .), ...
      const float comp = (e % 3 == 0) ? cx : ((e % 3 == 1) ? cy : cz);
      if (e < 3) v = comp;
      else {
        const int blk = (e - 3) / 3;                // 0: sin f0, 1: cos f0, 2: sin f1, ...
        const float f = a.freq0 * (float)(1 << (blk >> 1));
        // the reference evaluates sin(x * f + phase) with phase = 0 or pi/2 (MG/utils.py:152-166): keep that form
        v = sinf(fmaf(comp, f, (blk & 1) ? 1.57079632679489661923f : 0.f));
      }
    }
    out[i] = v;
  }
}

// gradient w.r.t. the volume (the points are data): thread = (point, channel); g_voxel must be zero on entry
__global__ void __launch_bounds__(256) voxel_sample_bwd_kernel(VoxelArgs a, const float* __restrict__ xyz,
                                                                const float* __restrict__ g_out, int row_stride,
                                                                float* __restrict__ g_voxel) {
  const size_t total = (size_t)a.N * a.C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / a.C;
    const int c = (int)(i - n * a.C);
    const float g = g_out[n * row_stride + c];
    if (g == 0.f) continue;
    const float cx = (xyz[3 * n] - a.bmin[0]) * a.inv_ext[0];
    const float cy = (xyz[3 * n + 1] - a.bmin[1]) * a.inv_ext[1];
    const float cz = (xyz[3 * n + 2] - a.bmin[2]) * a.inv_ext[2];
    const Corner8 k = corners(a, cx, cy, cz);
    float* __restrict__ plane = g_voxel + (size_t)c * a.D * a.H * a.W;
#pragma unroll
    for (int dz = 0; dz < 2; dz++)
#pragma unroll
      for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
          const int x = k.x0 + dx, y = k.y0 + dy, z = k.z0 + dz;
          const float w = (dx ? k.fx : 1.f - k.fx) * (dy ? k.fy : 1.f - k.fy) * (dz ? k.fz : 1.f - k.fz);
          if (x >= 0 && x < a.W && y >= 0 && y < a.H && z >= 0 && z < a.D)
            unsafeAtomicAdd(plane + ((size_t)z * a.H + y) * a.W + x, w * g);
        }
  }
}

}  // namespace mgs

using namespace mgs;

static int fill_voxel_args(VoxelArgs& a, int N, int C, int D, int H, int W, int K, const float* bounds, float freq_factor) {
  if (N < 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || K < 0 || K > 16 || !bounds) {
    set_error("voxel_sample: bad shape (N=%d C=%d D=%d H=%d W=%d K=%d) or NULL bounds", N, C, D, H, W, K);
    return MGS_ERR_INVALID_ARG;
  }
  a.N = N; a.C = C; a.D = D; a.H = H; a.W = W; a.K = K; a.freq0 = freq_factor;
  for (int i = 0; i < 3; i++) { a.bmin[i] = bounds[i]; a.inv_ext[i] = 1.0f / (bounds[3 + i] - bounds[i]); }
  return MGS_OK;
}

extern "C" {

int mgs_voxel_sample_pe_forward(int N, int C, int D, int H, int W, int K, float freq_factor, const float* bounds_host,
                                const float* voxel, const float* xyz, float* out, mgs_stream_t stream) {
  VoxelArgs a;
  int rc = fill_voxel_args(a, N, C, D, H, W, K, bounds_host, freq_factor);
  if (rc) return rc;
  if (N == 0) return MGS_OK;
  if (!voxel || !xyz || !out) { set_error("voxel_sample_fwd: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  const size_t total = (size_t)N * (C + 3 + 6 * K);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(voxel_sample_pe_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, voxel, xyz, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("voxel_sample_fwd: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

int mgs_voxel_sample_backward(int N, int C, int D, int H, int W, const float* bounds_host, const float* xyz,
                              const float* g_out, int row_stride, float* g_voxel, mgs_stream_t stream) {
  VoxelArgs a;
  int rc = fill_voxel_args(a, N, C, D, H, W, 0, bounds_host, 0.f);
  if (rc) return rc;
  if (N == 0) return MGS_OK;
  if (!xyz || !g_out || !g_voxel || row_stride < C) { set_error("voxel_sample_bwd: NULL pointer or row_stride < C"); return MGS_ERR_INVALID_ARG; }
  const size_t total = (size_t)N * C;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(voxel_sample_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, xyz, g_out, row_stride,
                     g_voxel);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("voxel_sample_bwd: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

}  // extern "C"
