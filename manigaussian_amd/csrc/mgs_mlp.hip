// mgs_mlp.hip -- the elementwise passes between the GEMMs of the deformation field's ResnetFC (MG/.../resnetfc.py:10-62,
// 65-177), fused: the GEMMs themselves stay on hipBLASLt (M x 512 x 512 in fp32: 60-76 % of the MFMA peak as measured), but
// autograd's ReLU / residual / bias / bias-gradient passes were 4 ms of a 22 ms configs[3] step.  Streaming kernels, HBM-bound.
//   forward :  a = relu(x),  xb = x + bias          one read, two writes (the residual stream picks up the biases that are
//                                                   added to it later, so that the next GEMM can accumulate into it)
//   backward:  g = g_pre * (act > 0) [+ g_res],  colsum[n] += sum_m g[m][n]     (ReLU mask, residual add and the bias
//                                                   gradient in one pass instead of three)
#include "mgs_common.h"

namespace mgs {

__global__ void __launch_bounds__(256) mlp_relu_bias_kernel(size_t total4, int n4, const float4* __restrict__ x,
                                                            const float4* __restrict__ bias, float4* __restrict__ a,
                                                            float4* __restrict__ xb) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    if (a) a[i] = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    if (xb) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias) b = bias[i % (size_t)n4];
      xb[i] = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
    }
  }
}

// One workgroup = ROWS consecutive rows; thread t owns the float4 column t % n4 of the rows t / n4, t / n4 + 256 / n4, ...
// (n4 = N / 4 divides 256): coalesced rows, private column sums, one atomic per column and workgroup at the end.
constexpr int MLP_ROWS = 128;
// g_out may alias g_pre or g_res (the caller runs it in place): none of the three is __restrict__; every thread loads its own
// elements before it stores them, and no other thread touches them.
__global__ void __launch_bounds__(256) mlp_relu_bwd_kernel(int M, int n4, const float4* g_pre,
                                                           const float4* __restrict__ act,
                                                           const float4* g_res, float4* g_out,
                                                           float* __restrict__ colsum) {
  __shared__ float4 red[256];
  const int c4 = threadIdx.x % n4, r0 = threadIdx.x / n4, rstep = 256 / n4;
  const int row_begin = blockIdx.x * MLP_ROWS, row_end = min(M, row_begin + MLP_ROWS);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int U = 4;  // rows in flight per thread: 8-12 independent 16-B loads before the first use
  for (int rb = row_begin + r0; rb < row_end; rb += U * rstep) {
    float4 gp[U], av[U], q[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int r = rb + u * rstep;
      const size_t i = (size_t)r * n4 + c4;
      const bool ok = r < row_end;
      gp[u] = ok ? g_pre[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      av[u] = ok ? act[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      q[u] = (ok && g_res) ? g_res[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int r = rb + u * rstep;
      if (r >= row_end) break;
      float4 g = make_float4(av[u].x > 0.f ? gp[u].x : 0.f, av[u].y > 0.f ? gp[u].y : 0.f, av[u].z > 0.f ? gp[u].z : 0.f,
                             av[u].w > 0.f ? gp[u].w : 0.f);
      g.x += q[u].x; g.y += q[u].y; g.z += q[u].z; g.w += q[u].w;
      g_out[(size_t)r * n4 + c4] = g;
      s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
    }
  }
  if (!colsum) return;
  red[threadIdx.x] = s;
  __syncthreads();
  if (r0 == 0) {
    for (int k = 1; k < rstep; k++) { const float4 o = red[k * n4 + c4]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    unsafeAtomicAdd(colsum + 4 * c4 + 0, s.x); unsafeAtomicAdd(colsum + 4 * c4 + 1, s.y);
    unsafeAtomicAdd(colsum + 4 * c4 + 2, s.z); unsafeAtomicAdd(colsum + 4 * c4 + 3, s.w);
  }
}

static bool mlp_shape_ok(int M, int N) { return M >= 0 && N > 0 && N % 4 == 0 && (N / 4) <= 256 && 256 % (N / 4) == 0; }

}  // namespace mgs

using namespace mgs;

extern "C" {

int mgs_mlp_relu_bias(int M, int N, const float* x, const float* bias, float* relu_out, float* xb_out, mgs_stream_t stream) {
  if (!mlp_shape_ok(M, N)) { set_error("mlp_relu_bias: N must be a multiple of 4 with N/4 dividing 256"); return MGS_ERR_INVALID_ARG; }
  if (M == 0) return MGS_OK;
  if (!x || (!relu_out && !xb_out)) { set_error("mlp_relu_bias: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  const size_t total4 = (size_t)M * (N / 4);
  const int grid = (int)std::min<size_t>((total4 + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(mlp_relu_bias_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, total4, N / 4, (const float4*)x,
                     (const float4*)bias, (float4*)relu_out, (float4*)xb_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("mlp_relu_bias: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

int mgs_mlp_relu_backward(int M, int N, const float* g_pre, const float* act, const float* g_res, float* g_out,
                          float* colsum, mgs_stream_t stream) {
  if (!mlp_shape_ok(M, N)) { set_error("mlp_relu_backward: N must be a multiple of 4 with N/4 dividing 256"); return MGS_ERR_INVALID_ARG; }
  if (M == 0) return MGS_OK;
  if (!g_pre || !act || !g_out) { set_error("mlp_relu_backward: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  const int grid = (M + MLP_ROWS - 1) / MLP_ROWS;
  hipLaunchKernelGGL(mlp_relu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, N / 4, (const float4*)g_pre,
                     (const float4*)act, (const float4*)g_res, (float4*)g_out, colsum);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("mlp_relu_backward: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

}  // extern "C"
