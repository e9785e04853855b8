// resblock_fused.hip -- EXPERIMENT (round 5, measured, NOT shipped: profiles/r05_exp_resblock.log, profiles/EXPERIMENTS.md): one
// ResnetBlockFC of the deformation MLP as ONE kernel, forward and data-gradient backward (gfx950).  Built by scripts/ubench/Makefile
// into scripts/ubench/libresblock.so and driven by scripts/bench_resblock.py; nothing in manigaussian_amd/ uses it.
//
// Reference: MG/resnetfc.py:10-62 (ResnetBlockFC: x + fc_1(relu(fc_0(relu(x)))), size_in = size_h = size_out = 512, beta = 0 ->
// ReLU) inside ResnetFC.forward (:137-177), MG = agents/manigaussian_bc.  Rounds 2-4 ran a block as two hipBLASLt GEMMs
// (100 000 x 512 x 512 each, 89 % of the fp32 matrix peak) plus streaming passes around them (ReLU / bias / residual forward,
// ReLU mask + residual + bias gradient backward: 2.6 ms of a 16 ms step).  Here the two contractions of a block are CHAINED
// inside a workgroup: a 64-row tile of the activation sits in LDS, the first product stays in registers, goes through its
// elementwise step and back into the same LDS buffer as the A operand of the second product; biases, ReLU, masks, the residual
// and the bias-gradient column sums ride in the epilogues.  Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32, an fmaf chain
// (cdna_hip_programming.md 3) -- so the result differs from the GEMM library's only by the order of the 512-term sums.
//
//   forward :  a = relu(s);  h = relu(a W0^T + b0);  out = s + bias1 + h W1^T           writes a, h (the backward's operands), out
//   backward:  gh = (g W1) . [h > 0];  g' = (gh W0) . [s > 0] + g                       writes gh, g', column sums of both
//
// Layout.  A tile in LDS: Ap[p = 2 kb + kh][row ^ (p & 7)][4]: the four floats are k = 8 kb + 4 kh + 0..3 of one row -- an
// MFMA lane (row, kh) reads its operands of four consecutive MFMAs with one ds_read_b128; the XOR keeps both the reads and
// the epilogue's column-major writes conflict-free.  Weights are repacked once per step (1 MB each, mgs_mlp_pack_weight) to
// Wp[p][n][4] = B[8 kb + 4 kh + 0..3][n]: a wave's B operands of four MFMAs are one coalesced global_load_dwordx4 per column
// tile, straight from L2 into registers (the fp32 MFMA takes 64 cycles per issue: operand delivery is nowhere near a limit).
// A workgroup = 4 waves on a 32-row tile, TWO workgroups per CU (64 KB of LDS and <= 256 registers each): while one is in an
// epilogue or loading its rows, the other keeps the matrix pipe busy (a first version -- 64-row tiles, one workgroup per CU, every
// wave in the same phase at the same time -- measured 66 % matrix-pipe utilisation: profiles/r05_exp_resblock.log).  Wave w
// owns output columns [128 w, 128 w + 128): four 32 x 32 accumulator tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace mgs {
typedef void* mgs_stream_t;
enum { MGS_OK = 0, MGS_ERR_INVALID_ARG = -1, MGS_ERR_HIP = -2 };
static void set_error(const char* fmt, ...) { (void)fmt; }

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int HID = 512;   // d_hidden (conf/method/ManiGaussian_BC.yaml:146-157)
constexpr int MB_ROWS = 32;
constexpr int NKB = HID / 8;

__device__ __forceinline__ int a_slot(int p, int row) { return (p * MB_ROWS + (row ^ (p & 7))) * 4; }  // float index

// acc[ct] += A[rows 0..31][k] B[k][cols 128 w + 32 ct ..] over k = 0 .. 511
__device__ __forceinline__ void chain_gemm(const float* __restrict__ As, const float4* __restrict__ Wp, int w, int lane,
                                           f32x16 (&acc)[4], int dbg) {
  const int i = lane & 31, kh = lane >> 5;
  const float4* __restrict__ bp = Wp + (size_t)kh * HID + w * 128 + i;
  float4 a0, b0[4], a1, b1[4];
  auto load = [&](int kb, float4& a, float4 (&b)[4]) {
    a = *reinterpret_cast<const float4*>(As + a_slot(2 * kb + kh, i));
#pragma unroll
    for (int ct = 0; ct < 4; ct++) b[ct] = bp[(size_t)((dbg & 4) ? 0 : kb) * 2 * HID + ct * 32];
  };
  auto mma = [&](const float4& a, const float4 (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        const float av = j == 0 ? a.x : j == 1 ? a.y : j == 2 ? a.z : a.w;
        const float bv = j == 0 ? b[ct].x : j == 1 ? b[ct].y : j == 2 ? b[ct].z : b[ct].w;
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ct], 0, 0, 0);
      }
  };
  load(0, a0, b0);
#pragma unroll 1
  for (int kb = 0; kb < NKB; kb += 2) {
    load(kb + 1, a1, b1);
    mma(a0, b0);
    if (kb + 2 < NKB) load(kb + 2, a0, b0);
    mma(a1, b1);
  }
}

// One 32-row tile.  MODE 0: forward, 1: backward (data gradients).  FULL: every row of the tile exists (no bounds tests).
template <int MODE, bool FULL>
__device__ __forceinline__ void resblock_tile(float* As, int M, int row0, int tid, int lane, int w, const float* __restrict__ in,
                                              const float* __restrict__ aux_h, const float* __restrict__ aux_s,
                                              const float4* __restrict__ WpA, const float* __restrict__ biasA,
                                              const float4* __restrict__ WpB, const float* __restrict__ biasB,
                                              float* __restrict__ out_a, float* __restrict__ out_mid, float* __restrict__ out,
                                              float (&cs_mid)[4], float (&cs_out)[4], int dbg) {
  const int i = lane & 31, kh = lane >> 5;
  // ---- the tile's rows -> LDS (A layout); forward: through the ReLU, which also leaves as `a`.  Eight loads in flight per
  //      thread (written as load-all / store-all: a conditional store between the loads serialises them) ----
#pragma unroll 1
  for (int itb = 0; itb < MB_ROWS / 16; itb++) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int f = tid + 256 * (8 * itb + u);
      const int row = f >> 7, k4 = (f & 127) * 4;
      const int grow = FULL ? row0 + row : min(row0 + row, M - 1);
      v[u] = *reinterpret_cast<const float4*>(in + (size_t)grow * HID + k4);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int f = tid + 256 * (8 * itb + u);
      const int row = f >> 7, k4 = (f & 127) * 4;
      if constexpr (MODE == 0) {
        v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f); v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
        if (FULL || row0 + row < M) *reinterpret_cast<float4*>(out_a + (size_t)(row0 + row) * HID + k4) = v[u];
      }
      *reinterpret_cast<float4*>(As + a_slot(k4 >> 2, row)) = v[u];
    }
  }
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[ct][r] = 0.f;
  chain_gemm(As, WpA, w, lane, acc, dbg);
  __syncthreads();  // every wave has read the whole A tile: the buffer takes the middle activation now
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {  // one 32 x 32 accumulator tile at a time: 16 loads in flight
    const int n = w * 128 + ct * 32 + i;
    const float bA = MODE == 0 ? biasA[n] : 0.f;
    float hm[16];
    if constexpr (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        hm[r] = aux_h[(size_t)(FULL ? grow : min(grow, M - 1)) * HID + n];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const int grow = row0 + row;
      const bool ok = FULL || grow < M;
      float v = acc[ct][r];
      if constexpr (MODE == 0) {
        v = fmaxf(v + bA, 0.f);
      } else {
        v = (hm[r] > 0.f && ok) ? v : 0.f;
        cs_mid[ct] += v;
      }
      if (ok && !(dbg & 2)) out_mid[(size_t)grow * HID + n] = v;
      As[a_slot(n >> 2, row) + (n & 3)] = v;
      acc[ct][r] = 0.f;
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  chain_gemm(As, WpB, w, lane, acc, dbg);
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    const int n = w * 128 + ct * 32 + i;
    const float bB = MODE == 0 ? biasB[n] : 0.f;
    float res[16], sm[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      const size_t o = (size_t)(FULL ? grow : min(grow, M - 1)) * HID + n;
      res[r] = in[o];  // forward: s (the residual); backward: g (the skip path's gradient)
      if constexpr (MODE == 1) sm[r] = aux_s[o];
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      const bool ok = FULL || grow < M;
      float v = acc[ct][r];
      if constexpr (MODE == 0) {
        v = v + bB + res[r];
      } else {
        v = ok ? (sm[r] > 0.f ? v : 0.f) + res[r] : 0.f;
        cs_out[ct] += v;
      }
      if (ok && !(dbg & 1)) out[(size_t)grow * HID + n] = v;
    }
    asm volatile("" ::: "memory");
  }
}

// in: s (forward) / g (backward), [M, 512].
template <int MODE>
__global__ void __launch_bounds__(256, 2) resblock_kernel(int M, int ntiles, const float* __restrict__ in,
                                                         const float* __restrict__ aux_h,   // bwd: h (the inner mask)
                                                         const float* __restrict__ aux_s,   // bwd: s (the outer mask)
                                                         const float4* __restrict__ WpA, const float* __restrict__ biasA,
                                                         const float4* __restrict__ WpB, const float* __restrict__ biasB,
                                                         float* __restrict__ out_a,    // fwd: relu(s)
                                                         float* __restrict__ out_mid,  // fwd: h          bwd: gh
                                                         float* __restrict__ out,      // fwd: block out  bwd: g'
                                                         float* __restrict__ colsum_mid, float* __restrict__ colsum_out, int dbg) {
  __shared__ float As[2 * NKB * MB_ROWS * 4];  // 64 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float cs_mid[4] = {0.f, 0.f, 0.f, 0.f}, cs_out[4] = {0.f, 0.f, 0.f, 0.f};
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * MB_ROWS;
    __syncthreads();  // the previous tile's second product has read the buffer
    if (row0 + MB_ROWS <= M)
      resblock_tile<MODE, true>(As, M, row0, tid, lane, w, in, aux_h, aux_s, WpA, biasA, WpB, biasB, out_a, out_mid, out, cs_mid, cs_out, dbg);
    else
      resblock_tile<MODE, false>(As, M, row0, tid, lane, w, in, aux_h, aux_s, WpA, biasA, WpB, biasB, out_a, out_mid, out, cs_mid, cs_out, dbg);
  }
  if constexpr (MODE == 1) {  // one atomic per column and workgroup (the two row halves of a column are joined first)
    const int i = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
      const int n = w * 128 + ct * 32 + i;
      const float m = cs_mid[ct] + __shfl_xor(cs_mid[ct], 32, 64), o = cs_out[ct] + __shfl_xor(cs_out[ct], 32, 64);
      if (kh == 0) {
        if (colsum_mid) unsafeAtomicAdd(colsum_mid + n, m);
        if (colsum_out) unsafeAtomicAdd(colsum_out + n, o);
      }
    }
  }
}

// Wp[(2 kb + kh) * 512 + n] = { B[8 kb + 4 kh + j][n] : j = 0..3 };  transpose: B[k][n] = W[n][k] (forward: x W^T), else W[k][n]
__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ W, int transpose, float4* __restrict__ Wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // p * 512 + n
  if (idx >= 2 * NKB * HID) return;
  const int p = idx / HID, n = idx % HID, k0 = 4 * p;
  float4 v;
  if (transpose) v = *reinterpret_cast<const float4*>(W + (size_t)n * HID + k0);
  else v = make_float4(W[(size_t)k0 * HID + n], W[(size_t)(k0 + 1) * HID + n], W[(size_t)(k0 + 2) * HID + n], W[(size_t)(k0 + 3) * HID + n]);
  Wp[idx] = v;
}

}  // namespace mgs

using namespace mgs;

extern "C" {

int mgs_mlp_pack_weight(const float* W, int transpose, float* Wp, mgs_stream_t stream) {
  if (!W || !Wp) { set_error("mlp_pack_weight: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  hipLaunchKernelGGL(pack_weight_kernel, dim3(2 * NKB * HID / 256), dim3(256), 0, (hipStream_t)stream, W, transpose,
                     reinterpret_cast<float4*>(Wp));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("mlp_pack_weight failed: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

static int mlp_dbg() { const char* e = getenv("MGS_MLP_DBG"); return e ? atoi(e) : 0; }  // timing experiments only
static int block_grid(int ntiles) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return ntiles < 2 * cus ? ntiles : 2 * cus;  // two workgroups per CU
}

int mgs_mlp_resblock_forward(int M, const float* s, const float* W0p, const float* b0, const float* W1p, const float* bias1,
                             float* a_out, float* h_out, float* out, mgs_stream_t stream) {
  if (M <= 0) return MGS_OK;
  if (!s || !W0p || !b0 || !W1p || !bias1 || !a_out || !h_out || !out) { set_error("mlp_resblock_forward: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  const int ntiles = (M + MB_ROWS - 1) / MB_ROWS;
  hipLaunchKernelGGL(resblock_kernel<0>, dim3(block_grid(ntiles)), dim3(256), 0, (hipStream_t)stream, M, ntiles, s, nullptr, nullptr,
                     reinterpret_cast<const float4*>(W0p), b0, reinterpret_cast<const float4*>(W1p), bias1, a_out, h_out, out,
                     nullptr, nullptr, mlp_dbg());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("mlp_resblock_forward failed: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

int mgs_mlp_resblock_backward(int M, const float* g, const float* h, const float* s, const float* W1p, const float* W0p,
                              float* gh_out, float* g_out, float* colsum_gh, float* colsum_g, mgs_stream_t stream) {
  if (M <= 0) return MGS_OK;
  if (!g || !h || !s || !W1p || !W0p || !gh_out || !g_out) { set_error("mlp_resblock_backward: NULL pointer"); return MGS_ERR_INVALID_ARG; }
  const int ntiles = (M + MB_ROWS - 1) / MB_ROWS;
  hipLaunchKernelGGL(resblock_kernel<1>, dim3(block_grid(ntiles)), dim3(256), 0, (hipStream_t)stream, M, ntiles, g, h, s,
                     reinterpret_cast<const float4*>(W1p), nullptr, reinterpret_cast<const float4*>(W0p), nullptr, nullptr, gh_out,
                     g_out, colsum_gh, colsum_g, mlp_dbg());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("mlp_resblock_backward failed: %s", hipGetErrorString(e)); return MGS_ERR_HIP; }
  return MGS_OK;
}

}  // extern "C"
