// mgs_device.h -- wave64 cross-lane primitives for gfx950 (CDNA4).  Device code only.
#pragma once
#include <hip/hip_runtime.h>

namespace mgs {

// ---- DPP controls (gfx9 encoding) -----------------------------------------------------------------
constexpr int DPP_QUAD_XOR1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_ROR8 = 0x128;       // row_ror:8  (lane l <- lane (l+8)%16 of its row == l^8)
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane l <- lane 7-(l%8) of its 8-lane half

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ float bcast_lane(float v, int j /*wave-uniform*/) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ uint32_t bcast_lane_u32(uint32_t v, int j) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, j);
}

// v_permlane32_swap x, y: lanes 32..63 of x <-> lanes 0..31 of y.
__device__ __forceinline__ void swap32(float& x, float& y) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(x), __float_as_int(y), false, false);
  x = __int_as_float(r[0]);
  y = __int_as_float(r[1]);
}
// v_permlane16_swap x, y: odd rows (16-lane groups 1,3) of x <-> even rows (0,2) of y.
__device__ __forceinline__ void swap16(float& x, float& y) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(x), __float_as_int(y), false, false);
  x = __int_as_float(r[0]);
  y = __int_as_float(r[1]);
}

// Value of lane (l ^ D) for D in {1,2,4,8,16,32}: DPP within a row, permlane swaps across rows.
constexpr int DPP_QUAD_REVERSE = 0x1B;  // quad_perm:[3,2,1,0]
template <int D>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
  const int iv = (int)v;
  if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, iv, DPP_QUAD_XOR1, 0xf, 0xf, false);
  else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, iv, DPP_QUAD_XOR2, 0xf, 0xf, false);
  else if constexpr (D == 4) {  // l^4 = quad-reverse of the half-mirror (7 - l)
    const int t = __builtin_amdgcn_update_dpp(0, iv, DPP_ROW_HALF_MIRROR, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(0, t, DPP_QUAD_REVERSE, 0xf, 0xf, false);
  } else if constexpr (D == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, iv, DPP_ROW_ROR8, 0xf, 0xf, false);
  else if constexpr (D == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
    return (uint32_t)(((lane >> 4) & 1) ? r[0] : r[1]);
  } else {
    auto r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
    return (uint32_t)((lane & 32) ? r[0] : r[1]);
  }
}
template <int D>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v, int lane) {
  const uint32_t lo = lane_xor<D>((uint32_t)v, lane), hi = lane_xor<D>((uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

// ---- scans over each 32-lane half of the wave (lanes 0..31 and 32..63 independently) -----------------
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142;  // lane 15 of every row -> all lanes of the next row (use row_mask 0xA)

template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ float dpp_keep(float old, float v) {  // lanes without a source keep `old`
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
// Inclusive scans over the lanes of each 32-lane half, ONE asm block each (11 issue slots).  A DPP read needs two wait
// states after a VALU write of its SOURCE register only, so the first three steps -- which all shift the unmodified input
// %1 and fold it into the running value %0 (a plain operand: no hazard) -- issue back to back; after them lane i holds
// x[i-3..i] (within its row of 16), and shifts by 4 and 8 and the row broadcast finish the 32 lanes.  Lanes whose DPP
// source is out of range are not written (bound_ctrl off) and keep their value: exactly the scan's identity.  hipcc pads
// nothing inside asm and nothing between these instructions (separate asm statements each got an extra s_nop from it).
__device__ __forceinline__ float half_incl_scan_add(float v) {
  float r = v;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"
      : "+&v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ float half_incl_scan_mul(float v) {
  float r = v;
  asm volatile(
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"
      : "+&v"(r) : "v"(v));
  return r;
}
// Two independent inclusive scans in ONE asm block: the instructions of the two chains alternate, so each chain's DPP read finds
// its source written two issue slots earlier (the other chain's instruction + one s_nop 0) -- 12 DPP + 4 s_nop for two scans
// instead of 2 x (6 DPP + 4 s_nop 1).
__device__ __forceinline__ void half_incl_scan_mul2(float& a, float& b) {
  float ra = a, rb = b;
  asm volatile(
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %2, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %3, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %2, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %3, %1 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf"
      : "+&v"(ra), "+&v"(rb) : "v"(a), "v"(b));
  a = ra; b = rb;
}
__device__ __forceinline__ void half_incl_scan_add2(float& a, float& b) {
  float ra = a, rb = b;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %2, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %3, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %2, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %3, %1 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf"
      : "+&v"(ra), "+&v"(rb) : "v"(a), "v"(b));
  a = ra; b = rb;
}
// two exclusive prefix products (see half_excl_scan_mul)
__device__ __forceinline__ void half_excl_scan_mul2(float& a, float& b, int lane) {
  float sa = dpp_keep<DPP_ROW_SHR1>(1.f, a), sb = dpp_keep<DPP_ROW_SHR1>(1.f, b);
  const float pa = dpp_keep<DPP_ROW_BCAST15, 0xA>(1.f, a), pb = dpp_keep<DPP_ROW_BCAST15, 0xA>(1.f, b);
  const bool first = (lane & 15) == 0 && (lane & 16);
  sa = first ? pa : sa;
  sb = first ? pb : sb;
  half_incl_scan_mul2(sa, sb);
  a = sa; b = sb;
}

// exclusive prefix product over the lanes of each 32-lane half (lane 0 / 32 get 1): the input shifted by one lane, scanned
__device__ __forceinline__ float half_excl_scan_mul(float v, int lane) {
  float s = dpp_keep<DPP_ROW_SHR1>(1.f, v);                       // lane i <- v[i-1] inside a row
  const float prev_row_last = dpp_keep<DPP_ROW_BCAST15, 0xA>(1.f, v);  // rows 1,3 <- v[15], v[47]
  s = ((lane & 15) == 0 && (lane & 16)) ? prev_row_last : s;
  return half_incl_scan_mul(s);
}
// value of lane 31 (lanes 0..31) / lane 63 (lanes 32..63)
__device__ __forceinline__ float half_last(float v, int lane) {
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  return (lane & 32) ? b : a;
}

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n / 2); }
constexpr int next_pow2(int n) { int p = 1; while (p < n) p *= 2; return p; }

// Reference (slow) all-lanes sum through ds_bpermute shuffles -- used to validate the butterfly.
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Maximum over the 64 lanes, in every lane.  DPP / permlane butterflies: no LDS, and -- unlike __shfl_xor's ds_bpermute -- no
// per-lane address registers for the compiler to hoist out of the callers' loops and spill (round 3: four scratch reloads
// per chunk of the render backward, 4 600 cycles, were exactly that).
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
  const int lane = lane_id();
  uint32_t o;
  o = lane_xor<1>(v, lane); v = o > v ? o : v;
  o = lane_xor<2>(v, lane); v = o > v ? o : v;
  o = lane_xor<4>(v, lane); v = o > v ? o : v;
  o = lane_xor<8>(v, lane); v = o > v ? o : v;
  o = lane_xor<16>(v, lane); v = o > v ? o : v;
  o = lane_xor<32>(v, lane); v = o > v ? o : v;
  return v;
}

// Inclusive prefix sum over the 64 lanes (uint32): row_shr steps inside the rows of 16, row_bcast:15 / row_bcast:31 across them.
// Lanes without a DPP source keep their value (bound_ctrl off).  One asm block, like the float scans above.
__device__ __forceinline__ uint32_t wave_incl_scan_add_u32(uint32_t v) {
  uint32_t r = v;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_u32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_u32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+&v"(r) : "v"(v));
  return r;
}

}  // namespace mgs
