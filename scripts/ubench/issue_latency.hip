// issue_latency.hip -- micro-benchmark (round 5): what one wave64 instruction costs on a gfx950 SIMD, dependent and independent,
// alone and with 1..4 waves per SIMD.  profiles/r05_floor.md models the render kernels with these numbers.
//   hipcc --offload-arch=gfx950 -O2 -o issue_latency issue_latency.hip && ./issue_latency
// Method: a workgroup of 64 * 4 * K threads puts K waves on every SIMD of one CU; every wave runs REP x 32 copies of the
// instruction pattern between two s_memtime reads (shader clock); reported: cycles per instruction for the slowest wave of the
// workgroup, i.e. per-SIMD cost with K waves interleaved.  "dep": each instruction consumes the previous one's result; "ind": eight
// independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R32(x) R8(x) R8(x) R8(x) R8(x)

constexpr int REP = 16;  // x 32 instructions (dep) or x 32 x 8 / 8 (ind: 8 chains x 4 per R32 ... see each case)

template <int CASE>
__global__ void lat(unsigned long long* out, float seed) {
  const int lane = threadIdx.x & 63;
  float a0 = seed + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0000001f, c = 1e-9f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
  __shared__ unsigned int chase[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) chase[i] = (unsigned)(((i * 17 + 5) & 1023) * 4);
  __syncthreads();
  unsigned int idx = (unsigned)lane * 4;
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  f32x16 acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  int sdummy = 0;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
    if constexpr (CASE == 0) {  // v_fma_f32 dependent
      asm volatile(R32("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
    } else if constexpr (CASE == 1) {  // v_fma_f32, 8 independent chains (32 instructions)
      asm volatile(R4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if constexpr (CASE == 2) {  // v_pk_fma_f32 dependent
      asm volatile(R32("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(p0) : "v"(pb), "v"(pc));
    } else if constexpr (CASE == 3) {  // v_pk_fma_f32, 4 independent chains (32 instructions)
      asm volatile(R8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
    } else if constexpr (CASE == 4) {  // v_exp_f32 dependent
      asm volatile(R32("v_exp_f32 %0, %0\n") : "+v"(a0));
    } else if constexpr (CASE == 5) {  // v_exp_f32, 8 independent
      asm volatile(R4("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
                      "v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (CASE == 6) {  // DPP row_shr:1 add, dependent (the scans' step), with the assembler-required wait states
      asm volatile(R32("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a0));
    } else if constexpr (CASE == 7) {  // v_cmp -> v_cndmask through vcc, dependent
      asm volatile(R32("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n") : "+v"(a0) : "v"(b), "v"(c) : "vcc");
    } else if constexpr (CASE == 8) {  // v_readlane -> s_add -> v_add (VALU -> SGPR -> SALU -> VALU), dependent
      asm volatile(R32("v_readlane_b32 %1, %0, 63\n s_add_i32 %1, %1, 1\n v_add_u32 %0, %1, %0\n") : "+v"(idx), "+s"(sdummy) : : "scc");  // (s_add writes SCC: the first
      // version of this file did not say so, the loop's own compare lost its SCC and the kernel never ended)
    } else if constexpr (CASE == 9) {  // ds_read_b32 pointer chase (dependent LDS latency)
      asm volatile(R32("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(idx)::"memory");
    } else if constexpr (CASE == 10) {  // v_mfma_f32_32x32x2_f32, dependent accumulator
      asm volatile(R32("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n") : "+v"(acc) : "v"(a1), "v"(a2));
    } else if constexpr (CASE == 11) {  // v_permlane32_swap + dependent mul (the forward's alpha exchange)
      asm volatile(R32("v_permlane32_swap_b32 %0, %1\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n") : "+v"(a0), "+v"(a1) : "v"(b));
    } else if constexpr (CASE == 12) {  // v_rcp_f32 dependent
      asm volatile(R32("v_rcp_f32 %0, %0\n") : "+v"(a0));
    } else if constexpr (CASE == 13) {  // v_pk_mul_f32 dependent on a v_pk_fma (mixed packed chain)
      asm volatile(R32("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2\n") : "+v"(p0) : "v"(pb), "v"(pc));
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + acc[0] + (float)idx + (float)sdummy;
  if (sink == 12345.678f) out[1023] = 1;  // keep everything alive
  if (lane == 0) out[blockIdx.x * 64 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Case { int id; const char* name; int instr_per_r32; };

template <int CASE>
static double run(int waves_per_simd, int instr_per_rep) {
  unsigned long long* d;
  hipMalloc(&d, 1024 * sizeof(unsigned long long));
  hipMemset(d, 0, 1024 * sizeof(unsigned long long));
  const int threads = 64 * 4 * waves_per_simd;
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(lat<CASE>, dim3(1), dim3(threads), 0, 0, d, 1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(64);
  hipMemcpy(h.data(), d, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  hipFree(d);
  unsigned long long mx = 0;
  for (int w = 0; w < threads / 64; w++) mx = h[w] > mx ? h[w] : mx;
  return (double)mx / (double)(REP * instr_per_rep);
}

#define ROW(C, NAME, N)                                                                                     \
  printf("%-62s", NAME);                                                                                    \
  for (int k = 1; k <= 4; k++) printf("  %7.2f", run<C>(k, N));                                             \
  printf("\n");

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  printf("start\n");
  printf("cycles per instruction of the pattern (s_memtime / shader clock), slowest wave, K waves per SIMD on one CU\n");
  printf("%-62s  %7s  %7s  %7s  %7s\n", "pattern", "K=1", "K=2", "K=3", "K=4");
  ROW(0, "v_fma_f32, dependent", 32)
  ROW(1, "v_fma_f32, 8 independent chains", 32)
  ROW(2, "v_pk_fma_f32, dependent", 32)
  ROW(3, "v_pk_fma_f32, 4 independent chains", 32)
  ROW(4, "v_exp_f32, dependent", 32)
  ROW(5, "v_exp_f32, 8 independent chains", 32)
  ROW(12, "v_rcp_f32, dependent", 32)
  ROW(6, "s_nop 1 + v_add_f32_dpp row_shr:1, dependent (per pair)", 32)
  ROW(7, "v_cmp -> vcc -> v_cndmask, dependent (per pair)", 32)
  ROW(8, "v_readlane -> s_add -> v_add, dependent (per triple)", 32)
  ROW(9, "ds_read_b32 pointer chase + s_waitcnt (per read)", 32)
  ROW(10, "v_mfma_f32_32x32x2_f32, dependent accumulator", 32)
  ROW(11, "v_permlane32_swap + 2 dependent v_mul (per triple)", 32)
  ROW(13, "v_pk_mul_f32 -> v_pk_add_f32, dependent (per pair)", 32)
  return 0;
}
