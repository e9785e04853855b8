// Microbenchmark (diagnostic, not part of the library): throughput of no-return float atomics to L2 in the shapes the render
// backward's epilogue issues them.  hipcc --offload-arch=gfx950 -O3 atomics.hip -o atomics && ./atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: feature rows (2 rows x 32 floats per instruction); 1: acc8 rows (8 rows x 8 floats, 6 used); 2: colour rows (16 x 3)
// share: waves that hit the same rows at the same time (1 = none)
template <int MODE>
__global__ void __launch_bounds__(1024) k_atomics(float* buf, uint32_t P, int iters, int share, int spin) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float v = 1.0f;
  for (int it = 0; it < iters; it++) {
    const uint32_t key = mix((wave / (uint32_t)share) * 977u + (uint32_t)it * 131071u);
    if (MODE == 0) {
      const uint32_t row = mix(key + (lane >> 5)) % P;
      unsafeAtomicAdd(buf + (size_t)row * 32 + (lane & 31), v);
    } else if (MODE == 1) {
      const uint32_t row = mix(key + (lane >> 3)) % P;
      if ((lane & 7) < 6) unsafeAtomicAdd(buf + (size_t)row * 8 + (lane & 7), v);
    } else {
      const uint32_t row = mix(key + (lane >> 2)) % P;
      if ((lane & 3) < 3) unsafeAtomicAdd(buf + (size_t)row * 3 + (lane & 3), v);
    }
    for (int s = 0; s < spin; s++) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);  // VALU work between the atomics
  }
  if (v == 123.456f) buf[0] = v;
}

template <int MODE>
static void run(const char* name, float* buf, uint32_t P, int blocks, int threads, int iters, int share, int spin) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k_atomics<MODE><<<blocks, threads>>>(buf, P, iters, share, spin);
  hipDeviceSynchronize();
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; i++) k_atomics<MODE><<<blocks, threads>>>(buf, P, iters, share, spin);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1000.0 / reps;
  const double instr = (double)blocks * (threads / 64) * iters;
  const double lanes = instr * (MODE == 0 ? 64 : 48);
  printf("%-10s blocks %4d waves/blk %2d iters %3d share %d spin %4d: %8.2f us  %7.1f k instr  %6.1f ns/instr/CU-serial  %6.2f G lane-adds/s\n",
         name, blocks, threads / 64, iters, share, spin, us, instr / 1e3, us * 1e3 / (instr / 256.0), lanes / us / 1e3);
}

int main() {
  const uint32_t P = 100000;
  float* buf;
  hipMalloc(&buf, (size_t)P * 32 * 4);
  hipMemset(buf, 0, (size_t)P * 32 * 4);
  for (int spin : {0, 2000}) {
    for (int share : {1, 4}) {
      run<0>("feature", buf, P, 256, 512, 32, share, spin);
      run<1>("acc8", buf, P, 256, 512, 8, share, spin);
      run<2>("colour", buf, P, 256, 512, 4, share, spin);
    }
  }
  run<0>("feature", buf, P, 256, 512, 320, 1, 0);
  run<1>("acc8", buf, P, 256, 512, 80, 1, 0);
  run<2>("colour", buf, P, 256, 512, 40, 1, 0);
  run<0>("feature", buf, P, 256, 1024, 160, 1, 0);
  run<0>("feature", buf, P, 1024, 1024, 80, 1, 0);
  return 0;
}
