// One (pixel, Gaussian) pair through every plausible rounding of the reference's exponent expression (forward.cu:345-347):
//   hipcc -O3 --offload-arch=gfx950 scripts/diag/pair_probe.hip -o /tmp/pair_probe && /tmp/pair_probe x y cx cy cz o px py
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void probe(float x, float y, float cx, float cy, float cz, float o, float px, float py, float* out) {
  const float dx = x - px, dy = y - py;
  // (a) as the source is written, contraction left to the compiler (what the reference's build does)
  const float pa = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
  float pb, pc, pd, pe;
  {
#pragma clang fp contract(off)
    const float t = __builtin_fmaf(cx * dx, dx, (cz * dy) * dy);      // (b) the library's gauss_power
    pb = __builtin_fmaf(-0.5f, t, -((cy * dx) * dy));
    const float t2 = __builtin_fmaf(cz * dy, dy, (cx * dx) * dx);     // (c) the other product fused
    pc = __builtin_fmaf(-0.5f, t2, -((cy * dx) * dy));
    pd = __builtin_fmaf(-(cy * dx), dy, -0.5f * t);                   // (d) the last product fused instead
    pe = -0.5f * ((cx * dx) * dx + (cz * dy) * dy) - (cy * dx) * dy;  // (e) no fusion at all
  }
  const float p[5] = {pa, pb, pc, pd, pe};
  for (int i = 0; i < 5; i++) { out[2 * i] = p[i]; out[2 * i + 1] = o * expf(p[i]); }
}
int main(int argc, char** argv) {
  float v[8];
  for (int i = 0; i < 8; i++) { unsigned u = (unsigned)strtoul(argv[1 + i], nullptr, 16); memcpy(&v[i], &u, 4); }
  float* d; hipMalloc(&d, 40); float h[10];
  hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], d);
  hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
  const char* names[5] = {"source expression, compiler's contraction", "library gauss_power", "other product fused", "last product fused", "no fusion"};
  const float thr = 1.0f / 255.0f;
  for (int i = 0; i < 5; i++) { unsigned pb, ab; memcpy(&pb, &h[2 * i], 4); memcpy(&ab, &h[2 * i + 1], 4);
    printf("%-44s power %.9g (%08x) alpha %.9g (%08x) %s\n", names[i], h[2 * i], pb, h[2 * i + 1], ab, h[2 * i + 1] < thr ? "SKIPPED (< 1/255)" : "blended"); }
  unsigned tb; memcpy(&tb, &thr, 4); printf("1/255 = %.9g (%08x)\n", thr, tb);
  return 0;
}
