// mgs_binning.hip -- tile binning: inclusive scan of tile counts (K3), (tile|depth) key emission (K4),
// stable radix sort (K5), tile ranges + packed sorted instance records (K6).
//
// Follows RAST/cuda_rasterizer/rasterizer_impl.cu:70-138,280-320 for WHAT is produced (64-bit keys
// tile<<32 | depth bits, stable order, per-tile [start,end) ranges).  The scan and the radix sort come
// from rocPRIM through hipCUB exactly as the reference takes them from CUB.  New here: K6 also
// gathers the sorted per-instance record {xy, conic, opacity, cull extents} so that the render kernels
// stream it linearly (coalesced 32 B/lane) instead of chasing point_list -> means2D/conic_opacity.
#include <hipcub/hipcub.hpp>

#include "mgs_common.h"

namespace mgs {

size_t scan_temp_bytes(int P) {
  size_t bytes = 0;
  hipcub::DeviceScan::InclusiveSum(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, P);
  return bytes + 256;
}

size_t sort_temp_bytes(int R) {
  size_t bytes = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                     (uint32_t*)nullptr, R);
  return bytes + 256;
}

hipError_t launch_scan(const GeomView& g, int P, hipStream_t s) {
  if (P <= 0) return hipSuccess;
  size_t bytes = g.scan_temp_bytes;
  return hipcub::DeviceScan::InclusiveSum(g.scan_temp, bytes, g.tiles_touched, g.point_offsets, P, s);
}

__device__ __forceinline__ void get_rect_b(float px, float py, int rad, int gx, int gy, int& x0, int& y0, int& x1,
                                           int& y1) {
  x0 = min(gx, max(0, (int)((px - rad) / TILE)));
  y0 = min(gy, max(0, (int)((py - rad) / TILE)));
  x1 = min(gx, max(0, (int)((px + rad + TILE - 1) / TILE)));
  y1 = min(gy, max(0, (int)((py + rad + TILE - 1) / TILE)));
}

__global__ void __launch_bounds__(256) duplicate_with_keys_kernel(int P, const float2* __restrict__ means2D,
                                                                  const float* __restrict__ depths,
                                                                  const float2* __restrict__ cullext,
                                                                  const uint32_t* __restrict__ offsets,
                                                                  const int32_t* __restrict__ radii,
                                                                  uint64_t* __restrict__ keys,
                                                                  uint32_t* __restrict__ vals, int gx, int gy,
                                                                  int tight_bins) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P) return;
  const int rad = radii[idx];
  if (rad <= 0) return;
  uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
  const uint32_t end = offsets[idx];
  if (off == end) return;
  const float2 p = means2D[idx];
  int x0, y0, x1, y1;
  get_rect_b(p.x, p.y, rad, gx, gy, x0, y0, x1, y1);
  if (tight_bins) {
    const float2 h = cullext[idx];
    if (h.x < 0.f) return;
    const int tx0 = (int)ceilf((p.x - h.x - (TILE - 1)) / TILE), tx1 = (int)floorf((p.x + h.x) / TILE) + 1;
    const int ty0 = (int)ceilf((p.y - h.y - (TILE - 1)) / TILE), ty1 = (int)floorf((p.y + h.y) / TILE) + 1;
    x0 = max(x0, tx0); x1 = max(x0, min(x1, tx1));
    y0 = max(y0, ty0); y1 = max(y0, min(y1, ty1));
  }
  const uint32_t dbits = __float_as_uint(depths[idx]);
  for (int y = y0; y < y1; y++)
    for (int x = x0; x < x1; x++) {
      uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
      key <<= 32;
      key |= dbits;
      keys[off] = key;
      vals[off] = (uint32_t)idx;
      off++;
    }
}

// ranges (rasterizer_impl.cu:116-138) + gather of the packed sorted instance records.
__global__ void __launch_bounds__(256) ranges_gather_kernel(int L, const uint64_t* __restrict__ keys,
                                                            const uint32_t* __restrict__ point_list,
                                                            const float2* __restrict__ means2D,
                                                            const float4* __restrict__ conic_opacity,
                                                            const float2* __restrict__ cullext,
                                                            uint2* __restrict__ ranges, float4* __restrict__ inst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L) return;
  const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
  if (idx == 0)
    ranges[currtile].x = 0;
  else {
    const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
    if (currtile != prevtile) {
      ranges[prevtile].y = (uint32_t)idx;
      ranges[currtile].x = (uint32_t)idx;
    }
  }
  if (idx == L - 1) ranges[currtile].y = (uint32_t)L;
  const uint32_t id = point_list[idx];
  const float2 xy = means2D[id];
  const float4 co = conic_opacity[id];
  const float2 h = cullext[id];
  inst[2 * (size_t)idx] = make_float4(xy.x, xy.y, co.x, co.y);
  inst[2 * (size_t)idx + 1] = make_float4(co.z, co.w, h.x, h.y);
}

// rasterizer_impl.cu:35-50
static uint32_t higher_msb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4, step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

hipError_t launch_duplicate(const GeomView& g, const BinView& b, const ImgView& im, const int32_t* radii, int P, int R,
                            int tiles_x, int tiles_y, int tight_bins, hipStream_t s) {
  hipError_t e = hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)tiles_x * tiles_y, s);
  if (e != hipSuccess) return e;
  if (R <= 0 || P <= 0) return hipSuccess;
  hipLaunchKernelGGL(duplicate_with_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g.means2D, g.depths,
                     g.cullext, g.point_offsets, radii, b.keys_unsorted, b.vals_unsorted, tiles_x, tiles_y,
                     tight_bins);
  return hipGetLastError();
}

hipError_t launch_sort(const BinView& b, int R, int tiles_x, int tiles_y, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  const int bit = (int)higher_msb((uint32_t)(tiles_x * tiles_y));
  size_t bytes = b.sort_temp_bytes;
  return hipcub::DeviceRadixSort::SortPairs(b.sort_temp, bytes, b.keys_unsorted, b.keys, b.vals_unsorted, b.point_list,
                                            R, 0, 32 + bit, s);
}

hipError_t launch_ranges(const GeomView& g, const BinView& b, const ImgView& im, int R, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  hipLaunchKernelGGL(ranges_gather_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, b.keys, b.point_list, g.means2D,
                     g.conic_opacity, g.cullext, im.ranges, b.inst);
  return hipGetLastError();
}

}  // namespace mgs
