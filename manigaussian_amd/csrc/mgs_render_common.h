// mgs_render_common.h -- helpers shared by the render kernels (per-subtile and chunk-parallel variants).
#pragma once
#include "mgs_common.h"
#include "mgs_device.h"

namespace mgs {

template <int F>
struct Row {
  static constexpr int NCH = F + 3;
  static constexpr int ROW4 = (NCH + 3) / 4;  // float4 per staged row: [f0..fF-1, r, g, b, pad]
};

template <bool FAST>
__device__ __forceinline__ float exp_(float x) {
  if constexpr (FAST) return __expf(x);
  else return expf(x);
}

// block index -> (tile, sub-block).  Blocks b, b+8, b+16, b+24 (same XCD under the observed
// round-robin dispatch) work on the same tile, so the tile's instance list is fetched into one L2.
__device__ __forceinline__ void map_block(int b, int& tile, int& sub) {
  tile = (b / 32) * 8 + (b % 8);
  sub = (b / 8) % 4;
}

template <int F>
__device__ __forceinline__ void stage_row(float4* stage, int lane, uint32_t id, const float* __restrict__ colors,
                                          const float* __restrict__ feats) {
  constexpr int ROW4 = Row<F>::ROW4;
  float tmp[ROW4 * 4];
#pragma unroll
  for (int i = 0; i < ROW4 * 4; i++) tmp[i] = 0.f;
  if constexpr (F > 0) {
    if (feats) {
      if constexpr (F % 4 == 0) {
        const float4* src = reinterpret_cast<const float4*>(feats + (size_t)id * F);
#pragma unroll
        for (int i = 0; i < F / 4; i++) {
          const float4 v = src[i];
          tmp[4 * i] = v.x; tmp[4 * i + 1] = v.y; tmp[4 * i + 2] = v.z; tmp[4 * i + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < F; i++) tmp[i] = feats[(size_t)id * F + i];
      }
    }
  }
  tmp[F] = colors[(size_t)id * 3];
  tmp[F + 1] = colors[(size_t)id * 3 + 1];
  tmp[F + 2] = colors[(size_t)id * 3 + 2];
#pragma unroll
  for (int i = 0; i < ROW4; i++)
    stage[lane * ROW4 + i] = make_float4(tmp[4 * i], tmp[4 * i + 1], tmp[4 * i + 2], tmp[4 * i + 3]);
}


// conservative lane-parallel cull of one instance record against a pixel block [bxmin,bxmax]x[bymin,bymax]
__device__ __forceinline__ bool overlaps_block(const float4& g0, const float4& g1, float bxmin, float bxmax,
                                               float bymin, float bymax) {
  return g1.z >= 0.f && (g0.x + g1.z >= bxmin) && (g0.x - g1.z <= bxmax) && (g0.y + g1.w >= bymin) &&
         (g0.y - g1.w <= bymax);
}

// Feature widths compiled in.  Other widths are padded up by the host shim (zero channels change nothing).
#define MGS_FOR_EACH_F(X) X(0) X(3) X(4) X(8) X(16) X(32) X(64)

}  // namespace mgs
