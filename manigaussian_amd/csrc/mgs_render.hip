// mgs_render.hip -- alpha-composite render, forward (K7) and backward (K8), for gfx950 wave64.
//
// WHAT is computed is the reference's per-pixel front-to-back blend and its gradient
// (RAST/cuda_rasterizer/forward.cu:262-398, backward.cu:399-593): same test order (power > 0 skip,
// alpha = min(0.99, o*exp(power)), alpha < 1/255 skip, stop before the Gaussian that would push
// T below 1e-4), RGB gets T*bg, features do not, n_contrib = position of the last blended entry.
//
// HOW is different (MI355X-first):
//  * execution granularity is one wave64 per 8x8 pixel block (4 independent workgroups per 16x16
//    tile, 4x the reference's parallelism at 128x128) -- tile MEMBERSHIP stays 16x16;
//  * each wave streams the tile's sorted, packed instance records 64 at a time (one coalesced 32-B
//    record per lane), culls them lane-parallel against its own 8x8 block with a conservative
//    bbox of {alpha >= 1/255} (the per-pixel test stays exact, so results are unchanged), and walks
//    only the survivors: geometry is broadcast with v_readlane, colour/feature rows through LDS;
//  * backward: the per-channel accum_rec recurrence collapses to one scalar per pixel
//    (A = accum_rec . dL_dpixel), and the 9+F per-Gaussian sums are reduced across the 64 pixels
//    with a transposing butterfly (v_permlane32/16_swap + DPP) that leaves each sum in its own
//    lane: one coalesced atomic instruction per Gaussian per wave instead of (9+F)*64 atomics.
#include "mgs_render_common.h"

namespace mgs {

// ------------------------------------------- forward ------------------------------------------------
template <int F, bool FAST>
__global__ void __launch_bounds__(64) render_fwd_kernel(RenderArgs r, const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ point_list,
                                                        const float4* __restrict__ inst,
                                                        float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                        float* __restrict__ out_color, float* __restrict__ out_feat) {
  constexpr int ROW4 = Row<F>::ROW4;
  __shared__ float4 stage[64 * ROW4];
  const int lane = threadIdx.x;
  int tile, sub;
  map_block(blockIdx.x, tile, sub);
  if (tile >= r.tiles_x * r.tiles_y) return;
  const int tx = tile % r.tiles_x, ty = tile / r.tiles_x;
  const int bx0 = tx * TILE + (sub & 1) * SUB, by0 = ty * TILE + (sub >> 1) * SUB;
  const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
  const bool inside = px < r.W && py < r.H;
  const float pxf = (float)px, pyf = (float)py;
  const float bxmin = (float)bx0, bxmax = (float)min(bx0 + SUB - 1, r.W - 1);
  const float bymin = (float)by0, bymax = (float)min(by0 + SUB - 1, r.H - 1);
  const uint2 rng = ranges[tile];
  const bool use_feat = (F > 0) && r.include_feature;

  float T = 1.0f;
  float C[3] = {0.f, 0.f, 0.f};
  float Fv[F > 0 ? F : 1];
#pragma unroll
  for (int i = 0; i < (F > 0 ? F : 1); i++) Fv[i] = 0.f;
  uint32_t last = 0;
  bool done = !inside;

  for (uint32_t k0 = rng.x; k0 < rng.y; k0 += 64) {
    if (ballot(!done) == 0) break;
    const uint32_t e = k0 + lane;
    const bool valid = e < rng.y;
    float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
    if (valid) { g0 = inst[2 * (size_t)e]; g1 = inst[2 * (size_t)e + 1]; }
    const bool surv = valid && g1.z >= 0.f && (g0.x + g1.z >= bxmin) && (g0.x - g1.z <= bxmax) &&
                      (g0.y + g1.w >= bymin) && (g0.y - g1.w <= bymax);
    unsigned long long mask = ballot(surv);
    if (mask == 0) continue;
    __syncthreads();
    if (surv) stage_row<F>(stage, lane, point_list[e], point_list[e], r.colors, use_feat ? r.feats : nullptr);
    __syncthreads();
    while (mask) {
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float ex = bcast_lane(g0.x, j), ey = bcast_lane(g0.y, j);
      const float cx = bcast_lane(g0.z, j), cy = bcast_lane(g0.w, j), cz = bcast_lane(g1.x, j);
      const float op = bcast_lane(g1.y, j);
      const float dx = ex - pxf, dy = ey - pyf;
      const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
      const float alpha = fminf(0.99f, op * exp_<FAST>(power));
      const bool skip = (power > 0.0f) || (alpha < 1.0f / 255.0f);
      const float test_T = T * (1.0f - alpha);
      const bool cand = !done && !skip;
      const bool term = cand && (test_T < 0.0001f);
      done = done || term;
      const bool blend = cand && !term;
      if (ballot(blend) == 0) continue;
      const float w = blend ? alpha * T : 0.f;
      const float4* row = stage + j * ROW4;
      if constexpr (F > 0) {
        if (use_feat) {
          if constexpr (F % 4 == 0) {
#pragma unroll
            for (int i = 0; i < F / 4; i++) {
              const float4 v = row[i];
              Fv[4 * i] += v.x * w; Fv[4 * i + 1] += v.y * w; Fv[4 * i + 2] += v.z * w; Fv[4 * i + 3] += v.w * w;
            }
          } else {
            const float* rf = reinterpret_cast<const float*>(row);
#pragma unroll
            for (int i = 0; i < F; i++) Fv[i] += rf[i] * w;
          }
        }
      }
      {
        const float* rf = reinterpret_cast<const float*>(row);
        C[0] += rf[F] * w; C[1] += rf[F + 1] * w; C[2] += rf[F + 2] * w;
      }
      T = blend ? test_T : T;
      last = blend ? (k0 - rng.x) + (uint32_t)j + 1u : last;
    }
  }
  if (inside) {
    const size_t HW = (size_t)r.H * r.W;
    const size_t pix = (size_t)py * r.W + px;
    final_T[pix] = T;
    n_contrib[pix] = last;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = C[ch] + T * r.bg[ch];
    if constexpr (F > 0) {
      if (use_feat) {
#pragma unroll
        for (int ch = 0; ch < F; ch++) out_feat[ch * HW + pix] = Fv[ch];
      }
    }
  }
}

// ------------------------------------------- backward -----------------------------------------------
template <int F, bool FAST, int RED>
__global__ void __launch_bounds__(64) render_bwd_kernel(RenderArgs r, const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ point_list,
                                                        const float4* __restrict__ inst,
                                                        const float* __restrict__ final_T,
                                                        const uint32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ dL_dpix,
                                                        const float* __restrict__ dL_dpix_F, float* __restrict__ acc8,
                                                        float* __restrict__ dL_dcolors, float* __restrict__ dL_dfeat) {
  constexpr int ROW4 = Row<F>::ROW4;
  constexpr int FP = F > 0 ? next_pow2(F) : 1;
  __shared__ float4 stage[64 * ROW4];
  const int lane = threadIdx.x;
  int tile, sub;
  map_block(blockIdx.x, tile, sub);
  if (tile >= r.tiles_x * r.tiles_y) return;
  const int tx = tile % r.tiles_x, ty = tile / r.tiles_x;
  const int bx0 = tx * TILE + (sub & 1) * SUB, by0 = ty * TILE + (sub >> 1) * SUB;
  const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
  const bool inside = px < r.W && py < r.H;
  const float pxf = (float)px, pyf = (float)py;
  const float bxmin = (float)bx0, bxmax = (float)min(bx0 + SUB - 1, r.W - 1);
  const float bymin = (float)by0, bymax = (float)min(by0 + SUB - 1, r.H - 1);
  const uint2 rng = ranges[tile];
  const bool use_feat = (F > 0) && r.include_feature;
  const size_t HW = (size_t)r.H * r.W;
  const size_t pix = (size_t)py * r.W + px;

  const float T_final = inside ? final_T[pix] : 0.f;
  const uint32_t last = inside ? n_contrib[pix] : 0u;
  float dLc[3] = {0.f, 0.f, 0.f};
  float dLf[F > 0 ? F : 1];
#pragma unroll
  for (int i = 0; i < (F > 0 ? F : 1); i++) dLf[i] = 0.f;
  if (inside) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++) dLc[ch] = dL_dpix[ch * HW + pix];
    if constexpr (F > 0) {
      if (use_feat) {
#pragma unroll
        for (int ch = 0; ch < F; ch++) dLf[ch] = dL_dpix_F[ch * HW + pix];
      }
    }
  }
  const float bgdot = r.bg[0] * dLc[0] + r.bg[1] * dLc[1] + r.bg[2] * dLc[2];
  const uint32_t kmax = wave_umax(last);
  if (kmax == 0) return;

  float T = T_final, A = 0.f, last_alpha = 0.f, last_D = 0.f;
  const float ddelx_dx = 0.5f * r.W, ddely_dy = 0.5f * r.H;  // backward.cu:476-477

  const int nb = (int)((kmax + 63u) / 64u);
  for (int bi = nb - 1; bi >= 0; --bi) {
    const uint32_t e = rng.x + (uint32_t)bi * 64u + lane;
    const uint32_t pos_l = (uint32_t)bi * 64u + lane + 1u;
    const bool valid = e < rng.y && pos_l <= kmax;
    float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
    if (valid) { g0 = inst[2 * (size_t)e]; g1 = inst[2 * (size_t)e + 1]; }
    const bool surv = valid && g1.z >= 0.f && (g0.x + g1.z >= bxmin) && (g0.x - g1.z <= bxmax) &&
                      (g0.y + g1.w >= bymin) && (g0.y - g1.w <= bymax);
    unsigned long long mask = ballot(surv);
    if (mask == 0) continue;
    uint32_t id_l = 0;
    __syncthreads();
    if (surv) {
      id_l = point_list[e];
      stage_row<F>(stage, lane, id_l, id_l, r.colors, use_feat ? r.feats : nullptr);
    }
    __syncthreads();
    while (mask) {
      const int j = 63 - __builtin_clzll(mask);  // back to front
      mask &= ~(1ull << j);
      const float ex = bcast_lane(g0.x, j), ey = bcast_lane(g0.y, j);
      const float cx = bcast_lane(g0.z, j), cy = bcast_lane(g0.w, j), cz = bcast_lane(g1.x, j);
      const float op = bcast_lane(g1.y, j);
      const uint32_t pos = (uint32_t)bi * 64u + (uint32_t)j + 1u;
      const float dx = ex - pxf, dy = ey - pyf;
      const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
      const float G = exp_<FAST>(power);
      const float alpha = fminf(0.99f, op * G);
      const bool active = inside && pos <= last && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
      if (ballot(active) == 0) continue;

      const float4* row = stage + j * ROW4;
      const float* rf = reinterpret_cast<const float*>(row);
      float D = rf[F] * dLc[0] + rf[F + 1] * dLc[1] + rf[F + 2] * dLc[2];
      if constexpr (F > 0) {
        if (use_feat) {
          if constexpr (F % 4 == 0) {
#pragma unroll
            for (int i = 0; i < F / 4; i++) {
              const float4 v = row[i];
              D += v.x * dLf[4 * i] + v.y * dLf[4 * i + 1] + v.z * dLf[4 * i + 2] + v.w * dLf[4 * i + 3];
            }
          } else {
#pragma unroll
            for (int i = 0; i < F; i++) D += rf[i] * dLf[i];
          }
        }
      }
      const float one_m = 1.f - alpha;
      const float Tn = T / one_m;                                    // backward.cu:521
      const float An = last_alpha * last_D + (1.f - last_alpha) * A;  // scalar form of accum_rec, :533,549
      float dL_dalpha = (D - An) * Tn;
      dL_dalpha += (-T_final / one_m) * bgdot;                       // :567-570
      if (active) { T = Tn; A = An; last_alpha = alpha; last_D = D; }
      const float wa = active ? alpha * Tn : 0.f;                     // dchannel_dcolor
      const float dL_dG = op * dL_dalpha;
      const float gdx = G * dx, gdy = G * dy;
      const float dG_ddelx = -gdx * cx - gdy * cy;
      const float dG_ddely = -gdy * cz - gdx * cy;
      float s[16];
      s[0] = active ? dL_dG * dG_ddelx * ddelx_dx : 0.f;
      s[1] = active ? dL_dG * dG_ddely * ddely_dy : 0.f;
      s[2] = active ? -0.5f * gdx * dx * dL_dG : 0.f;
      s[3] = active ? -0.5f * gdx * dy * dL_dG : 0.f;
      s[4] = active ? -0.5f * gdy * dy * dL_dG : 0.f;
      s[5] = active ? G * dL_dalpha : 0.f;
      s[6] = wa * dLc[0]; s[7] = wa * dLc[1]; s[8] = wa * dLc[2];
#pragma unroll
      for (int i = 9; i < 16; i++) s[i] = 0.f;
      const uint32_t id = bcast_lane_u32(id_l, j);

      if constexpr (RED == 1) {
        bfly_reduce<16>(s, lane);
        {
          const int idx = (lane >> 2) & 15;
          if ((lane & 3) == 0 && idx < 9) {
            float* dst = idx < 6 ? (acc8 + (size_t)id * 8 + idx) : (dL_dcolors + (size_t)id * 3 + (idx - 6));
            unsafeAtomicAdd(dst, s[0]);
          }
        }
        if constexpr (F > 0) {
          if (use_feat) {
            float f[FP];
#pragma unroll
            for (int i = 0; i < FP; i++) f[i] = (i < F) ? wa * dLf[i < F ? i : 0] : 0.f;
            bfly_reduce<FP>(f, lane);
            constexpr int SH = 6 - ilog2(FP);
            const int idx = (lane >> SH) & (FP - 1);
            if ((lane & ((1 << SH) - 1)) == 0 && idx < F) unsafeAtomicAdd(dL_dfeat + (size_t)id * F + idx, f[0]);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 9; i++) {
          const float t = wave_sum_shfl(s[i]);
          if (lane == 0) {
            float* dst = i < 6 ? (acc8 + (size_t)id * 8 + i) : (dL_dcolors + (size_t)id * 3 + (i - 6));
            unsafeAtomicAdd(dst, t);
          }
        }
        if constexpr (F > 0) {
          if (use_feat) {
#pragma unroll
            for (int i = 0; i < F; i++) {
              const float t = wave_sum_shfl(wa * dLf[i]);
              if (lane == 0) unsafeAtomicAdd(dL_dfeat + (size_t)id * F + i, t);
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------- dispatch ------------------------------------------------
static int render_grid(const RenderArgs& r) {
  const int T = r.tiles_x * r.tiles_y;
  return ((T + 7) / 8) * 32;
}

template <int F>
static hipError_t fwd_F(const RenderArgs& r, const BinView& b, const ImgView& im, float* oc, float* of, hipStream_t s) {
  const int grid = render_grid(r);
  if (r.fast_exp)
    hipLaunchKernelGGL((render_fwd_kernel<F, true>), dim3(grid), dim3(64), 0, s, r, im.ranges, b.point_list, b.inst,
                       im.final_T, im.n_contrib, oc, of);
  else
    hipLaunchKernelGGL((render_fwd_kernel<F, false>), dim3(grid), dim3(64), 0, s, r, im.ranges, b.point_list, b.inst,
                       im.final_T, im.n_contrib, oc, of);
  return hipGetLastError();
}

template <int F>
static hipError_t bwd_F(const RenderArgs& r, const BinView& b, const ImgView& im, const float* dc, const float* df,
                        float* acc8, float* dcol, float* dfeat, hipStream_t s) {
  const int grid = render_grid(r);
#define MGS_BWD(FAST, RED)                                                                                        \
  hipLaunchKernelGGL((render_bwd_kernel<F, FAST, RED>), dim3(grid), dim3(64), 0, s, r, im.ranges, b.point_list,   \
                     b.inst, im.final_T, im.n_contrib, dc, df, acc8, dcol, dfeat)
  if (r.bwd_reduce == 0) {
    if (r.fast_exp) MGS_BWD(true, 0); else MGS_BWD(false, 0);
  } else {
    if (r.fast_exp) MGS_BWD(true, 1); else MGS_BWD(false, 1);
  }
#undef MGS_BWD
  return hipGetLastError();
}

hipError_t launch_render_fwd(const RenderArgs& r, const BinView& b, const ImgView& im, float* out_color,
                             float* out_feat, hipStream_t s) {
  const int F = r.include_feature ? r.F : 0;
  switch (F) {
#define X(N) case N: return fwd_F<N>(r, b, im, out_color, out_feat, s);
    MGS_FOR_EACH_F(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_render_bwd(const RenderArgs& r, const BinView& b, const ImgView& im, const float* dL_dcolor_px,
                             const float* dL_dfeat_px, float* acc8, float* dL_dcolors, float* dL_dfeat,
                             hipStream_t s) {
  const int F = r.include_feature ? r.F : 0;
  switch (F) {
#define X(N) case N: return bwd_F<N>(r, b, im, dL_dcolor_px, dL_dfeat_px, acc8, dL_dcolors, dL_dfeat, s);
    MGS_FOR_EACH_F(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

// ------------------------------------------- self-test ------------------------------------------------
// Checks the cross-lane primitives against their definitions with small integers (exact in fp32).
template <int N>
__device__ int selftest_bfly(int lane) {
  float a[N];
#pragma unroll
  for (int i = 0; i < N; i++) a[i] = (float)(((lane * 7 + i * 13) % 11) - 5);
  bfly_reduce<N>(a, lane);
  constexpr int SH = 6 - ilog2(N);
  const int idx = (lane >> SH) & (N - 1);
  float expect = 0.f;
  for (int l = 0; l < 64; l++) expect += (float)(((l * 7 + idx * 13) % 11) - 5);
  return (a[0] == expect) ? 0 : 1;
}

__global__ void selftest_kernel(int* result) {
  const int lane = threadIdx.x;
  int bad = 0;
  {  // swap32: lanes 32..63 of x <-> lanes 0..31 of y
    float x = (float)lane, y = (float)(100 + lane);
    swap32(x, y);
    const float ex = lane < 32 ? (float)lane : (float)(100 + lane - 32);
    const float ey = lane < 32 ? (float)(lane + 32) : (float)(100 + lane);
    if (x != ex || y != ey) bad |= 1;
  }
  {  // swap16: odd rows of x <-> even rows of y
    float x = (float)lane, y = (float)(100 + lane);
    swap16(x, y);
    const int row = lane >> 4;
    const float ex = (row & 1) ? (float)(100 + lane - 16) : (float)lane;
    const float ey = (row & 1) ? (float)(100 + lane) : (float)(lane + 16);
    if (x != ex || y != ey) bad |= 2;
  }
  if (dpp_mov<DPP_ROW_ROR8>((float)lane) != (float)(lane ^ 8)) bad |= 4;
  if (dpp_mov<DPP_ROW_HALF_MIRROR>((float)lane) != (float)((lane & ~7) | (7 - (lane & 7)))) bad |= 8;
  if (dpp_mov<DPP_QUAD_XOR2>((float)lane) != (float)(lane ^ 2)) bad |= 16;
  if (dpp_mov<DPP_QUAD_XOR1>((float)lane) != (float)(lane ^ 1)) bad |= 32;
  if (selftest_bfly<64>(lane)) bad |= 64;
  if (selftest_bfly<32>(lane)) bad |= 128;
  if (selftest_bfly<16>(lane)) bad |= 256;
  if (selftest_bfly<4>(lane)) bad |= 512;
  if (selftest_bfly<1>(lane)) bad |= 1024;
  if (bcast_lane((float)lane, 37) != 37.f) bad |= 2048;
  if (wave_sum_shfl((float)lane) != 2016.f) bad |= 4096;
  if (wave_umax((uint32_t)lane * 3u) != 189u) bad |= 8192;
  {
    const uint32_t v = (uint32_t)lane * 2654435761u;
    if (lane_xor<1>(v, lane) != (uint32_t)(lane ^ 1) * 2654435761u) bad |= 1 << 14;
    if (lane_xor<2>(v, lane) != (uint32_t)(lane ^ 2) * 2654435761u) bad |= 1 << 15;
    if (lane_xor<4>(v, lane) != (uint32_t)(lane ^ 4) * 2654435761u) bad |= 1 << 16;
    if (lane_xor<8>(v, lane) != (uint32_t)(lane ^ 8) * 2654435761u) bad |= 1 << 17;
    if (lane_xor<16>(v, lane) != (uint32_t)(lane ^ 16) * 2654435761u) bad |= 1 << 18;
    if (lane_xor<32>(v, lane) != (uint32_t)(lane ^ 32) * 2654435761u) bad |= 1 << 19;
  }
  {  // half-wave scans (small integers: exact in float)
    const int n = lane & 31;
    const float a = (float)((lane * 5) % 7 + 1);
    float ea = 0.f;
    for (int i = 0; i <= n; i++) ea += (float)((((lane & 32) + i) * 5) % 7 + 1);
    if (half_incl_scan_add(a) != ea) bad |= 1 << 20;
    const float m = (lane % 3 == 0) ? 2.f : 1.f;
    float em = 1.f;
    for (int i = 0; i < n; i++) em *= (((lane & 32) + i) % 3 == 0) ? 2.f : 1.f;
    if (half_excl_scan_mul(m, lane) != em) bad |= 1 << 21;
    if (half_last((float)lane, lane) != ((lane & 32) ? 63.f : 31.f)) bad |= 1 << 22;
  }
  if (bad) atomicOr(result, bad);
}

hipError_t launch_selftest(int* result_dev, hipStream_t s) {
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, s, result_dev);
  return hipGetLastError();
}

}  // namespace mgs
