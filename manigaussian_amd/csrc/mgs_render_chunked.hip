// mgs_render_chunked.hip -- chunk-parallel alpha-composite render (forward + backward) for gfx950.
//
// Same results as mgs_render.hip (and therefore the reference's renderCUDA fwd/bwd,
// RAST/cuda_rasterizer/forward.cu:262-398, backward.cu:399-593); different decomposition:
// a 128x128 image has only 64 tiles, so "one workgroup per tile" (the reference) or even "one wave per
// 8x8 block" leaves most of the 256 CUs / 1024 SIMDs idle and every latency exposed.  Here each tile's
// depth-sorted list is cut into chunks of `CH` entries and the work item is (tile, 8x8 block, chunk):
// thousands of independent waves.  Compositing is associative in (C, T), so the sequential dependency
// along the list reduces to a per-pixel transmittance prefix over chunks:
//
//   K7a transmittance : per item, Tprod = prod over the chunk's passing entries of (1 - alpha)
//   K7b blend         : T_in = prod_{c' < c} Tprod[c'] (same order everywhere, so every consumer sees
//                       bit-identical values); pixels with T_in < 1e-4 are finished (the reference's stop
//                       rule is monotone in T); the rest walk the chunk with the exact reference test order
//                       and write a partial colour/feature sum, T_end and the last blended position
//   K7c combine       : per pixel, sums the partials of the visited chunks, writes image, final_T and the
//                       index of the last visited chunk
//   K8a q-dot         : per item, q = dL_dpixel . partial  (what the later chunks contribute to this
//                       pixel's gradient state)
//   K8  backward      : per item, starts from T_end and A = (sum of q over later visited chunks) / T_end
//                       -- the reference's accum_rec state at the chunk boundary, in its scalar form -- and
//                       walks the chunk back to front exactly like the per-block kernel.
#include "mgs_render_common.h"

namespace mgs {

// ---- work table ---------------------------------------------------------------------------------
// chunk_base[t] = number of chunks in tiles < t; chunk_base[T] = total chunks NC; chunk_base[T+1] = items per XCD.
__global__ void __launch_bounds__(256) chunk_table_kernel(const uint2* __restrict__ ranges, int T, int CH,
                                                          uint32_t* __restrict__ chunk_base) {
  __shared__ uint32_t warp_sums[4];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + tid;
    uint32_t n = 0;
    if (t < T) { const uint2 r = ranges[t]; n = (r.y - r.x + (uint32_t)CH - 1u) / (uint32_t)CH; }
    uint32_t incl = n;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) warp_sums[wid] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += warp_sums[w];
    const uint32_t carry = carry_s;
    if (t < T) chunk_base[t] = carry + woff + incl - n;
    __syncthreads();
    if (tid == 255) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t NC = carry_s;
    chunk_base[T] = NC;
    chunk_base[T + 1] = 4u * ((NC + 7u) / 8u);  // items per XCD (multiple of 4: a chunk's 4 blocks stay together)
  }
}

struct Item {
  int tile, chunk, sub;
  uint32_t item;      // global item index = (chunk_base[tile] + chunk) * 4 + sub
  uint32_t base_item; // item index of chunk 0 of this (tile, sub)
  uint32_t nch;       // chunks in this tile
};

// block -> item.  Blocks b, b+8, b+16, ... (one XCD under the observed round-robin dispatch) take a contiguous
// range of items, so the four 8x8 blocks of a chunk and neighbouring chunks of a tile share one L2.
__device__ __forceinline__ bool find_item(const uint32_t* __restrict__ chunk_base, int T, Item& it) {
  const uint32_t per_xcd = chunk_base[T + 1];
  const uint32_t NC = chunk_base[T];
  const uint32_t b = blockIdx.x;
  const uint32_t slot = b >> 3;
  if (slot >= per_xcd) return false;
  const uint32_t i = (b & 7u) * per_xcd + slot;
  if (i >= 4u * NC) return false;
  const uint32_t g = i >> 2;
  int lo = 0, hi = T;  // largest t with chunk_base[t] <= g
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_base[mid] <= g) lo = mid; else hi = mid;
  }
  it.tile = lo;
  it.chunk = (int)(g - chunk_base[lo]);
  it.sub = (int)(i & 3u);
  it.item = i;
  it.base_item = chunk_base[lo] * 4u + (i & 3u);
  it.nch = chunk_base[lo + 1] - chunk_base[lo];
  return true;
}

struct PixelBlock {
  int px, py;
  bool inside;
  float pxf, pyf, bxmin, bxmax, bymin, bymax;
};
__device__ __forceinline__ PixelBlock pixel_block(const RenderArgs& r, int tile, int sub, int lane) {
  PixelBlock p;
  const int tx = tile % r.tiles_x, ty = tile / r.tiles_x;
  const int bx0 = tx * TILE + (sub & 1) * SUB, by0 = ty * TILE + (sub >> 1) * SUB;
  p.px = bx0 + (lane & 7);
  p.py = by0 + (lane >> 3);
  p.inside = p.px < r.W && p.py < r.H;
  p.pxf = (float)p.px; p.pyf = (float)p.py;
  p.bxmin = (float)bx0; p.bxmax = (float)min(bx0 + SUB - 1, r.W - 1);
  p.bymin = (float)by0; p.bymax = (float)min(by0 + SUB - 1, r.H - 1);
  return p;
}

// ---- K7a: per-chunk transmittance products -------------------------------------------------------
template <bool FAST>
__global__ void __launch_bounds__(64) chunk_transmittance_kernel(RenderArgs r, int CH, const uint2* __restrict__ ranges,
                                                                 const float4* __restrict__ inst,
                                                                 const uint32_t* __restrict__ chunk_base,
                                                                 float* __restrict__ Tprod) {
  Item it;
  if (!find_item(chunk_base, r.tiles_x * r.tiles_y, it)) return;
  const int lane = threadIdx.x;
  const PixelBlock p = pixel_block(r, it.tile, it.sub, lane);
  const uint2 rng = ranges[it.tile];
  const uint32_t e0 = rng.x + (uint32_t)it.chunk * (uint32_t)CH;
  const uint32_t e1 = min(e0 + (uint32_t)CH, rng.y);
  float Tp = 1.0f;
  for (uint32_t k0 = e0; k0 < e1; k0 += 64) {
    const uint32_t e = k0 + lane;
    const bool valid = e < e1;
    float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
    if (valid) { g0 = inst[2 * (size_t)e]; g1 = inst[2 * (size_t)e + 1]; }
    unsigned long long mask = ballot(valid && overlaps_block(g0, g1, p.bxmin, p.bxmax, p.bymin, p.bymax));
    while (mask) {
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float ex = bcast_lane(g0.x, j), ey = bcast_lane(g0.y, j);
      const float cx = bcast_lane(g0.z, j), cy = bcast_lane(g0.w, j), cz = bcast_lane(g1.x, j);
      const float op = bcast_lane(g1.y, j);
      const float dx = ex - p.pxf, dy = ey - p.pyf;
      const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
      const float alpha = fminf(0.99f, op * exp_<FAST>(power));
      const bool skip = (power > 0.0f) || (alpha < 1.0f / 255.0f);
      Tp = skip ? Tp : Tp * (1.0f - alpha);
    }
  }
  Tprod[(size_t)it.item * 64 + lane] = Tp;
}

// ---- K7b: blend one chunk ---------------------------------------------------------------------------
template <int F, bool FAST>
__global__ void __launch_bounds__(64) chunk_blend_kernel(RenderArgs r, int CH, const uint2* __restrict__ ranges,
                                                         const uint32_t* __restrict__ point_list,
                                                         const float4* __restrict__ inst,
                                                         const uint32_t* __restrict__ chunk_base,
                                                         const float* __restrict__ Tprod, float* __restrict__ T_end,
                                                         uint32_t* __restrict__ last_pos, float* __restrict__ partial) {
  constexpr int ROW4 = Row<F>::ROW4;
  constexpr int NCH = F + 3;
  __shared__ float4 stage[64 * ROW4];
  Item it;
  if (!find_item(chunk_base, r.tiles_x * r.tiles_y, it)) return;
  const int lane = threadIdx.x;
  const PixelBlock p = pixel_block(r, it.tile, it.sub, lane);
  const bool use_feat = (F > 0) && r.include_feature;

  float T = 1.0f;  // transmittance entering this chunk: same multiplication order as K7c
  for (int c = 0; c < it.chunk; c++) T *= Tprod[((size_t)it.base_item + 4u * (uint32_t)c) * 64 + lane];
  bool done = !p.inside || (T < 0.0001f);
  if (ballot(!done) == 0) return;  // every pixel of the block finished in an earlier chunk: nothing is written

  const uint2 rng = ranges[it.tile];
  const uint32_t e0 = rng.x + (uint32_t)it.chunk * (uint32_t)CH;
  const uint32_t e1 = min(e0 + (uint32_t)CH, rng.y);
  float C[3] = {0.f, 0.f, 0.f};
  float Fv[F > 0 ? F : 1];
#pragma unroll
  for (int i = 0; i < (F > 0 ? F : 1); i++) Fv[i] = 0.f;
  uint32_t last = 0;

  for (uint32_t k0 = e0; k0 < e1; k0 += 64) {
    if (ballot(!done) == 0) break;
    const uint32_t e = k0 + lane;
    const bool valid = e < e1;
    float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
    if (valid) { g0 = inst[2 * (size_t)e]; g1 = inst[2 * (size_t)e + 1]; }
    const bool surv = valid && overlaps_block(g0, g1, p.bxmin, p.bxmax, p.bymin, p.bymax);
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
      const float dx = ex - p.pxf, dy = ey - p.pyf;
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
      last = blend ? (k0 - e0) + (uint32_t)j + 1u : last;
    }
  }
  const size_t o = (size_t)it.item * 64 + lane;
  T_end[o] = T;
  last_pos[o] = last;
  float* pp = partial + (size_t)it.item * NCH * 64 + lane;
#pragma unroll
  for (int ch = 0; ch < 3; ch++) pp[ch * 64] = C[ch];
  if constexpr (F > 0) {
#pragma unroll
    for (int ch = 0; ch < F; ch++) pp[(3 + ch) * 64] = Fv[ch];
  }
}

// ---- K7c: combine chunks into the image ---------------------------------------------------------------
template <int F>
__global__ void __launch_bounds__(64) chunk_combine_kernel(RenderArgs r, const uint32_t* __restrict__ chunk_base,
                                                           const float* __restrict__ Tprod,
                                                           const float* __restrict__ T_end,
                                                           const float* __restrict__ partial,
                                                           float* __restrict__ final_T, uint32_t* __restrict__ last_chunk,
                                                           float* __restrict__ out_color, float* __restrict__ out_feat) {
  constexpr int NCH = F + 3;
  const int lane = threadIdx.x;
  int tile, sub;
  map_block(blockIdx.x, tile, sub);
  if (tile >= r.tiles_x * r.tiles_y) return;
  const PixelBlock p = pixel_block(r, tile, sub, lane);
  const bool use_feat = (F > 0) && r.include_feature;
  const uint32_t base = chunk_base[tile] * 4u + (uint32_t)sub;
  const uint32_t nch = chunk_base[tile + 1] - chunk_base[tile];
  float T = 1.0f, Tf = 1.0f;
  float acc[NCH];
#pragma unroll
  for (int i = 0; i < NCH; i++) acc[i] = 0.f;
  uint32_t visited = 0;
  for (uint32_t c = 0; c < nch; c++) {
    const bool act = p.inside && !(T < 0.0001f);
    if (ballot(act) == 0) break;
    const size_t item = (size_t)base + 4u * c;
    if (act) {
      const float* pp = partial + item * NCH * 64 + lane;
#pragma unroll
      for (int i = 0; i < NCH; i++) acc[i] += pp[i * 64];
      Tf = T_end[item * 64 + lane];
      visited = c + 1;
      T *= Tprod[item * 64 + lane];
    }
  }
  last_chunk[((size_t)tile * 4 + sub) * 64 + lane] = visited;
  if (p.inside) {
    const size_t HW = (size_t)r.H * r.W;
    const size_t pix = (size_t)p.py * r.W + p.px;
    final_T[pix] = Tf;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = acc[ch] + Tf * r.bg[ch];
    if constexpr (F > 0) {
      if (use_feat) {
#pragma unroll
        for (int ch = 0; ch < F; ch++) out_feat[ch * HW + pix] = acc[3 + ch];
      }
    }
  }
}

// ---- K8a: q = dL_dpixel . partial per (item, pixel) ---------------------------------------------------
template <int F>
__global__ void __launch_bounds__(64) chunk_qdot_kernel(RenderArgs r, const uint32_t* __restrict__ chunk_base,
                                                        const uint32_t* __restrict__ last_chunk,
                                                        const float* __restrict__ partial,
                                                        const float* __restrict__ dL_dpix,
                                                        const float* __restrict__ dL_dpix_F, float* __restrict__ q) {
  constexpr int NCH = F + 3;
  Item it;
  if (!find_item(chunk_base, r.tiles_x * r.tiles_y, it)) return;
  const int lane = threadIdx.x;
  const uint32_t lc = last_chunk[((size_t)it.tile * 4 + it.sub) * 64 + lane];
  const bool visited = (uint32_t)it.chunk < lc;
  if (ballot(visited) == 0) return;
  const PixelBlock p = pixel_block(r, it.tile, it.sub, lane);
  const bool use_feat = (F > 0) && r.include_feature;
  float s = 0.f;
  if (visited) {
    const size_t HW = (size_t)r.H * r.W;
    const size_t pix = (size_t)p.py * r.W + p.px;
    const float* pp = partial + (size_t)it.item * NCH * 64 + lane;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) s += dL_dpix[ch * HW + pix] * pp[ch * 64];
    if constexpr (F > 0) {
      if (use_feat) {
#pragma unroll
        for (int ch = 0; ch < F; ch++) s += dL_dpix_F[ch * HW + pix] * pp[(3 + ch) * 64];
      }
    }
  }
  q[(size_t)it.item * 64 + lane] = s;
}

// ---- K8: backward over one chunk -----------------------------------------------------------------------
template <int F, bool FAST, int RED>
__global__ void __launch_bounds__(64) chunk_bwd_kernel(RenderArgs r, int CH, const uint2* __restrict__ ranges,
                                                       const uint32_t* __restrict__ point_list,
                                                       const float4* __restrict__ inst,
                                                       const uint32_t* __restrict__ chunk_base,
                                                       const uint32_t* __restrict__ last_chunk,
                                                       const float* __restrict__ T_end,
                                                       const uint32_t* __restrict__ last_pos, const float* __restrict__ q,
                                                       const float* __restrict__ final_T, const float* __restrict__ dL_dpix,
                                                       const float* __restrict__ dL_dpix_F, float* __restrict__ acc8,
                                                       float* __restrict__ dL_dcolors, float* __restrict__ dL_dfeat) {
  constexpr int ROW4 = Row<F>::ROW4;
  constexpr int FP = F > 0 ? next_pow2(F) : 1;
  __shared__ float4 stage[64 * ROW4];
  Item it;
  if (!find_item(chunk_base, r.tiles_x * r.tiles_y, it)) return;
  const int lane = threadIdx.x;
  const uint32_t lc = last_chunk[((size_t)it.tile * 4 + it.sub) * 64 + lane];
  const bool visited = (uint32_t)it.chunk < lc;
  if (ballot(visited) == 0) return;
  const size_t o = (size_t)it.item * 64 + lane;
  const uint32_t last = visited ? last_pos[o] : 0u;
  const uint32_t kmax = wave_umax(last);
  if (kmax == 0) return;
  const PixelBlock p = pixel_block(r, it.tile, it.sub, lane);
  const bool use_feat = (F > 0) && r.include_feature;
  const size_t HW = (size_t)r.H * r.W;
  const size_t pix = (size_t)p.py * r.W + p.px;
  const bool live = last > 0;  // this pixel blended something in this chunk

  const float T_final = live ? final_T[pix] : 0.f;
  float dLc[3] = {0.f, 0.f, 0.f};
  float dLf[F > 0 ? F : 1];
#pragma unroll
  for (int i = 0; i < (F > 0 ? F : 1); i++) dLf[i] = 0.f;
  if (live) {
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

  // state at the chunk boundary: T after this chunk's last blend; A = accum_rec . dL as seen from there
  float T = live ? T_end[o] : 1.0f;
  float B = 0.f;
  {
    const uint32_t lcmax = wave_umax(lc);
    for (uint32_t c = (uint32_t)it.chunk + 1u; c < lcmax; c++) {
      const float v = q[((size_t)it.base_item + 4u * c) * 64 + lane];
      B += (live && c < lc) ? v : 0.f;
    }
  }
  float A = live ? B / T : 0.f, last_alpha = 0.f, last_D = 0.f;
  const float ddelx_dx = 0.5f * r.W, ddely_dy = 0.5f * r.H;

  const uint2 rng = ranges[it.tile];
  const uint32_t e0 = rng.x + (uint32_t)it.chunk * (uint32_t)CH;
  const int nb = (int)((kmax + 63u) / 64u);
  for (int bi = nb - 1; bi >= 0; --bi) {
    const uint32_t e = e0 + (uint32_t)bi * 64u + lane;
    const uint32_t pos_l = (uint32_t)bi * 64u + lane + 1u;
    const bool valid = e < rng.y && pos_l <= kmax;
    float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
    if (valid) { g0 = inst[2 * (size_t)e]; g1 = inst[2 * (size_t)e + 1]; }
    const bool surv = valid && overlaps_block(g0, g1, p.bxmin, p.bxmax, p.bymin, p.bymax);
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
      const int j = 63 - __builtin_clzll(mask);
      mask &= ~(1ull << j);
      const float ex = bcast_lane(g0.x, j), ey = bcast_lane(g0.y, j);
      const float cx = bcast_lane(g0.z, j), cy = bcast_lane(g0.w, j), cz = bcast_lane(g1.x, j);
      const float op = bcast_lane(g1.y, j);
      const uint32_t pos = (uint32_t)bi * 64u + (uint32_t)j + 1u;
      const float dx = ex - p.pxf, dy = ey - p.pyf;
      const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
      const float G = exp_<FAST>(power);
      const float alpha = fminf(0.99f, op * G);
      const bool active = pos <= last && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
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
      const float Tn = T / one_m;
      const float An = last_alpha * last_D + (1.f - last_alpha) * A;
      float dL_dalpha = (D - An) * Tn;
      dL_dalpha += (-T_final / one_m) * bgdot;
      if (active) { T = Tn; A = An; last_alpha = alpha; last_D = D; }
      const float wa = active ? alpha * Tn : 0.f;
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

// ---- dispatch ---------------------------------------------------------------------------------------------
static int items_grid(const ChunkView& cv) { return 8 * (4 * ((cv.max_chunks + 7) / 8)); }

template <int F>
static hipError_t fwd_F(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv, float* oc,
                        float* of, hipStream_t s) {
  const int T = r.tiles_x * r.tiles_y;
  hipLaunchKernelGGL(chunk_table_kernel, dim3(1), dim3(256), 0, s, im.ranges, T, cv.CH, cv.chunk_base);
  const int grid = items_grid(cv);
  if (r.fast_exp) {
    hipLaunchKernelGGL((chunk_transmittance_kernel<true>), dim3(grid), dim3(64), 0, s, r, cv.CH, im.ranges, b.inst,
                       cv.chunk_base, cv.Tprod);
    hipLaunchKernelGGL((chunk_blend_kernel<F, true>), dim3(grid), dim3(64), 0, s, r, cv.CH, im.ranges, b.point_list,
                       b.inst, cv.chunk_base, cv.Tprod, cv.T_end, cv.last_pos, cv.partial);
  } else {
    hipLaunchKernelGGL((chunk_transmittance_kernel<false>), dim3(grid), dim3(64), 0, s, r, cv.CH, im.ranges, b.inst,
                       cv.chunk_base, cv.Tprod);
    hipLaunchKernelGGL((chunk_blend_kernel<F, false>), dim3(grid), dim3(64), 0, s, r, cv.CH, im.ranges, b.point_list,
                       b.inst, cv.chunk_base, cv.Tprod, cv.T_end, cv.last_pos, cv.partial);
  }
  hipLaunchKernelGGL((chunk_combine_kernel<F>), dim3(((T + 7) / 8) * 32), dim3(64), 0, s, r, cv.chunk_base, cv.Tprod,
                     cv.T_end, cv.partial, im.final_T, cv.last_chunk, oc, of);
  return hipGetLastError();
}

template <int F>
static hipError_t bwd_F(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv, const float* dc,
                        const float* df, float* acc8, float* dcol, float* dfeat, hipStream_t s) {
  const int grid = items_grid(cv);
  hipLaunchKernelGGL((chunk_qdot_kernel<F>), dim3(grid), dim3(64), 0, s, r, cv.chunk_base, cv.last_chunk, cv.partial, dc,
                     df, cv.q);
#define MGS_CBWD(FAST, RED)                                                                                          \
  hipLaunchKernelGGL((chunk_bwd_kernel<F, FAST, RED>), dim3(grid), dim3(64), 0, s, r, cv.CH, im.ranges, b.point_list, \
                     b.inst, cv.chunk_base, cv.last_chunk, cv.T_end, cv.last_pos, cv.q, im.final_T, dc, df, acc8, dcol, \
                     dfeat)
  if (r.bwd_reduce == 0) {
    if (r.fast_exp) MGS_CBWD(true, 0); else MGS_CBWD(false, 0);
  } else {
    if (r.fast_exp) MGS_CBWD(true, 1); else MGS_CBWD(false, 1);
  }
#undef MGS_CBWD
  return hipGetLastError();
}

hipError_t launch_render_fwd_chunked(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                     float* out_color, float* out_feat, hipStream_t s) {
  const int F = r.include_feature ? r.F : 0;
  switch (F) {
#define X(N) case N: return fwd_F<N>(r, b, im, cv, out_color, out_feat, s);
    MGS_FOR_EACH_F(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_render_bwd_chunked(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                     const float* dL_dcolor_px, const float* dL_dfeat_px, float* acc8,
                                     float* dL_dcolors, float* dL_dfeat, hipStream_t s) {
  const int F = r.include_feature ? r.F : 0;
  switch (F) {
#define X(N) case N: return bwd_F<N>(r, b, im, cv, dL_dcolor_px, dL_dfeat_px, acc8, dL_dcolors, dL_dfeat, s);
    MGS_FOR_EACH_F(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mgs
