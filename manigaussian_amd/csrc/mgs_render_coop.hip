// mgs_render_coop.hip -- cooperative chunk-parallel alpha-composite render (forward + backward), gfx950.
//
// Results: the reference's renderCUDA forward/backward (RAST/cuda_rasterizer/forward.cu:262-398,
// backward.cu:399-593) -- same per-pixel test order and stop rule -- see mgs_render.hip for the semantics.
//
// Decomposition (why it is not the reference's "one 256-thread block per tile"): at 128x128 there are 64
// tiles but 256 CUs / 1024 SIMDs, and a pixel's walk is a serial chain thousands of entries long.  Here a
// workgroup of NW waves owns one 8x8 pixel block of a tile; the tile's depth-sorted list is cut into chunks of
// CH entries and the NW waves sweep NW chunks per round:
//   phase A  every wave multiplies out (1 - alpha) over ITS chunk for its 64 pixels           -> LDS
//   prefix   T_in(chunk) = T_round * prod of the earlier waves' products, always in chunk order, so every
//            consumer (forward, backward) sees bit-identical transmittances
//   phase B  every wave blends its chunk from T_in with the exact reference test order; pixels whose T_in is
//            already < 1e-4 are finished (the stop rule is monotone in T).  The chunk's partial colour sums,
//            T_end and last blended position are kept for the backward.
//   the round loop ends as soon as every pixel of the block has terminated (typically after 1-2 rounds), so
//   the entries behind the termination depth are never touched.
// Backward: the same workgroup shape; per chunk the boundary state of the reference's back-to-front
// recurrences is rebuilt from what the forward kept: T = T_end, and the scalar accum_rec . dL_dpixel equals
// (sum over later visited chunks of dL_dpixel . partial) / T_end.  Chunks are then independent.
#include "mgs_render_common.h"

namespace mgs {

// ------------------------------------------- forward ------------------------------------------------
template <int F, bool FAST, bool EXACT, int NW>
__global__ void __launch_bounds__(NW * 64) coop_fwd_kernel(RenderArgs r, int CH, const uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ point_list,
                                                            const float4* __restrict__ inst, float* __restrict__ T_end,
                                                            uint32_t* __restrict__ last_pos, float* __restrict__ partial,
                                                            float* __restrict__ final_T, uint32_t* __restrict__ last_chunk,
                                                            float* __restrict__ out_color, float* __restrict__ out_feat) {
  constexpr int ROW4 = Row<F>::ROW4;
  constexpr int NCH = F + 3;
  constexpr int STAGE4 = 64 * ROW4;                         // float4 per wave
  __shared__ float4 lds[NW * STAGE4];
  __shared__ float Tp[NW][64];
  __shared__ float red_Tf[64];
  __shared__ uint32_t red_vis[64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int tile, sub;
  map_block(blockIdx.x, tile, sub);
  if (tile >= r.tiles_x * r.tiles_y) return;
  const PixBlk p = pix_blk(r, tile, sub, lane);
  const uint2 rng = ranges[tile];
  const uint32_t len = rng.y - rng.x;
  const uint32_t nch = (len + (uint32_t)CH - 1u) / (uint32_t)CH;
  const bool use_feat = (F > 0) && r.include_feature;
  float4* stage = lds + w * STAGE4;

  float Tround = 1.0f;
  uint32_t my_vis = 0;
  float my_Tf = 1.0f;

  for (uint32_t r0 = 0; r0 < nch; r0 += NW) {
    const uint32_t c = r0 + (uint32_t)w;
    const bool has = c < nch;
    const uint32_t e0 = rng.x + c * (uint32_t)CH;
    const uint32_t e1 = min(e0 + (uint32_t)CH, rng.y);
    // ---- phase A: transmittance product of this chunk ----
    float tp = 1.0f;
    if (has && !(r.dbg & 2)) {
      for (uint32_t k0 = e0; k0 < e1; k0 += 64) {
        const uint32_t e = k0 + lane;
        const bool valid = e < e1;
        float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
        if (valid) { g0 = inst[2 * (size_t)e]; g1 = inst[2 * (size_t)e + 1]; }
        unsigned long long mask = ballot(valid && cull_ok<EXACT>(g0, g1, p));
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
          tp = skip ? tp : tp * (1.0f - alpha);
        }
      }
    }
    Tp[w][lane] = tp;
    __syncthreads();
    // ---- prefix in chunk order (identical arithmetic in every wave) ----
    float T = Tround;
    for (int w2 = 0; w2 < w; w2++) T *= Tp[w2][lane];
    float Tnext = T;
    for (int w2 = w; w2 < NW; w2++) Tnext *= Tp[w2][lane];
    // ---- phase B: blend this chunk ----
    const bool live = has && p.inside && !(T < 0.0001f);
    if (ballot(live) != 0) {
      bool done = !live;
      float C[NCH];
#pragma unroll
      for (int i = 0; i < NCH; i++) C[i] = 0.f;
      uint32_t last = 0;
      for (uint32_t k0 = e0; k0 < e1; k0 += 64) {
        if (ballot(!done) == 0) break;
        const uint32_t e = k0 + lane;
        const bool valid = e < e1;
        float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
        if (valid) { g0 = inst[2 * (size_t)e]; g1 = inst[2 * (size_t)e + 1]; }
        const bool surv = valid && cull_ok<EXACT>(g0, g1, p);
        unsigned long long mask = ballot(surv);
        if (mask == 0) continue;
        wave_lds_sync();
        if (surv && !(r.dbg & 8)) stage_row<F>(stage, lane, point_list[e], point_list[e], r.colors, use_feat ? r.feats : nullptr);
        wave_lds_sync();
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
          if (ballot(blend) == 0 || (r.dbg & 1)) continue;
          const float wgt = blend ? alpha * T : 0.f;
          const float4* row = stage + j * ROW4;
          if constexpr (F > 0) {
            if (use_feat) {
              if constexpr (F % 4 == 0) {
#pragma unroll
                for (int i = 0; i < F / 4; i++) {
                  const float4 v = row[i];
                  C[3 + 4 * i] += v.x * wgt; C[3 + 4 * i + 1] += v.y * wgt;
                  C[3 + 4 * i + 2] += v.z * wgt; C[3 + 4 * i + 3] += v.w * wgt;
                }
              } else {
                const float* rf = reinterpret_cast<const float*>(row);
#pragma unroll
                for (int i = 0; i < F; i++) C[3 + i] += rf[i] * wgt;
              }
            }
          }
          {
            const float* rf = reinterpret_cast<const float*>(row);
            C[0] += rf[F] * wgt; C[1] += rf[F + 1] * wgt; C[2] += rf[F + 2] * wgt;
          }
          T = blend ? test_T : T;
          last = blend ? (k0 - e0) + (uint32_t)j + 1u : last;
        }
      }
      const size_t slot = chunk_slot(rng.x, tile, CH, c, sub);
      T_end[slot * 64 + lane] = T;
      last_pos[slot * 64 + lane] = last;
      float* pp = partial + slot * NCH * 64 + lane;
#pragma unroll
      for (int i = 0; i < NCH; i++) pp[i * 64] = C[i];
      if (live) { my_vis = c + 1; my_Tf = T; }
    }
    __syncthreads();  // Tp is rewritten next round
    Tround = Tnext;
    if (ballot(p.inside && !(Tround < 0.0001f)) == 0) break;
  }

  // ---- image = sum of the visited chunks' partials, in chunk order (deterministic); channels split over waves ----
  if (w == 0) { red_vis[lane] = 0; red_Tf[lane] = 1.0f; }
  __syncthreads();
  if (my_vis > 0) atomicMax(&red_vis[lane], my_vis);
  __syncthreads();
  const uint32_t vis = red_vis[lane];
  if (my_vis > 0 && my_vis == vis) red_Tf[lane] = my_Tf;  // exactly one wave owns the last visited chunk
  __syncthreads();  // also makes this workgroup's partial[] stores visible to all its waves
  const float Tf = red_Tf[lane];
  const uint32_t vismax = wave_umax(vis);
  const size_t HW = (size_t)r.H * r.W;
  const size_t pix = p.pixa;
  for (int ch = w; ch < NCH; ch += NW) {
    if (ch >= 3 && !use_feat) break;
    float sum = 0.f;
    for (uint32_t c = 0; c < ((r.dbg & 4) ? 0u : vismax); c++) {
      const float v = partial[chunk_slot(rng.x, tile, CH, c, sub) * NCH * 64 + (size_t)ch * 64 + lane];
      sum += (c < vis) ? v : 0.f;
    }
    if (p.inside) {
      if (ch < 3) out_color[ch * HW + pix] = sum + Tf * r.bg[ch];
      else out_feat[(ch - 3) * HW + pix] = sum;
    }
  }
  if (w == 0) {
    last_chunk[((size_t)tile * 4 + sub) * 64 + lane] = vis;
    if (p.inside) final_T[pix] = Tf;
  }
}

// ------------------------------------- forward, CH == 64 ---------------------------------------------
// Same results as coop_fwd_kernel bit for bit (same per-pixel arithmetic, same chunk-order image sum); what
// changes is WHEN things are fetched and where the image sum runs (measured on MI355X, C3: the version above
// spends 14 us re-reading its own partial sums from global memory in a dependent loop and ~3 global round
// trips per round on the critical path):
//   * one 64-entry batch per chunk: the packed records, the cull mask and the Gaussian id are loaded once per
//     round and kept in registers for both phases;
//   * the colour/feature rows of the surviving entries are gathered into the wave's LDS stage BEFORE phase A,
//     so the gather's latency hides behind the transmittance products;
//   * after phase B every wave parks its chunk's partial colours in its own (now free) LDS stage and the
//     image is accumulated across waves in chunk order from LDS, channels split over the waves; the global
//     partial[] stores stay (the backward needs them) but nothing reads them back here.
template <int F, bool FAST, bool EXACT, int NW>
__global__ void __launch_bounds__(NW * 64) coop_fwd64_kernel(RenderArgs r, const uint2* __restrict__ ranges,
                                                              const uint32_t* __restrict__ point_list,
                                                              const float4* __restrict__ inst, float* __restrict__ T_end,
                                                              uint32_t* __restrict__ last_pos, float* __restrict__ partial,
                                                              float* __restrict__ final_T, uint32_t* __restrict__ last_chunk,
                                                              float* __restrict__ out_color, float* __restrict__ out_feat) {
  constexpr int CH = 64;
  constexpr int ROW4 = Row<F>::ROW4;
  constexpr int NCH = F + 3;
  constexpr int STAGE4 = 64 * ROW4;  // float4 per wave
  constexpr int NOWN = (NCH + NW - 1) / NW;  // image channels owned by one wave
  __shared__ float4 lds[NW * STAGE4];
  __shared__ float Tp[NW][64];
  __shared__ float red_Tf[64];
  __shared__ uint32_t red_vis[64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int tile, sub;
  map_block(blockIdx.x, tile, sub);
  if (tile >= r.tiles_x * r.tiles_y) return;
  const PixBlk p = pix_blk(r, tile, sub, lane);
  const uint2 rng = ranges[tile];
  const uint32_t len = rng.y - rng.x;
  const uint32_t nch = (len + (uint32_t)CH - 1u) / (uint32_t)CH;
  const bool use_feat = (F > 0) && r.include_feature;
  float4* stage = lds + w * STAGE4;
  float* stage_f = reinterpret_cast<float*>(stage);

  float Tround = 1.0f;
  uint32_t my_vis = 0;
  float my_Tf = 1.0f;
  float img[NOWN];
#pragma unroll
  for (int k = 0; k < NOWN; k++) img[k] = 0.f;

  for (uint32_t r0 = 0; r0 < nch; r0 += NW) {
    const uint32_t c = r0 + (uint32_t)w;
    const bool has = c < nch;
    const uint32_t e = rng.x + c * (uint32_t)CH + (uint32_t)lane;
    const bool valid = has && e < rng.y;
    float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
    uint32_t id = 0;
    if (valid) { g0 = inst[2 * (size_t)e]; g1 = inst[2 * (size_t)e + 1]; id = point_list[e]; }
    const bool surv = valid && cull_ok<EXACT>(g0, g1, p);
    const unsigned long long smask = ballot(surv);
    // rows of the survivors -> LDS (the previous round's readers of this stage passed the round's last barrier)
    if (surv && !(r.dbg & 8)) {
      const uint32_t gid = gauss_of(r, id);
      stage_row<F>(stage, lane, gid, r.colors_per_view ? id : gid, r.colors, use_feat ? r.feats : nullptr);
    }
    // ---- phase A: transmittance product of this chunk ----
    float tp = 1.0f;
    {
      unsigned long long mask = smask;
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
        tp = skip ? tp : tp * (1.0f - alpha);
      }
    }
    Tp[w][lane] = tp;
    __syncthreads();
    // ---- prefix in chunk order (identical arithmetic in every wave) ----
    float T = Tround, Tnext = Tround;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) {  // all NW reads in flight; the multiplication order is the chunk order
      const float t2 = Tp[w2][lane];
      T = (w2 < w) ? T * t2 : T;
      Tnext *= t2;
    }
    // ---- phase B: blend this chunk ----
    const bool live = has && p.inside && !(T < 0.0001f);
    const bool enter = ballot(live) != 0;
    float C[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) C[i] = 0.f;
    if (enter) {
      bool done = !live;
      uint32_t last = 0;
      wave_lds_sync();  // this wave's stage rows are written
      unsigned long long mask = smask;
      while (mask) {
        if (ballot(!done) == 0) break;
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
        if (ballot(blend) == 0 || (r.dbg & 1)) continue;
        const float wgt = blend ? alpha * T : 0.f;
        const float4* row = stage + j * ROW4;
        if constexpr (F > 0) {
          if (use_feat) {
            if constexpr (F % 4 == 0) {
#pragma unroll
              for (int i = 0; i < F / 4; i++) {
                const float4 v = row[i];
                C[3 + 4 * i] += v.x * wgt; C[3 + 4 * i + 1] += v.y * wgt;
                C[3 + 4 * i + 2] += v.z * wgt; C[3 + 4 * i + 3] += v.w * wgt;
              }
            } else {
              const float* rf = reinterpret_cast<const float*>(row);
#pragma unroll
              for (int i = 0; i < F; i++) C[3 + i] += rf[i] * wgt;
            }
          }
        }
        {
          const float* rf = reinterpret_cast<const float*>(row);
          C[0] += rf[F] * wgt; C[1] += rf[F + 1] * wgt; C[2] += rf[F + 2] * wgt;
        }
        T = blend ? test_T : T;
        last = blend ? (uint32_t)j + 1u : last;
      }
      const size_t slot = chunk_slot(rng.x, tile, CH, c, sub);
      T_end[slot * 64 + lane] = T;
      last_pos[slot * 64 + lane] = last;
      float* pp = partial + slot * NCH * 64 + lane;
      if (!(r.dbg & 16)) {
#pragma unroll
        for (int i = 0; i < NCH; i++) pp[i * 64] = C[i];
      }
      if (live) { my_vis = c + 1; my_Tf = T; }
    }
    wave_lds_sync();  // every lane is done reading the stage rows
#pragma unroll
    for (int i = 0; i < NCH; i++) stage_f[i * 64 + lane] = C[i];  // zeros from a wave that had nothing to blend
    __syncthreads();
    // ---- image += this round's partial colours, in chunk order; wave w owns channels w, w + NW, ... ----
#pragma unroll
    for (int k = 0; k < NOWN; k++) {
      const int ch = w + k * NW;
      if (ch < NCH && (ch < 3 || use_feat) && !(r.dbg & 64)) {
        float acc = img[k];
#pragma unroll
        for (int w2 = 0; w2 < NW; w2++) acc += reinterpret_cast<const float*>(lds + w2 * STAGE4)[ch * 64 + lane];
        img[k] = acc;
      }
    }
    __syncthreads();  // Tp, entered and the stages are rewritten next round
    Tround = Tnext;
    if (ballot(p.inside && !(Tround < 0.0001f)) == 0) break;
  }

  if (w == 0) { red_vis[lane] = 0; red_Tf[lane] = 1.0f; }
  __syncthreads();
  if (my_vis > 0) atomicMax(&red_vis[lane], my_vis);
  __syncthreads();
  const uint32_t vis = red_vis[lane];
  if (my_vis > 0 && my_vis == vis) red_Tf[lane] = my_Tf;  // exactly one wave owns the last visited chunk
  __syncthreads();
  const float Tf = red_Tf[lane];
  const size_t HW = (size_t)r.Hv * r.W;  // one image plane of one view
  if (p.inside) {
#pragma unroll
    for (int k = 0; k < NOWN; k++) {
      const int ch = w + k * NW;
      if (ch < 3) out_color[((size_t)p.v * 3 + ch) * HW + p.pixl] = img[k] + Tf * r.bg[ch];
      else if (ch < NCH && use_feat) out_feat[((size_t)p.v * F + (ch - 3)) * HW + p.pixl] = img[k];
    }
  }
  if (w == 0) {
    last_chunk[((size_t)tile * 4 + sub) * 64 + lane] = vis;
    if (p.inside) final_T[p.pixa] = Tf;
  }
}

// ------------------------------------------- backward -----------------------------------------------
template <int F, bool FAST, bool EXACT, int RED, int NW>
__global__ void __launch_bounds__(NW * 64) coop_bwd_kernel(RenderArgs r, int CH, const uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ point_list,
                                                            const float4* __restrict__ inst,
                                                            const uint32_t* __restrict__ last_chunk,
                                                            const float* __restrict__ T_end,
                                                            const uint32_t* __restrict__ last_pos,
                                                            const float* __restrict__ partial, float* __restrict__ q,
                                                            const float* __restrict__ final_T,
                                                            const float* __restrict__ dL_dpix,
                                                            const float* __restrict__ dL_dpix_F, float* __restrict__ acc8,
                                                            float* __restrict__ dL_dcolors, float* __restrict__ dL_dfeat) {
  constexpr int ROW4 = Row<F>::ROW4;
  constexpr int NCH = F + 3;
  constexpr int FP = F > 0 ? next_pow2(F) : 1;
  __shared__ float4 lds[NW * 64 * ROW4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int tile, sub;
  map_block(blockIdx.x, tile, sub);
  if (tile >= r.tiles_x * r.tiles_y) return;
  const uint32_t lc = last_chunk[((size_t)tile * 4 + sub) * 64 + lane];
  const uint32_t lcmax = wave_umax(lc);
  if (lcmax == 0) return;
  const PixBlk p = pix_blk(r, tile, sub, lane);
  const uint2 rng = ranges[tile];
  const bool use_feat = (F > 0) && r.include_feature;
  const size_t HW = (size_t)r.H * r.W;
  const size_t pix = p.pixa;
  float4* stage = lds + w * 64 * ROW4;
  const bool any = lc > 0;  // this pixel visited at least one chunk (=> inside)

  const float T_final = any ? final_T[pix] : 0.f;
  float dLc[3] = {0.f, 0.f, 0.f};
  float dLf[F > 0 ? F : 1];
#pragma unroll
  for (int i = 0; i < (F > 0 ? F : 1); i++) dLf[i] = 0.f;
  if (any) {
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

  // ---- phase 0: q[c] = dL . partial[c] for the visited chunks ----
  for (uint32_t c = (uint32_t)w; c < lcmax; c += NW) {
    const size_t slot = chunk_slot(rng.x, tile, CH, c, sub);
    float s = 0.f;
    if (c < lc) {
      const float* pp = partial + slot * NCH * 64 + lane;
#pragma unroll
      for (int ch = 0; ch < 3; ch++) s += dLc[ch] * pp[ch * 64];
      if constexpr (F > 0) {
        if (use_feat) {
#pragma unroll
          for (int ch = 0; ch < F; ch++) s += dLf[ch] * pp[(3 + ch) * 64];
        }
      }
    }
    q[slot * 64 + lane] = s;
  }
  __syncthreads();  // q (global, this workgroup only) is visible to the other waves

  const float ddelx_dx = 0.5f * r.W, ddely_dy = 0.5f * r.H;
  // ---- phase 1: independent chunks ----
  for (uint32_t c = (uint32_t)w; c < lcmax; c += NW) {
    const size_t slot = chunk_slot(rng.x, tile, CH, c, sub);
    const uint32_t last = (c < lc) ? last_pos[slot * 64 + lane] : 0u;
    const uint32_t kmax = wave_umax(last);
    if (kmax == 0) continue;
    const bool live = last > 0;
    float B = 0.f;
    for (uint32_t c2 = c + 1; c2 < lcmax; c2++) {
      const float v = q[chunk_slot(rng.x, tile, CH, c2, sub) * 64 + lane];
      B += (live && c2 < lc) ? v : 0.f;
    }
    float T = live ? T_end[slot * 64 + lane] : 1.0f;
    float A = live ? B / T : 0.f, last_alpha = 0.f, last_D = 0.f;
    const uint32_t e0 = rng.x + c * (uint32_t)CH;
    const int nb = (int)((kmax + 63u) / 64u);
    for (int bi = nb - 1; bi >= 0; --bi) {
      const uint32_t e = e0 + (uint32_t)bi * 64u + lane;
      const uint32_t pos_l = (uint32_t)bi * 64u + lane + 1u;
      const bool valid = e < rng.y && pos_l <= kmax;
      float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
      if (valid) { g0 = inst[2 * (size_t)e]; g1 = inst[2 * (size_t)e + 1]; }
      const bool surv = valid && cull_ok<EXACT>(g0, g1, p);
      unsigned long long mask = ballot(surv);
      if (mask == 0) continue;
      uint32_t id_l = 0;
      wave_lds_sync();
      if (surv) {
        id_l = point_list[e];
        stage_row<F>(stage, lane, id_l, id_l, r.colors, use_feat ? r.feats : nullptr);
      }
      wave_lds_sync();
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
}

// ------------------------------------------- dispatch ------------------------------------------------
template <int F>
constexpr int waves_for() { return F <= 32 ? 16 : 8; }  // 64 staged rows per wave must fit the 160 KB LDS

template <int F>
static hipError_t fwd_F(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv, float* oc,
                        float* of, hipStream_t s) {
  constexpr int NW = waves_for<F>();
  const int T = r.tiles_x * r.tiles_y;
  const int grid = ((T + 7) / 8) * 32;
#define MGS_CF(FAST, EXACT)                                                                                         \
  hipLaunchKernelGGL((coop_fwd_kernel<F, FAST, EXACT, NW>), dim3(grid), dim3(NW * 64), 0, s, r, cv.CH, im.ranges,    \
                     b.point_list, b.inst, cv.T_end, cv.last_pos, cv.partial, im.final_T, cv.last_chunk, oc, of)
#define MGS_CF64(FAST, EXACT)                                                                                       \
  hipLaunchKernelGGL((coop_fwd64_kernel<F, FAST, EXACT, NW>), dim3(grid), dim3(NW * 64), 0, s, r, im.ranges,          \
                     b.point_list, b.inst, cv.T_end, cv.last_pos, cv.partial, im.final_T, cv.last_chunk, oc, of)
  if (cv.CH == 64 && options().fwd_mode == 1) {
    if (r.fast_exp) { if (r.exact_cull) MGS_CF64(true, true); else MGS_CF64(true, false); }
    else            { if (r.exact_cull) MGS_CF64(false, true); else MGS_CF64(false, false); }
  } else if (r.fast_exp) { if (r.exact_cull) MGS_CF(true, true); else MGS_CF(true, false); }
  else            { if (r.exact_cull) MGS_CF(false, true); else MGS_CF(false, false); }
#undef MGS_CF64
#undef MGS_CF
  return hipGetLastError();
}

template <int F>
static hipError_t bwd_F(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv, const float* dc,
                        const float* df, float* acc8, float* dcol, float* dfeat, hipStream_t s) {
  constexpr int NW = waves_for<F>();
  const int T = r.tiles_x * r.tiles_y;
  const int grid = ((T + 7) / 8) * 32;
#define MGS_CB(FAST, EXACT, RED)                                                                                     \
  hipLaunchKernelGGL((coop_bwd_kernel<F, FAST, EXACT, RED, NW>), dim3(grid), dim3(NW * 64), 0, s, r, cv.CH, im.ranges, \
                     b.point_list, b.inst, cv.last_chunk, cv.T_end, cv.last_pos, cv.partial, cv.q, im.final_T, dc, df, \
                     acc8, dcol, dfeat)
  if (r.bwd_reduce == 0) {
    if (r.fast_exp) MGS_CB(true, true, 0); else MGS_CB(false, true, 0);
  } else if (r.fast_exp) {
    if (r.exact_cull) MGS_CB(true, true, 1); else MGS_CB(true, false, 1);
  } else {
    if (r.exact_cull) MGS_CB(false, true, 1); else MGS_CB(false, false, 1);
  }
#undef MGS_CB
  return hipGetLastError();
}

hipError_t launch_render_fwd_coop(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                  float* out_color, float* out_feat, hipStream_t s) {
  const int F = r.include_feature ? r.F : 0;
  switch (F) {
#define X(N) case N: return fwd_F<N>(r, b, im, cv, out_color, out_feat, s);
    MGS_FOR_EACH_F(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_render_bwd_coop(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                  const float* dL_dcolor_px, const float* dL_dfeat_px, float* acc8, float* dL_dcolors,
                                  float* dL_dfeat, hipStream_t s) {
  const int F = r.include_feature ? r.F : 0;
  switch (F) {
#define X(N) case N: return bwd_F<N>(r, b, im, cv, dL_dcolor_px, dL_dfeat_px, acc8, dL_dcolors, dL_dfeat, s);
    MGS_FOR_EACH_F(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mgs
