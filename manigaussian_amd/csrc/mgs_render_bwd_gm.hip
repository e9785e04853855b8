// mgs_render_bwd_gm.hip -- Gaussian-major render backward (K8) for the chunk-parallel forward (mgs_render_dense.hip), gfx950.
//
// Results: the reference's renderCUDA backward (RAST/cuda_rasterizer/backward.cu:399-593): for every blended
// (pixel, Gaussian) pair  dL/dalpha = (D - accum_rec.dL) * T_before - T_final/(1-alpha) * (bg.dL_rgb)  with
// D = colour/feature row . dL_dpixel, then the chain to mean2D (NDC units), conic, opacity, colour and feature
// rows, summed over the pixels.  Which pairs are blended comes from the forward (last_pos, T_end per chunk).
//
// Decomposition.  A pixel-major backward (lane = pixel) has to reduce 9+F values per Gaussian over the 64 pixel lanes:
// measured VALU-bound (8000 VALU instr. per wave, half of them reduction).  Here the roles flip inside a chunk:
//   lane (n, h) = Gaussian n (0..31) of a group of <= 32 block-reaching entries of the chunk, h = pixel half;
//   the lane walks its 32 pixels p(u, r, h) = 32u + (r&3) + 8(r>>2) + 4h serially, so every per-Gaussian sum
//   (mean2D, conic, opacity, colour) is a private register accumulation -- no cross-lane reduction at all;
//   per pixel, transmittance is an exclusive prefix PRODUCT of (1-alpha) over the group's lanes and the
//   "colour behind" term an exclusive suffix SUM of D*alpha*T  (DPP row_shr / row_bcast15 scans over each
//   32-lane half):  accum_rec.dL = S/T_after  =>  dL/dalpha = D*T - (S + T_final*bg.dL)/(1-alpha);
//   the two dense contractions over channels run on the matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32,
//   bit-for-bit an fmaf chain): D[pixel][Gaussian] = dL[pixel][ch] . row[Gaussian][ch]  (K = channels), and
//   dL_dfeature[ch][Gaussian] += dL[pixel][ch] * (alpha*T)[pixel][Gaussian]  (K = pixels).  The pixel order
//   p(u, r, h) IS the MFMA C-layout (col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5)), so D lands in the lane
//   that needs it and alpha*T is consumed from the lane that made it: no LDS transposes.
// One atomic per value per group leaves each lane (the pixel lanes of the old kernel needed 64x the adds
// before their butterfly).  Chunks, workgroup shape and the forward's saved state are unchanged.
#include "mgs_render_common.h"

namespace mgs {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Phase timeline (diagnostic, MgsOptions.dbg = 256): s_memtime stamps per (workgroup, wave, event), first chunk of a wave only
constexpr int BTRACE_EVENTS = 16;
__device__ unsigned long long g_btrace[512 * 16 * BTRACE_EVENTS];
#define MGS_BTRACE(ev)                                                                                     \
  do {                                                                                                     \
    if ((r.dbg & 256) && lane == 0 && blockIdx.x < 512 && w < 16)                                           \
      g_btrace[((size_t)blockIdx.x * 16 + w) * BTRACE_EVENTS + (ev)] = __builtin_amdgcn_s_memtime();       \
  } while (0)

template <int F>
struct GmCfg {
  static constexpr int NCH = F + 3;                 // feature channels, then r, g, b
  static constexpr int KCH = (NCH + 1) & ~1;        // channels padded to the MFMA's K = 2
  static constexpr int NCT = (F + 31) / 32;         // 32-row tiles of the dL_dfeature contraction
  static constexpr int SROW = (NCT > 0 ? NCT * 32 : 0) + 1;  // dLs row stride (odd: conflict-free column reads)
};

template <bool FAST>
__device__ __forceinline__ float gm_exp(float x) { return exp_<FAST>(x); }

// Per-wave LDS record of one entry: {instance id, 1-based position in the chunk}
struct GmRec { float4 g0, g1; uint32_t id, pos; };

// A chunk is 64 consecutive SURVIVORS of this block's compacted list `surv` (written by the forward: no culling here,
// groups are full) and T_mid is the transmittance entering the second group.  Chunk c's record is
// round_base[round c / NWF] + c % NWF with NWF = the forward's waves (RenderArgs::nwf).
// NWF: waves of the forward that wrote the state (chunk records per round).  TWO: two workgroups share a CU.
// PAIR (round 5, the default form at one workgroup per CU): a lane walks its pixels TWO AT A TIME (steps 2j, 2j + 1: neighbours
// in a pixel row) -- two independent dependency chains per lane, the arithmetic and the exact exp's multiply-adds in packed fp32
// (v_pk_*), the four DPP scans of a double step as two interleaved pairs (each chain's wait states are the other's issue slots):
// 148 vector issue slots per double step against 2 x 87, 14 s_nop against 30.  It needs 167 registers, hence 12 waves per
// workgroup (3 per SIMD: 168 registers each) instead of 16 x 128; a block's chunks 12 .. 15 (rare) go to waves 0 .. 3 in a second
// pass.  configs[2]: 55.3 -> 50.4 us; gradients equal the one-pixel form's to 1e-6 of the tensor max (another summation order).
template <int F, bool FAST, int NW, int NWF_, bool TWO, bool PAIR = false>
__global__ void __launch_bounds__(NW * 64, TWO ? 4 : 1) gm_bwd_kernel(RenderArgs r, const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ round_base,
                                                          const uint32_t* __restrict__ last_chunk,
                                                          const float* __restrict__ T_end,
                                                          const uint32_t* __restrict__ last_pos,
                                                          const float* __restrict__ partial, float* __restrict__ q,
                                                          const float* __restrict__ final_T,
                                                          const float* __restrict__ dL_dpix,
                                                          const float* __restrict__ dL_dpix_F, float* __restrict__ acc8,
                                                          float* __restrict__ dL_dcolors, float* __restrict__ dL_dfeat,
                                                          const float* __restrict__ T_mid,
                                                          const uint32_t* __restrict__ surv, size_t surv_stride,
                                                          const uint2* __restrict__ nsurv) {
  using C = GmCfg<F>;
  constexpr int NCH = C::NCH, KCH = C::KCH, NCT = C::NCT;
  constexpr int CH = CHUNK;
  constexpr uint32_t NWF = NWF_;
  constexpr uint32_t RBH = 8;             // rounds whose first record is staged in LDS
  __shared__ uint32_t rb_hist[RBH];
  __shared__ float dLT[KCH][65];          // [channel][pixel] (odd stride): A operand of the D contraction, read by rows, and
                                          // of the feature contraction, read by columns (lane = channel)
  __shared__ float4 pd[NW][64];           // per wave, per pixel: {T_in, S_after + T_final*bg.dL, last (bits), T_in of group 1}
  __shared__ float4 rec0[NW][64], rec1[NW][64];  // per wave: compacted entries of the current chunk (packed records)
  __shared__ uint32_t recid[NW][64];             // ... their instance ids (entry e of the chunk sits at index e)
  constexpr int TROW = NCT * 32 + 9;      // per wave: [32 Gaussians][feature sums | 9 scalar sums], odd stride
  __shared__ float trbuf[NW][32 * TROW];
  __shared__ uint2 gid[NW][32];           // per wave: {instance id, Gaussian} of the group's lanes (0xffffffff: empty lane)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: scalar chunk loops and addresses
  // Chunks of a block by wave: wave w walks chunks cw, cw + NW, ...  Waves w, w + 4, w + 8 share a SIMD, and a block's FRONT
  // chunks are its long ones (every pixel alive): chunks 4 .. 7 go to waves 7 .. 4, so that the SIMD that holds chunk 0 holds
  // chunk 7 (not 4) beside it, the one with chunk 3 holds chunk 4 (experiment, MgsOptions.dbg & 2048: the identity)
  int cw = ((r.dbg & 2048) == 0 && (w & 4) && NW >= 8) ? (w ^ 3) : w;
  int tile, sub;
  map_block(blockIdx.x, tile, sub);
  if (tile >= r.tiles_x * r.tiles_y) return;
  MGS_BTRACE(0);
  // The kernel is a chain of dependent memory round trips before the pixel loops start (measured: 36 k of a block's 132 k
  // cycles, scripts/trace_bwd.py): everything that depends on the pixel alone is requested at once, up front.
  const PixBlk p = pix_blk(r, tile, sub, lane);
  const bool use_feat = (F > 0) && r.include_feature;
  const size_t HW = (size_t)r.Hv * r.W;  // one image plane of one view
  const size_t pix = p.pixl;
  const uint32_t lc = last_chunk[((size_t)tile * 4 + sub) * 64 + lane];
  const uint2 rng = ranges[tile];
  const uint2 nsv = nsurv[(size_t)tile * 4 + sub];  // {survivors the forward listed for this block, round 0's first record}
  const uint32_t nsb = nsv.x;
  float dLc[3] = {0.f, 0.f, 0.f};
  float dLf[F > 0 ? F : 1];
#pragma unroll
  for (int i = 0; i < (F > 0 ? F : 1); i++) dLf[i] = 0.f;
  float T_final = 0.f;
  if (p.inside) {
    T_final = final_T[p.pixa];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) dLc[ch] = dL_dpix[((size_t)p.v * 3 + ch) * HW + pix];
    if constexpr (F > 0) {
      if (use_feat) {
#pragma unroll
        for (int ch = 0; ch < F; ch++) dLf[ch] = dL_dpix_F[((size_t)p.v * F + ch) * HW + pix];
      }
    }
  }
  const uint32_t lcmax = wave_umax(lc);
  if (lcmax == 0) return;
  const bool any = lc > 0;  // this pixel visited at least one chunk (=> inside)
  // Round 6: a block of 9 or 10 chunks -- the blocks this kernel's duration is made of (profiles/r06_trace_bwd.log: the THIRD wave
  // of a SIMD is served last and finishes ~20 k cycles after the other two) -- cuts its extra chunks 8, 9 in two: waves 8 + 9 take
  // the pixel tiles u = 0 and u = 1 of chunk 8, waves 10 + 11 those of chunk 9.  A wave is self-contained (its own staging
  // buffers, sums handed over by atomics), so a half needs nothing but the tile range; only these chunks' atomics double.
  int u_lo = 0, u_hi = 2;
  if constexpr (PAIR && NW == 12) {
    const uint32_t lcu = (uint32_t)__builtin_amdgcn_readfirstlane((int)lcmax);  // (wave_umax's result lives in a VGPR: say it is uniform)
    if (!(r.dbg & 4096) && lcu > 8u && lcu <= 10u && w >= 8) {  // (workgroup- and wave-uniform)
      cw = 8 + ((w - 8) >> 1);
      u_lo = (w - 8) & 1;
      u_hi = u_lo + 1;
    }
  }
  const unsigned long long umask = (u_hi - u_lo == 2) ? ~0ull : (0xffffffffull << (32 * u_lo));  // my tiles' pixels
  // (Measured, profiles/r03_exp_bwd_units.log: cutting a block's chunks into (chunk, pixel tile) units so that the ~9 idle
  //  waves of a block at BASELINE configs[2] take half of a chunk each does NOT pay: without the atomics the kernel goes from
  //  45 to 43 us only -- the pixel steps are issue-bound per SIMD, not latency-bound per wave -- and with twice the atomics
  //  (each unit hands over its own sums) it takes 73 us: the L2's float atomics are the scarce unit, see the epilogue.)
  const uint32_t* __restrict__ my_rounds = round_base + round_entry(rng.x, tile, sub, 0);
  // records of round 0 start at nsv.y (one load, no table, no barrier); later rounds (rare) go through the table in LDS
  if (tid >= 1 && tid < (int)RBH && (uint32_t)tid * NWF < lcmax) rb_hist[tid] = my_rounds[4 * (size_t)tid];
  if (tid == 0) rb_hist[0] = nsv.y;
  // (c is wave-uniform.  Three sources behind three BRANCHES: written as one select, the compiler merges the LDS and the
  //  memory read into a flat load of a selected address with a full s_waitcnt behind it -- every loop over chunks below
  //  then runs one serialised round trip per chunk.  The atomic load keeps the rare memory read a separate instruction.)
  auto slot_of = [&](uint32_t c) -> size_t {
    const uint32_t rr = c / NWF;
    uint32_t b = nsv.y;
    if (rr != 0u) {
      b = rb_hist[rr < RBH ? rr : 1u];
      if (__builtin_expect(rr >= RBH, 0))
        b = __hip_atomic_load(&my_rounds[4 * (size_t)rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return (size_t)b + (size_t)(c % NWF);
  };
  if (lcmax > NWF) __syncthreads();  // (workgroup-uniform) only blocks with a second round read the table
  // ---- this wave's FIRST chunk (c = w): its per-pixel state and its survivors' records are requested now, so that they
  //      arrive while q is being computed (they used to be three more round trips after the barrier) ----
  const bool pf = (uint32_t)cw < lcmax;
  const uint32_t c1 = (uint32_t)cw;           // my first chunk
  uint32_t pf_last = 0u, pf_id = 0u;
  float pf_Tin = 1.0f, pf_Tmid = 1.0f;
  float4 pf_g0 = make_float4(0, 0, 0, 0), pf_g1 = make_float4(0, 0, -1.f, -1.f);
  if (pf) {
    const size_t slot0 = slot_of(c1);
    // (a pixel that did not visit the chunk -- c1 >= lc -- may belong to a half block the forward wrote nothing for:
    //  neutral values instead of whatever the record holds)
    if (c1 < lc) { pf_last = last_pos[slot0 * 64 + lane]; pf_Tmid = T_mid[slot0 * 64 + lane]; }
    if (c1 > 0) pf_Tin = T_end[slot_of(c1 - 1u) * 64 + lane];
    const uint32_t first = c1 * (uint32_t)CH;
    const uint32_t nin = nsb > first ? min((uint32_t)CH, nsb - first) : 0u;
    if ((uint32_t)lane < nin) {
      pf_id = surv[(size_t)sub * surv_stride + rng.x + first + (uint32_t)lane];
      pf_g0 = r.rec[2 * (size_t)pf_id]; pf_g1 = r.rec[2 * (size_t)pf_id + 1];
    }
  }

  // ---- pixel-lane prologue: dL of this block into LDS (both layouts), q[c] = dL . partial[c] ----
  if (!any) {
    T_final = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) dLc[ch] = 0.f;
#pragma unroll
    for (int i = 0; i < (F > 0 ? F : 1); i++) dLf[i] = 0.f;
  }
  float bgT = 0.f;
  float* qs = &trbuf[0][0];  // q of the block's first QCAP chunks, [chunk][pixel]: trbuf is idle until the first epilogue
  constexpr uint32_t QCAP = (uint32_t)(NW * 32 * TROW / 64);
  // (workgroup-uniform) every q of the block fits in LDS: each wave then forms the sums behind ALL its chunks here, one pass
  // over qs, and parks those of its later chunks in q's own records (which only it reads again).  Otherwise (lists of more
  // than QCAP * 64 survivors per block) q goes to memory and every chunk sums what lies behind it from there.
  const bool fastB = lcmax <= QCAP;
  {
    bgT = T_final * (r.bg[0] * dLc[0] + r.bg[1] * dLc[1] + r.bg[2] * dLc[2]);
    if (w == 0) {
      if constexpr (F > 0) {
#pragma unroll
        for (int ch = 0; ch < F; ch++) dLT[ch][lane] = dLf[ch];
      }
#pragma unroll
      for (int ch = 0; ch < 3; ch++) dLT[F + ch][lane] = dLc[ch];
#pragma unroll
      for (int ch = NCH; ch < KCH; ch++) dLT[ch][lane] = 0.f;
    }
    MGS_BTRACE(1);
    for (uint32_t c = (uint32_t)w; c < lcmax; c += NW) {
      const size_t slot = slot_of(c);
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
      if (!fastB) q[slot * 64 + lane] = s;
      if (c < QCAP) qs[c * 64 + (uint32_t)lane] = s;
    }
  }
  MGS_BTRACE(2);
  __syncthreads();  // dLT/dLs, qs (LDS) and q (global, this workgroup only) are visible to every wave
  MGS_BTRACE(3);
  // what lies behind this wave's first chunk: dL . (the later chunks' partial colours), in chunk order; then the chunk is
  // staged in this wave's LDS buffers right here, so that nothing prefetched stays live in registers across the loops
  if (pf) {
    const bool live0 = pf_last > 0u;
    float pf_B = 0.f;
    if (fastB) {
      // my chunks from the last one down to the first, one running sum (back to front, chunk order)
      uint32_t hi = lcmax;  // the sum holds q[hi .. lcmax)
      for (uint32_t c = (uint32_t)cw + ((lcmax - 1u - (uint32_t)cw) / (uint32_t)NW) * (uint32_t)NW;; c -= (uint32_t)NW) {
        for (uint32_t c2 = hi; c2-- > c + 1u;) {
          const float v = qs[c2 * 64 + (uint32_t)lane];
          pf_B += (c2 < lc) ? v : 0.f;
        }
        hi = c + 1u;
        if (c == (uint32_t)cw) break;
        q[slot_of(c) * 64 + lane] = pf_B;
      }
    } else {
      const uint32_t cl = min(lcmax, QCAP);
      for (uint32_t c2 = c1 + 1u; c2 < cl; c2++) {
        const float v = qs[c2 * 64 + (uint32_t)lane];
        pf_B += (c2 < lc) ? v : 0.f;
      }
      for (uint32_t c2 = max(c1 + 1u, QCAP); c2 < lcmax; c2++) {
        const float v = q[slot_of(c2) * 64 + lane];
        pf_B += (c2 < lc) ? v : 0.f;
      }
    }
    pd[w][lane] = make_float4((live0 && c1 > 0) ? pf_Tin : 1.0f, live0 ? pf_B + bgT : 0.f, __uint_as_float(pf_last), pf_Tmid);
    rec0[w][lane] = pf_g0; rec1[w][lane] = pf_g1; recid[w][lane] = pf_id;
  }
  MGS_BTRACE(10);
  __syncthreads();  // qs has been read: trbuf belongs to the epilogues again
  MGS_BTRACE(11);

  const float ddelx_dx = 0.5f * r.W, ddely_dy = 0.5f * r.Hv;
  const int n_ = lane & 31, h_ = lane >> 5;
  const float bx0 = p.bxmin, by0 = p.bymin;  // block origin (pixel coordinates are bx0 + (p&7), by0 + (p>>3))

  for (uint32_t c = (uint32_t)cw; c < lcmax; c += NW) {
    // ---- pixel-lane: state of this chunk for my pixel ----
    const bool first_it = c == (uint32_t)cw;  // staged by the prologue
    const size_t slot = slot_of(c);
    if (first_it) wave_lds_sync();
    const uint32_t last = first_it ? __float_as_uint(pd[w][lane].z) : ((c < lc) ? last_pos[slot * 64 + lane] : 0u);
    const uint32_t kmax = wave_umax(last);
    if (kmax == 0) continue;
    const bool live = last > 0;
    // pixels that blended something of group 0 / group 1 of this chunk: a pixel step whose two pixels (one per half-wave)
    // are both dead for the group contributes exact zeros to every sum and is skipped
    const unsigned long long lm0 = ballot(live), lm1 = ballot(last > 32u);
    if (first_it) MGS_BTRACE(4);
    int ns;
    {
      const uint32_t first = c * (uint32_t)CH;
      const uint32_t nin = nsb > first ? min((uint32_t)CH, nsb - first) : 0u;
      ns = (int)min(nin, kmax);
      if (ns == 0) continue;
      if (!first_it) {
        float B = 0.f;
        if (fastB) {
          B = q[slot * 64 + lane];  // parked by the prologue
        } else {
          for (uint32_t c2 = c + 1; c2 < lcmax; c2++) {
            const float v = q[slot_of(c2) * 64 + lane];
            B += (c2 < lc) ? v : 0.f;
          }
        }
        const float T_in = (live && c > 0) ? T_end[slot_of(c - 1) * 64 + lane] : 1.0f;
        // ---- entry-lane: the chunk's survivors (the forward listed them in order) ----
        float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
        uint32_t id_e = 0;
        if (lane < ns) {
          id_e = surv[(size_t)sub * surv_stride + rng.x + first + (uint32_t)lane];
          g0 = r.rec[2 * (size_t)id_e]; g1 = r.rec[2 * (size_t)id_e + 1];
        }
        wave_lds_sync();  // the previous chunk's readers of pd/recs are done (same wave)
        pd[w][lane] = make_float4(T_in, live ? B + bgT : 0.f, __uint_as_float(last), (c < lc) ? T_mid[slot * 64 + lane] : 1.0f);
        rec0[w][lane] = g0; rec1[w][lane] = g1; recid[w][lane] = id_e;
        wave_lds_sync();
      }
    }
    const int ngroups = (ns + 31) >> 5;

    // ---- Gaussian-lane: groups of <= 32 entries, last group first (suffix sums run back to front) ----
    if (first_it) MGS_BTRACE(5);
    for (int g = ngroups - 1; g >= 0; --g) {
      // (lane-derived indices pass through an opaque asm once per group: otherwise the compiler hoists the address arithmetic
      //  of the unrolled LDS accesses below out of the chunk / group loops and spills it)
      int n = n_, h = h_;
      asm volatile("" : "+v"(n), "+v"(h));
      const int gi = 32 * g + n;
      const bool has = gi < ns;
      GmRec rec;
      {
        const int ri = has ? gi : 0;
        rec.g0 = rec0[w][ri]; rec.g1 = rec1[w][ri]; rec.id = recid[w][ri]; rec.pos = (uint32_t)ri + 1u;  // 1-based position
      }
      const float ex = rec.g0.x, ey = rec.g0.y, cx = rec.g0.z, cy = rec.g0.w, cz = rec.g1.x;
      const float op = has ? rec.g1.y : 0.f;
      const uint32_t pos = has ? rec.pos : 0xffffffffu;
      const uint32_t id = rec.id;                 // instance id (virtual in a multi-view batch): acc8 / per-view colour row
      const uint32_t gidn = gauss_of(r, id);      // the Gaussian: feature row, feature gradient
      const uint32_t cid = r.colors_per_view ? id : gidn;

      float a_mx = 0.f, a_my = 0.f, a_cx = 0.f, a_cy = 0.f, a_cz = 0.f, a_op = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f;
      float touched = 0.f;  // > 0 iff my Gaussian was blended for one of my pixels (else all its sums are exact zeros)
      f32x16 Cf[NCT > 0 ? NCT : 1];
#pragma unroll
      for (int ct = 0; ct < (NCT > 0 ? NCT : 1); ct++)
#pragma unroll
        for (int i = 0; i < 16; i++) Cf[ct][i] = 0.f;

      // two passes of 16 pixels per lane (tile u = pixel rows 4u..4u+3 of the block); kept rolled so that only one
      // D tile and no colour/feature row are live during the pixel steps
#pragma unroll 1
      for (int u = u_lo; u < u_hi; u++) {
        const uint32_t lmu = (uint32_t)((g == 0 ? lm0 : lm1) >> (32 * u));  // live pixels of this tile (block rows 4u .. 4u+3)
        if (lmu == 0u) continue;
        // D = dL . row for my 16 pixels of this tile, on the matrix cores.  B operand: lane (n, h) feeds channel
        // 2t + h of its Gaussian's row (features, then r, g, b, zero padding), straight from memory.
        // The contraction index pairs the feature channels (t, FH + t), FH = F / 2, so that lane (n, h) needs the
        // CONTIGUOUS channels h FH .. h FH + FH - 1 of its Gaussian's row: float4 loads instead of one gather per
        // channel (18 gathers of 4 bytes from 32 different rows each, per tile and group, at F = 32); then (r, g), (b, 0).
        f32x16 Dt;
        {
          constexpr int FH = (F + 1) / 2;
          float bop[FH + 2];
          const float* frow = r.feats + (size_t)gidn * F + h * FH;
          if constexpr (F > 0 && FH % 4 == 0 && F % 4 == 0) {
#pragma unroll
            for (int t4 = 0; t4 < FH / 4; t4++) {
              const float4 v = (has && use_feat) ? *reinterpret_cast<const float4*>(frow + 4 * t4) : make_float4(0, 0, 0, 0);
              bop[4 * t4] = v.x; bop[4 * t4 + 1] = v.y; bop[4 * t4 + 2] = v.z; bop[4 * t4 + 3] = v.w;
            }
          } else {
#pragma unroll
            for (int t = 0; t < FH; t++) bop[t] = (has && use_feat && h * FH + t < F) ? frow[t] : 0.f;
          }
          const float* crow = r.colors + (size_t)cid * 3;
          bop[FH] = has ? crow[h] : 0.f;                       // (r, g)
          bop[FH + 1] = (has && h == 0) ? crow[2] : 0.f;       // (b, 0)
#pragma unroll
          for (int i = 0; i < 16; i++) Dt[i] = 0.f;
#pragma unroll
          for (int t = 0; t < FH; t++) {
            const int ch = h * FH + t;
            float a;  // A[i = pixel n of tile u][k = h]
            if constexpr (2 * FH == F) a = dLT[ch][32 * u + n];  // (every channel h FH + t exists: no per-lane guard)
            else a = (ch < F) ? dLT[ch < F ? ch : 0][32 * u + n] : 0.f;
            Dt = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bop[t], Dt, 0, 0, 0);  // B[k = h][j = Gaussian n]
          }
          Dt = __builtin_amdgcn_mfma_f32_32x32x2f32(dLT[F + h][32 * u + n], bop[FH], Dt, 0, 0, 0);
          Dt = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? dLT[F + 2][32 * u + n] : 0.f, bop[FH + 1], Dt, 0, 0, 0);
        }
        const float pyu = by0 + (float)(4 * u);
        if constexpr (PAIR) {
          f32x2 p_mx = {0.f, 0.f}, p_my = {0.f, 0.f}, p_cx = {0.f, 0.f}, p_cy = {0.f, 0.f}, p_cz = {0.f, 0.f}, p_op = {0.f, 0.f};
          f32x2 p_r = {0.f, 0.f}, p_g = {0.f, 0.f}, p_b = {0.f, 0.f};
#pragma unroll 1
          for (int j = 0; j < 8; j++) {  // steps 2j, 2j + 1: pixels (x, x + 1) of one row of my half tile
            const int rr = 2 * j;
            const int bit = (rr & 3) + 8 * (rr >> 2);
            if (((lmu >> bit) & 0x33u) == 0u) continue;  // all four pixels of the double step (two per half-wave) dead in this group
            const int pp = 32 * u + bit + 4 * h;         // my first pixel; the second is pp + 1
            const float4 s0 = pd[w][pp], s1 = pd[w][pp + 1];
            const f32x2 Tg = (g == 0) ? f32x2{s0.x, s1.x} : f32x2{s0.w, s1.w};
            // (both offsets as ONE subtraction from the exact pixel coordinate, like the forward and the one-pixel form:
            //  dx0 - 1.0f rounds twice and can differ by an ulp where |dx| crosses a power of two inside the pair)
            const float px0 = bx0 + (float)((rr & 3) + 4 * h);
            const f32x2 dx = f32x2{ex, ex} - f32x2{px0, px0 + 1.0f};
            const float dy = ey - (pyu + (float)(rr >> 2));
            const f32x2 power = gauss_power2(cx, cy, cz, dx, dy);
            const f32x2 G = exp2_<FAST>(power);
            const f32x2 alpha = {fminf(0.99f, op * G.x), fminf(0.99f, op * G.y)};
            const bool act0 = pos <= __float_as_uint(s0.z) && !(power.x > 0.0f) && !(alpha.x < 1.0f / 255.0f);
            const bool act1 = pos <= __float_as_uint(s1.z) && !(power.y > 0.0f) && !(alpha.y < 1.0f / 255.0f);
            const f32x2 om = {act0 ? 1.0f - alpha.x : 1.0f, act1 ? 1.0f - alpha.y : 1.0f};
            float e0 = om.x, e1 = om.y;
            half_excl_scan_mul2(e0, e1, lane);
            const f32x2 T = Tg * f32x2{e0, e1};                        // transmittance before my Gaussian, both pixels
            const f32x2 waT = alpha * T;
            const f32x2 wa = {act0 ? waT.x : 0.f, act1 ? waT.y : 0.f};
            const f32x2 D = {Dt[rr], Dt[rr + 1]};
            const f32x2 Dw = D * wa;
            float q0 = Dw.x, q1 = Dw.y;
            half_incl_scan_add2(q0, q1);
            const f32x2 tot = {half_last(q0, lane), half_last(q1, lane)};
            const f32x2 sy = {s0.y, s1.y};
            const f32x2 S = (tot - f32x2{q0, q1}) + sy;
            if (g > 0 && n == 0) { pd[w][pp].y = sy.x + tot.x; pd[w][pp + 1].y = sy.y + tot.y; }
            const f32x2 rom = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
            const f32x2 dLa_ = D * T - S * rom;
            const f32x2 dLa = {act0 ? dLa_.x : 0.f, act1 ? dLa_.y : 0.f};
            const f32x2 dLG = op * dLa;
            const f32x2 gdx = G * dx, gdy = G * dy;
            p_mx += dLG * (-gdx * cx - gdy * cy);
            p_my += dLG * (-gdy * cz - gdx * cy);
            p_cx += gdx * dx * dLG;
            p_cy += gdx * dy * dLG;
            p_cz += gdy * dy * dLG;
            p_op += G * dLa;
            touched = (act0 || act1) ? 1.0f : touched;
            p_r += wa * f32x2{dLT[F][pp], dLT[F][pp + 1]};
            p_g += wa * f32x2{dLT[F + 1][pp], dLT[F + 1][pp + 1]};
            p_b += wa * f32x2{dLT[F + 2][pp], dLT[F + 2][pp + 1]};
            if constexpr (NCT > 0) {
              if (use_feat) {
#pragma unroll
                for (int ct = 0; ct < NCT; ct++) {
                  const bool okc = 32 * ct + n < KCH;
                  const float a0 = okc ? dLT[32 * ct + n][pp] : 0.f, a1 = okc ? dLT[32 * ct + n][pp + 1] : 0.f;
                  Cf[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wa.x, Cf[ct], 0, 0, 0);
                  Cf[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wa.y, Cf[ct], 0, 0, 0);
                }
              }
            }
          }
          a_mx += p_mx.x + p_mx.y; a_my += p_my.x + p_my.y; a_cx += p_cx.x + p_cx.y; a_cy += p_cy.x + p_cy.y;
          a_cz += p_cz.x + p_cz.y; a_op += p_op.x + p_op.y; a_r += p_r.x + p_r.y; a_g += p_g.x + p_g.y; a_b += p_b.x + p_b.y;
        } else
#pragma unroll 1
        for (int rr = 0; rr < 16; rr++) {  // rolled: one pixel step's worth of registers (Dt[rr]: uniform index)
          if (((lmu >> ((rr & 3) + 8 * (rr >> 2))) & 0x11u) == 0u) continue;  // both pixels of the step dead in this group
          const int pp = 32 * u + (rr & 3) + 8 * (rr >> 2) + 4 * h;  // my pixel of this step
          const float4 st = pd[w][pp];
          const float Tg = (g == 0) ? st.x : st.w;                   // transmittance entering this group
          const float dx = ex - (bx0 + (float)((rr & 3) + 4 * h));
          const float dy = ey - (pyu + (float)(rr >> 2));
          const float power = gauss_power(cx, cy, cz, dx, dy);
          const float G = gm_exp<FAST>(power);
          const float alpha = fminf(0.99f, op * G);
          const bool act = pos <= __float_as_uint(st.z) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
          const float om = act ? 1.0f - alpha : 1.0f;
          const float T = Tg * half_excl_scan_mul(om, lane);         // transmittance before my Gaussian
          const float wa = act ? alpha * T : 0.f;
          const float D = Dt[rr];
          const float Dw = D * wa;
          const float P = half_incl_scan_add(Dw);
          const float tot = half_last(P, lane);
          const float S = (tot - P) + st.y;                          // D*alpha*T of everything behind me (+ bg term)
          if (g > 0 && n == 0) pd[w][pp].y = st.y + tot;             // group g-1 sees this group behind it
          const float dL_dalpha = act ? D * T - S * __builtin_amdgcn_rcpf(om) : 0.f;
          const float dL_dG = op * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          a_mx += dL_dG * (-gdx * cx - gdy * cy);
          a_my += dL_dG * (-gdy * cz - gdx * cy);
          a_cx += gdx * dx * dL_dG;
          a_cy += gdx * dy * dL_dG;
          a_cz += gdy * dy * dL_dG;
          a_op += G * dL_dalpha;
          touched = act ? 1.0f : touched;
          a_r += wa * dLT[F][pp];
          a_g += wa * dLT[F + 1][pp];
          a_b += wa * dLT[F + 2][pp];
          if constexpr (NCT > 0) {
            if (use_feat) {
#pragma unroll
              for (int ct = 0; ct < NCT; ct++) {
                const float a = (32 * ct + n < KCH) ? dLT[32 * ct + n][pp] : 0.f;  // A[i = channel 32ct+n][k = h]: my pixel's dL
                Cf[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wa, Cf[ct], 0, 0, 0);  // B[k = h][j = n] = wa
              }
            }
          }
        }
      }
      if (g > 0) wave_lds_sync();  // pd[].y updates visible before the next group reads them

      if (first_it) MGS_BTRACE(g == 0 ? 8 : 6);
      // ---- hand the group's sums to memory: transpose through LDS so that every atomic instruction covers whole
      //      rows (32 consecutive feature channels of one Gaussian = one 128-B line; 8 Gaussians x 6 geometry sums;
      //      16 Gaussians x 3 colour sums) instead of 64 different lines.  (Nothing to hand over if no pixel of the
      //      block blended anything of this group.) ----
      if (((g == 0 ? lm0 : lm1) & umask) != 0ull) {
        float v[9] = {a_mx * ddelx_dx, a_my * ddely_dy, -0.5f * a_cx, -0.5f * a_cy, -0.5f * a_cz, a_op, a_r, a_g, a_b};
#pragma unroll
        for (int i = 0; i < 9; i++) v[i] += __uint_as_float(lane_xor<32>(__float_as_uint(v[i]), lane));
        float* tr = trbuf[w];
        // a Gaussian no pixel of the block blended in this group adds exact zeros everywhere: its atomics are skipped
        const bool any_px = (touched + __uint_as_float(lane_xor<32>(__float_as_uint(touched), lane))) > 0.f;
        if (h == 0) gid[w][n] = (has && any_px) ? make_uint2(id, gidn) : make_uint2(0xffffffffu, 0u);
        if constexpr (NCT > 0) {
          if (use_feat) {
#pragma unroll
            for (int ct = 0; ct < NCT; ct++)
#pragma unroll
              for (int i = 0; i < 16; i++)
                tr[n * TROW + 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * h] = Cf[ct][i];
          }
        }
#pragma unroll
        for (int i = 0; i < 9; i++)
          if ((i & 1) == h) tr[n * TROW + NCT * 32 + i] = v[i];
        wave_lds_sync();
        // (Every batch below first reads ALL its operands from LDS -- ids and sums, pinned by an opaque use -- and only then
        //  issues its atomics.  Written as "read id, test, read sum, add" per instruction, the compiler emitted exactly that: two
        //  dependent LDS round trips and a branch in front of each of the 22 atomic instructions, 5-7 k cycles per hand-over
        //  and ~15 k of a block's 102 k at configs[2].)
        if constexpr (NCT > 0) {
          if (use_feat) {
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) {  // instruction k of a tile: Gaussians 2k, 2k+1 of the group, 32 channels each
              const int ch = 32 * ct + n;
              uint2 g2[16];
              float tv[16];
#pragma unroll
              for (int k = 0; k < 16; k++) {
                const int gg = 2 * k + h;
                g2[k] = gid[w][gg];
                tv[k] = tr[gg * TROW + ch];
              }
#pragma unroll
              for (int k = 0; k < 16; k++) asm volatile("" : "+v"(g2[k].x), "+v"(g2[k].y), "+v"(tv[k]));
#pragma unroll
              for (int k = 0; k < 16; k++)
                if (g2[k].x != 0xffffffffu && ch < F) unsafeAtomicAdd(dL_dfeat + (size_t)g2[k].y * F + ch, tv[k]);
            }
          }
        }
        {
          uint32_t gq[4];
          uint2 gc[2];
          float tq[4], tc[2];
          const int i8 = lane & 7, i4 = lane & 3;
#pragma unroll
          for (int k = 0; k < 4; k++) {  // geometry sums: 8 Gaussians x 8 slots (6 used) per instruction
            const int gg = 8 * k + (lane >> 3);
            gq[k] = gid[w][gg].x;
            tq[k] = tr[gg * TROW + NCT * 32 + (i8 < 6 ? i8 : 0)];
          }
#pragma unroll
          for (int k = 0; k < 2; k++) {  // colour sums: 16 Gaussians x 4 slots (3 used) per instruction
            const int gg = 16 * k + (lane >> 2);
            gc[k] = gid[w][gg];
            tc[k] = tr[gg * TROW + NCT * 32 + 6 + (i4 < 3 ? i4 : 0)];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) asm volatile("" : "+v"(gq[k]), "+v"(tq[k]));
#pragma unroll
          for (int k = 0; k < 2; k++) asm volatile("" : "+v"(gc[k].x), "+v"(gc[k].y), "+v"(tc[k]));
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (gq[k] != 0xffffffffu && i8 < 6) unsafeAtomicAdd(acc8 + (size_t)gq[k] * 8 + i8, tq[k]);
#pragma unroll
          for (int k = 0; k < 2; k++)
            if (gc[k].x != 0xffffffffu && i4 < 3)
              unsafeAtomicAdd(dL_dcolors + (size_t)(r.colors_per_view ? gc[k].x : gc[k].y) * 3 + i4, tc[k]);
        }
        wave_lds_sync();  // tr / gid are rewritten by the next group
        if (first_it) MGS_BTRACE(g == 0 ? 9 : 7);
      }
    }
  }
  MGS_BTRACE(15);
}

// ------------------------------------------- dispatch ------------------------------------------------
template <int F>
static hipError_t gm_F(const RenderArgs& r, const ImgView& im, const ChunkView& cv, const float* dc, const float* df,
                       float* acc8, float* dcol, float* dfeat, hipStream_t s) {
  const int T = r.tiles_x * r.tiles_y;
  const int grid = ((T + 7) / 8) * 32;
#define MGS_GM(FAST, NW, NWF, TWO, PAIR)                                                                              \
  hipLaunchKernelGGL((gm_bwd_kernel<F, FAST, NW, NWF, TWO, PAIR>), dim3(grid), dim3(NW * 64), 0, s, r, im.ranges,      \
                     cv.round_base, cv.last_chunk, cv.T_end, cv.last_pos, cv.partial, cv.q, im.final_T, dc, df, acc8, \
                     dcol, dfeat, cv.T_mid, cv.surv, cv.surv_stride, cv.nsurv)
#define MGS_GMF(NW, NWF, TWO, PAIR) do { if (r.fast_exp) MGS_GM(true, NW, NWF, TWO, PAIR); else MGS_GM(false, NW, NWF, TWO, PAIR); } while (0)
  if constexpr (F > 32) {
    MGS_GMF(8, 8, false, false);                 // wide rows: 8 waves x 256 registers (the forward ran 8 waves as well)
  } else if (r.nwf == 8) {
    MGS_GMF(8, 8, true, false);                  // more blocks than CUs: two 8-wave workgroups per CU
  } else if (r.gm_waves == 8) {
    MGS_GMF(8, 16, false, false);                // option: 8 waves x 256 registers behind a 16-wave forward
  } else if (r.gm_waves == 16) {
    MGS_GMF(16, 16, false, false);               // option: the rounds 2-4 form, 16 waves x 128 registers, one pixel per step
  } else {
    MGS_GMF(12, 16, false, true);                // default: 12 waves x 168 registers, two pixels per step
  }
#undef MGS_GMF
#undef MGS_GM
  return hipGetLastError();
}

hipError_t launch_render_bwd_gm(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                const float* dL_dcolor_px, const float* dL_dfeat_px, float* acc8, float* dL_dcolors,
                                float* dL_dfeat, hipStream_t s) {
  (void)b;
  const int F = r.include_feature ? r.F : 0;
  switch (F) {
#define X(N) case N: return gm_F<N>(r, im, cv, dL_dcolor_px, dL_dfeat_px, acc8, dL_dcolors, dL_dfeat, s);
    MGS_FOR_EACH_F(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mgs

// diagnostic: copy the backward's phase timeline out (count = 512 * 16 * 16 uint64)
extern "C" int mgs_debug_read_trace_bwd(unsigned long long* host, size_t count) {
  const size_t n = sizeof(mgs::g_btrace) / sizeof(unsigned long long);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mgs::g_btrace), (count < n ? count : n) * sizeof(unsigned long long));
}
