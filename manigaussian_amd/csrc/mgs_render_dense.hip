// mgs_render_dense.hip -- cooperative chunk-parallel render forward over SURVIVOR-DENSE chunks, gfx950.
//
// Results: the reference's renderCUDA forward (RAST/cuda_rasterizer/forward.cu:262-398), same per-pixel test order
// (power > 0 skip, alpha = min(0.99, o e^p), alpha < 1/255 skip, T(1-alpha) < 1e-4 stop WITHOUT blending) and stop rule.
//
// Decomposition: one workgroup of NW waves per 8x8 pixel block, lane = pixel.  The workgroup first compacts its tile's
// sorted list, in order, to the entries that reach ITS block (block-wide ballot/popcount scan; the survivors' instance ids
// go to a per-block list in memory that the rounds and later the backward read); a chunk is 64 consecutive SURVIVORS and a
// round is NW chunks, one per wave: phase A = per-chunk transmittance products, prefix in chunk order through LDS (every
// consumer sees bit-identical transmittances), phase B = blend from the exact T_in.  Records are gathered from geom.rec by
// id (L2-resident).  Per chunk the kernel keeps T_end, T_mid (transmittance after the first 32 survivors, where the
// backward's second group starts), the last blended position and the partial colour sums; the records come from a pool
// (ChunkView): one atomic per round takes the round's <= NW records.
#include "mgs_render_common.h"

namespace mgs {

// Phase timeline (diagnostic, MgsOptions.dbg = 256): s_memtime stamps per (workgroup, wave, event).
constexpr int TRACE_EVENTS = 24;
__device__ unsigned long long g_trace[512 * 16 * TRACE_EVENTS];
#define MGS_TRACE(ev)                                                                                      \
  do {                                                                                                     \
    if ((r.dbg & 256) && lane == 0 && blockIdx.x < 512 && (ev) < TRACE_EVENTS)                              \
      g_trace[((size_t)blockIdx.x * 16 + w) * TRACE_EVENTS + (ev)] = __builtin_amdgcn_s_memtime();         \
  } while (0)

// CHS = survivors per chunk = per wave and round (64: two full 32-lane groups for the backward; 32: one).
//
// Blend.  The phase timeline (scripts/trace_fwd.py) showed the LDS-staged broadcast rows of coop_fwd64_kernel to be the
// bottleneck of phase B: nine ds_read_b128 per entry and wave cost 72 LDS cycles each whatever the broadcast, times the
// waves of the CU.  Here no row goes through LDS:
//   * F >= 16: the feature contraction  C[pixel][ch] += w[pixel][entry] * feat[entry][ch]  runs on the matrix cores in
//     exact fp32 (v_mfma_f32_32x32x2_f32 = an fmaf chain).  The B operand is the feature row in its natural layout (lane
//     (k, ch) loads feat[entry 2kk + k][ch]: one coalesced 128-B row per entry, a ring of four pairs in flight); the A operand is
//     the blend weight the pixel lanes just computed: for an entry pair (j, j+1) ONE v_permlane32_swap turns
//     (w_j, w_j+1) into the operands of the two pixel tiles (pixels 0..31 / 32..63).  Accumulators leave through one LDS
//     transposition per chunk.
//   * the three colour channels (and every channel when F < 16) are FMAs against v_readlane broadcasts of the row
//     registers of the lane that owns the entry.
// TWO: two workgroups of this kernel share a CU (8 waves, <= 128 registers, <= 80 KB of LDS each)
template <int F, bool FAST, bool EXACT, int NW, int CHS, bool TWO>
__global__ void __launch_bounds__(NW * 64, TWO ? 4 : 1)
coop_fwd_dense_kernel(RenderArgs r, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      float* __restrict__ T_end, float* __restrict__ T_mid,
                      uint32_t* __restrict__ last_pos, float* __restrict__ partial, uint32_t* __restrict__ surv,
                      size_t surv_stride, uint2* __restrict__ nsurv, float* __restrict__ final_T,
                      uint32_t* __restrict__ last_chunk, float* __restrict__ out_color, float* __restrict__ out_feat,
                      uint32_t* __restrict__ round_base, uint32_t pool, uint32_t* __restrict__ flags, int nblocks,
                      uint64_t* host_status, uint32_t status_tag) {
  static_assert(CHS == 32 || CHS == 64, "a chunk is one or two 32-lane groups of the Gaussian-major backward");
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  constexpr int NCH = F + 3;
  constexpr bool MF = F >= 16;               // feature channels on the matrix cores
  constexpr int NT = MF ? (F + 31) / 32 : 0; // 32-channel tiles
  constexpr int NV = MF ? 3 : NCH;           // channels blended on the VALU: [features (F < 16)], r, g, b
  constexpr int NKK = CHS / 2;               // entry pairs per chunk
  constexpr int TRS = 65;                    // row stride of the transposition buffer (odd: conflict-free)
  constexpr int NOWN = (NCH + NW - 1) / NW;  // image channels owned by one wave
  constexpr uint32_t ROUND = NW * CHS;       // survivors blended per round
  constexpr uint32_t FSTEP = NW * 64;        // entries examined per fill sub-step (one per thread)
  constexpr int FILLK = 3;                   // sub-steps per fill step (all loads in flight together): at BASELINE configs[2]
                                             // a block needs ~2400 list entries for its first 1024 survivors -- one step, not two
  __shared__ float trs[MF ? NW * NT * 32 * TRS : 1];  // per wave: [channel][pixel] hand-over of the MFMA accumulators
  __shared__ float Tp[2][NW][64];            // per-chunk transmittance products, double buffered over rounds
  __shared__ float red_Tf[64];
  __shared__ uint32_t red_vis[64];
  __shared__ uint32_t cnt[2][FILLK * NW];    // survivors per (sub-step, wave) of a fill step, double buffered
  __shared__ uint32_t rbase[2];              // first chunk record of the round (pool index), double buffered over rounds
  constexpr uint32_t RBH = 16;               // rounds whose first record is also kept in LDS for the final sum
  __shared__ uint32_t rb_hist[RBH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: tell the compiler (scalar branches)
  int tile, sub;
  map_block(blockIdx.x, tile, sub);
  if (tile >= r.tiles_x * r.tiles_y) return;  // padding blocks of the grid (not counted in nblocks)
  const PixBlk p = pix_blk(r, tile, sub, lane);
  const uint2 rng = ranges[tile];
  const bool use_feat = (F > 0) && r.include_feature;
  uint32_t* __restrict__ my_rounds = round_base + round_entry(rng.x, tile, sub, 0);  // entry of round k: my_rounds[4 * k]
  bool overflow = false;   // the chunk pool ran out (workgroup-uniform): stop, the host will see the flag
  uint32_t* my_surv = surv + (size_t)sub * surv_stride + rng.x;  // this block's compacted list of instance ids: at most len entries

  float Tround = 1.0f;
  uint32_t my_vis = 0;
  float my_Tf = 1.0f;
  MGS_TRACE(0);

  // all of these are workgroup-uniform (every thread derives them from the same LDS counts)
  uint32_t next = rng.x;   // next list entry to examine
  uint32_t qhead = 0;      // survivors consumed so far
  uint32_t qtail = 0;      // survivors found so far
  uint32_t cbase = 0;      // dense index of this round's first chunk
  uint32_t fill = 0;       // fill steps done (parity selects the cnt buffer)
  uint32_t round = 0;

  const bool allocator = tid == (NW - 1) * 64;  // lane 0 of the last wave takes the rounds' chunk records from the pool
  for (;; round++) {
    // This round's chunk records: NW consecutive ones (a round blends at most NW chunks; the pool is sized from what
    // forwards actually took, so rounding up costs memory, not correctness), requested BEFORE the fill so that the
    // returning atomic's round trip hides behind it.
    uint32_t b0 = 0;
    const bool more = next < rng.y || qtail > qhead;
    if (allocator && more) b0 = atomicAdd(&flags[FLAG_CHUNKS_USED], (uint32_t)NW);
    // ---- fill: examine FILLK * FSTEP entries per step until a full round of survivors waits (or the list ends);
    //      survivors go, in list order, to this block's list in memory (read back below and by the backward) ----
    while (qtail - qhead < ROUND && next < rng.y) {
      uint32_t idd[FILLK], rk[FILLK];
      {
        float4 a0[FILLK], a1[FILLK];
#pragma unroll
        for (int k = 0; k < FILLK; k++) {
          const uint32_t e = next + (uint32_t)k * FSTEP + (uint32_t)tid;
          idd[k] = e < rng.y ? point_list[e] : 0xffffffffu;
        }
#pragma unroll
        for (int k = 0; k < FILLK; k++) {
          a0[k] = make_float4(0, 0, 0, 0); a1[k] = make_float4(0, 0, -1.f, -1.f);
          if (idd[k] != 0xffffffffu) { a0[k] = r.rec[2 * (size_t)idd[k]]; a1[k] = r.rec[2 * (size_t)idd[k] + 1]; }
        }
#pragma unroll
        for (int k = 0; k < FILLK; k++) {
          const bool sk = idd[k] != 0xffffffffu && cull_ok<EXACT>(a0[k], a1[k], p);
          const unsigned long long sm = ballot(sk);
          rk[k] = sk ? (uint32_t)__builtin_popcountll(sm & ((1ull << lane) - 1ull)) : 0xffffffffu;
          if (lane == 0) cnt[fill & 1][k * NW + w] = (uint32_t)__builtin_popcountll(sm);
        }
      }
      __syncthreads();
      // exclusive prefix over the FILLK * NW counters (sub-step major = list order), one counter per lane
      const uint32_t v = lane < FILLK * NW ? cnt[fill & 1][lane] : 0u;
      uint32_t incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        incl += lane >= d ? up : 0u;
      }
      const uint32_t total = bcast_lane_u32(incl, 63);
#pragma unroll
      for (int k = 0; k < FILLK; k++) {
        const uint32_t base = bcast_lane_u32(incl - v, k * NW + w);
        if (rk[k] != 0xffffffffu) my_surv[qtail + base + rk[k]] = idd[k];
      }
      qtail += total;
      next += FILLK * FSTEP;
      fill++;
    }
    if (allocator && more) {
      rbase[round & 1] = b0;
      if (round < RBH) rb_hist[round] = b0;
      my_rounds[4 * (size_t)round] = b0;  // for the backward
    }
    MGS_TRACE(1 + 8 * round);
    __syncthreads();  // the list is written (workgroup scope); the previous round's Tp readers are done
    MGS_TRACE(2 + 8 * round);
    const uint32_t avail = qtail - qhead;
    const bool exhausted = !(next < rng.y);
    uint32_t nchunk = exhausted ? (avail + CHS - 1) / CHS : avail / CHS;
    nchunk = min(nchunk, (uint32_t)NW);
    if (nchunk == 0) break;
    const uint32_t c = cbase + (uint32_t)w;
    const bool has = (uint32_t)w < nchunk;
    const uint32_t n_my = has ? min((uint32_t)CHS, avail - (uint32_t)w * CHS) : 0u;  // survivors of my chunk
    const bool valid = (uint32_t)lane < n_my;
    // ---- my chunk: lane e holds entry e's record and its VALU-blended row; the feature rows go to the B operands ----
    float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
    float rowv[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) rowv[i] = 0.f;
    uint32_t gidl = 0;  // Gaussian (feature row) of my entry
    if (valid) {
      const uint32_t id = my_surv[qhead + (uint32_t)w * CHS + (uint32_t)lane];
      g0 = r.rec[2 * (size_t)id]; g1 = r.rec[2 * (size_t)id + 1];
      const uint32_t gid = gauss_of(r, id);
      gidl = gid;
      const uint32_t cid = r.colors_per_view ? id : gid;  // colour row: per view when it comes from SH
      if constexpr (!MF && F > 0) {
        if (use_feat) {
#pragma unroll
          for (int i = 0; i < F; i++) rowv[i] = r.feats[(size_t)gid * F + i];
        }
      }
#pragma unroll
      for (int i = 0; i < 3; i++) rowv[NV - 3 + i] = r.colors[(size_t)cid * 3 + i];
    }
    // B operand of entry pair kk: lane (k = lane >> 5, ch = lane & 31) loads feat[entry 2kk + k][32 t + ch].  A ring of BD
    // pairs is kept in flight (loaded BD pairs ahead of their use): the whole chunk's operands would not fit the registers.
    constexpr int BD = 4;
    auto load_B = [&](float (&B)[NT > 0 ? NT : 1][BD], int kk) {
      const uint32_t gA = bcast_lane_u32(gidl, 2 * kk), gB = bcast_lane_u32(gidl, 2 * kk + 1);
      const uint32_t ent = 2u * kk + (uint32_t)(lane >> 5);
      const uint32_t gide = lane < 32 ? gA : gB;
#pragma unroll
      for (int t = 0; t < (NT > 0 ? NT : 1); t++) {
        const int ch = 32 * t + (lane & 31);
        B[t][kk % BD] = (MF && ent < n_my && ch < F && use_feat) ? r.feats[(size_t)gide * F + ch] : 0.f;
      }
    };
    float Bq[NT > 0 ? NT : 1][BD];
    if constexpr (MF) {
#pragma unroll
      for (int kk = 0; kk < BD; kk++) load_B(Bq, kk);
    }
    MGS_TRACE(3 + 8 * round);
    // alpha of entry j for my pixel with the reference's two skip tests folded in (forward.cu:345-356): 0 = skipped.
    // (1 - 0 = 1 exactly, so a skipped entry leaves every product bit for bit alone.)
    auto alpha_of = [&](int j) -> float {
      const float ex = bcast_lane(g0.x, j), ey = bcast_lane(g0.y, j);
      const float cx = bcast_lane(g0.z, j), cy = bcast_lane(g0.w, j), cz = bcast_lane(g1.x, j);
      const float op = bcast_lane(g1.y, j);
      const float dx = ex - p.pxf, dy = ey - p.pyf;
      const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
      const float alpha = fminf(0.99f, op * exp_<FAST>(power));
      return ((power > 0.0f) || (alpha < 1.0f / 255.0f)) ? 0.f : alpha;
    };
    // ---- phase A: transmittance product of this chunk (four independent alphas in flight, then the product chain) ----
    float tp = 1.0f;
    if (n_my == (uint32_t)CHS) {
#pragma unroll
      for (int j = 0; j < CHS; j += 4) {
        const float a0 = alpha_of(j), a1 = alpha_of(j + 1), a2 = alpha_of(j + 2), a3 = alpha_of(j + 3);
        tp = tp * (1.0f - a0); tp = tp * (1.0f - a1); tp = tp * (1.0f - a2); tp = tp * (1.0f - a3);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      for (uint32_t j = 0; j < n_my; j++) tp = tp * (1.0f - alpha_of((int)j));
    }
    MGS_TRACE(4 + 8 * round);
    Tp[round & 1][w][lane] = tp;
    __syncthreads();
    MGS_TRACE(5 + 8 * round);
    const uint32_t rb = rbase[round & 1];
    if (rb + (uint32_t)NW > pool) { overflow = true; break; }  // uniform: every thread reads the same word
    // ---- prefix in chunk order (identical arithmetic in every wave) ----
    float T = Tround, Tnext = Tround;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) {
      const float t2 = Tp[round & 1][w2][lane];
      T = (w2 < w) ? T * t2 : T;
      Tnext *= t2;
    }
    // ---- phase B: blend this chunk ----
    const bool live = has && p.inside && !(T < 0.0001f);
    if (ballot(live) != 0) {
      float C[NCH];
#pragma unroll
      for (int i = 0; i < NCH; i++) C[i] = 0.f;
      f32x16 acc[NT > 0 ? NT : 1][2];
#pragma unroll
      for (int t = 0; t < (NT > 0 ? NT : 1); t++)
#pragma unroll
        for (int i = 0; i < 16; i++) { acc[t][0][i] = 0.f; acc[t][1][i] = 0.f; }
      float alive = live ? 1.0f : 0.0f;  // 0 once the pixel has terminated (kept in a VGPR: no scalar mask algebra per entry)
      uint32_t last = 0;
      float Tm = T;  // transmittance entering the chunk's second group of 32 (CHS == 64)
      // one entry of the reference's per-pixel walk (forward.cu:357-380) given its alpha: returns the blend weight
      // alpha * T (0: not blended) and advances T / alive / last.  A live pixel always has T >= 1e-4 (it entered so, and
      // a blend only happens when the new T stays above), hence alpha == 0 (a skipped entry) can never trip the stop
      // test and needs no test of its own; T * (1 - a) is the reference's test_T bit for bit when a == alpha.
      auto advance = [&](int j, float alpha) -> float {
        const float test_T = T * (1.0f - alpha);
        const bool term = test_T < 0.0001f;
        const float a = (term ? 0.f : alpha) * alive;  // the entry's alpha if it is blended for this pixel, else 0
        alive = term ? 0.f : alive;
        const float wgt = a * T;
        T = T * (1.0f - a);
        last = wgt > 0.f ? (uint32_t)j + 1u : last;
        return wgt;
      };
      bool stop = false;  // wave-uniform: the chunk is exhausted or every pixel has terminated
#pragma unroll
      for (int kq = 0; kq < NKK / 2; kq++) {  // four entries = two MFMA pairs per step
        const int j = 4 * kq;
        if (CHS > 32 && j == 32) Tm = T;
        stop = stop || (uint32_t)j >= n_my || ballot(alive != 0.f) == 0;
        __builtin_amdgcn_sched_barrier(0);  // one quad at a time: hoisting later quads' alpha maths only adds live registers
        if (!stop) {
          float a[4], wq[4];
#pragma unroll
          for (int u = 0; u < 4; u++) a[u] = ((uint32_t)(j + u) < n_my) ? alpha_of(j + u) : 0.f;  // independent
#pragma unroll
          for (int u = 0; u < 4; u++) wq[u] = advance(j + u, a[u]);                                  // the serial chain
          if (ballot(wq[0] != 0.f || wq[1] != 0.f || wq[2] != 0.f || wq[3] != 0.f) != 0) {
#pragma unroll
            for (int i = 0; i < NV; i++) {
              const int ci = MF ? i : (i < F ? 3 + i : i - F);  // rowv = [features (F < 16)], r, g, b -> C = r, g, b, features
#pragma unroll
              for (int u = 0; u < 4; u++) C[ci] += bcast_lane(rowv[i], j + u) * wq[u];
            }
            if constexpr (MF) {
#pragma unroll
              for (int h2 = 0; h2 < 2; h2++) {
                const int kk = 2 * kq + h2;
                float w0 = wq[2 * h2], w1 = wq[2 * h2 + 1];
                swap32(w0, w1);  // w0: pixels 0..31 x (entry 2kk | 2kk+1), w1: pixels 32..63 x (entry 2kk | 2kk+1)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                  acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, Bq[t][kk % BD], acc[t][0], 0, 0, 0);
                  acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, Bq[t][kk % BD], acc[t][1], 0, 0, 0);
                }
              }
            }
          }
          if constexpr (MF) {
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {  // refill the two ring slots just used
              const int kk = 2 * kq + h2;
              if (kk + BD < NKK && (uint32_t)(2 * (kk + BD)) < n_my) load_B(Bq, kk + BD);
            }
          }
        }
      }
      if (last <= 32u) Tm = T;  // nothing of the second group was blended for this pixel (Tm is then never used)
      if constexpr (MF) {
        // accumulators (col = channel lane & 31, row = pixel (i & 3) + 8 (i >> 2) + 4 (lane >> 5) of the tile) -> C[pixel lane]
        float* tb = trs + (size_t)w * NT * 32 * TRS;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
          for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
            for (int i = 0; i < 16; i++)
              tb[(32 * t + (lane & 31)) * TRS + 32 * h2 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)] = acc[t][h2][i];
        wave_lds_sync();
#pragma unroll
        for (int ch = 0; ch < F; ch++) C[3 + ch] = tb[ch * TRS + lane];
        wave_lds_sync();  // the next round's writes come after these reads
      }
      MGS_TRACE(6 + 8 * round);
      const size_t slot = (size_t)rb + (size_t)w;
      T_end[slot * 64 + lane] = T;
      if (CHS > 32) T_mid[slot * 64 + lane] = Tm;
      last_pos[slot * 64 + lane] = last;
      float* pp = partial + slot * NCH * 64 + lane;
#pragma unroll
      for (int i = 0; i < NCH; i++) pp[i * 64] = C[i];
      if (live) { my_vis = c + 1; my_Tf = T; }
    }
    MGS_TRACE(7 + 8 * round);
    Tround = Tnext;
    qhead += min(avail, nchunk * CHS);
    cbase += nchunk;
    if (ballot(p.inside && !(Tround < 0.0001f)) == 0) break;
  }

  MGS_TRACE(TRACE_EVENTS - 3);
  // This block has taken its last chunk records: count it now (the returned ticket is looked at after the final sum, so
  // the atomic's round trip is hidden); whoever draws the last ticket reports the pool usage to the host.
  uint32_t ticket = 0;
  if (tid == 0) {
    // (device-scope atomics served by the L2; the OR's returned value feeds the ticket, so it has been performed when the
    //  ticket is counted -- no fence: a fence here costs every block ~3 us)
    const uint32_t dep = overflow ? (atomicOr(&flags[FLAG_PREFILTERED], 0x100u) & 0u) : 0u;
    ticket = atomicAdd(&flags[FLAG_BLOCKS_DONE], 1u + dep) + 1u;
  }
  if (w == 0) { red_vis[lane] = 0; red_Tf[lane] = 1.0f; }
  __syncthreads();  // also: every wave's partial sums are written (workgroup scope)
  if (my_vis > 0) atomicMax(&red_vis[lane], my_vis);
  __syncthreads();
  const uint32_t vis = red_vis[lane];
  if (my_vis > 0 && my_vis == vis) red_Tf[lane] = my_Tf;  // exactly one wave owns the last visited chunk
  __syncthreads();
  const float Tf = red_Tf[lane];
  // ---- image = sum of the visited chunks' partial colours, in chunk order; wave w owns channels w, w + NW, ... ----
  float img[NOWN];
#pragma unroll
  for (int k = 0; k < NOWN; k++) img[k] = 0.f;
  const uint32_t vmax = wave_umax(vis);
  constexpr int NFLY = 8;  // records whose loads are in flight together (a block visits ~7 chunks at BASELINE configs[2])
  for (uint32_t c0 = 0; c0 < vmax; c0 += NFLY) {  // the sum stays in chunk order
    float v[NFLY][NOWN];
#pragma unroll
    for (int u = 0; u < NFLY; u++) {
      const uint32_t cc = c0 + u;
      const uint32_t rr = cc / NW;
      const size_t slot = cc < vis ? (size_t)(rr < RBH ? rb_hist[rr] : my_rounds[4 * (size_t)rr]) + (cc % NW) : 0;
      const float* pp = partial + slot * NCH * 64 + lane;
#pragma unroll
      for (int k = 0; k < NOWN; k++) {
        const int ch = w + k * NW;
        v[u][k] = (cc < vis && ch < NCH && (ch < 3 || use_feat)) ? pp[ch * 64] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < NFLY; u++)
#pragma unroll
      for (int k = 0; k < NOWN; k++) img[k] += v[u][k];
  }
  MGS_TRACE(TRACE_EVENTS - 2);
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
    if (lane == 0) {
      nsurv[(size_t)tile * 4 + sub] = make_uint2(qtail, round > 0 || vis > 0 ? rb_hist[0] : 0u);  // + round 0's first record
      // the workgroup that drew the last ticket reports {tag, overflow, chunk records used} to the host (mapped pinned memory)
      if (ticket == (uint32_t)nblocks && host_status) {
        const uint32_t used = __hip_atomic_load(&flags[FLAG_CHUNKS_USED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t ovf = (__hip_atomic_load(&flags[FLAG_PREFILTERED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 8) & 1u;
        __hip_atomic_store(host_status + 1, ((uint64_t)(status_tag & 0xffffu) << 48) | ((uint64_t)ovf << 32) | used,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  MGS_TRACE(TRACE_EVENTS - 1);
}

// ------------------------------------------- dispatch ------------------------------------------------
template <int F>
static hipError_t dense_F(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv, float* oc,
                          float* of, StatusSink st, hipStream_t s) {
  const int T = r.tiles_x * r.tiles_y;
  const int grid = ((T + 7) / 8) * 32;
#define MGS_CFD_(FAST, EXACT, NW, TWO)                                                                                \
  hipLaunchKernelGGL((coop_fwd_dense_kernel<F, FAST, EXACT, NW, CHUNK, TWO>), dim3(grid), dim3(NW * 64), 0, s, r,      \
                     im.ranges, b.point_list, cv.T_end, cv.T_mid, cv.last_pos, cv.partial, cv.surv,                   \
                     cv.surv_stride, cv.nsurv, im.final_T, cv.last_chunk, oc, of, cv.round_base, cv.pool, im.flags,   \
                     4 * T, st.host, st.tag)
#define MGS_CFD(FAST, EXACT)                                                                                          \
  do {                                                                                                                \
    if constexpr (F > 32) MGS_CFD_(FAST, EXACT, 8, false);          /* 256 registers per lane */                     \
    else if (r.nwf == 8) MGS_CFD_(FAST, EXACT, 8, true);            /* two workgroups per CU */                        \
    else MGS_CFD_(FAST, EXACT, 16, false);                                                                            \
  } while (0)
  if (r.fast_exp) { if (r.exact_cull) MGS_CFD(true, true); else MGS_CFD(true, false); }
  else            { if (r.exact_cull) MGS_CFD(false, true); else MGS_CFD(false, false); }
#undef MGS_CFD
#undef MGS_CFD_
  return hipGetLastError();
}

hipError_t launch_render_fwd_dense(const RenderArgs& r, const BinView& b, const ImgView& im, const ChunkView& cv,
                                   float* out_color, float* out_feat, StatusSink st, hipStream_t s) {
  const int F = r.include_feature ? r.F : 0;
  switch (F) {
#define X(N) case N: return dense_F<N>(r, b, im, cv, out_color, out_feat, st, s);
    MGS_FOR_EACH_F(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mgs

// diagnostic: copy the phase timeline out (count = 512 * 16 * 24 uint64); not part of include/mgsplat.h
extern "C" int mgs_debug_read_trace(unsigned long long* host, size_t count) {
  const size_t n = sizeof(mgs::g_trace) / sizeof(unsigned long long);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mgs::g_trace), (count < n ? count : n) * sizeof(unsigned long long));
}
