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

// Blend: wave PAIRS, lane = (pixel of a half block, entry parity).
//
// Round 2's kernel gave a whole wave to a chunk with lane = pixel and walked the chunk's 64 entries one by one, every
// entry's record broadcast to the 64 pixel lanes with six v_readlane, and evaluated every alpha twice (phase A for the
// transmittance product, phase B for the blend): 97 VALU instructions per entry and wave (6 200 per chunk), ~7 live waves of
// 16 at BASELINE configs[2] -- latency-bound at 14-22 % VALU issue (profiles/r02_sq_counters.json).  Here TWO waves share
// a chunk, one per half block (8 x 4 pixels), and a lane is (pixel p, entry parity k): step s evaluates entries 2s and
// 2s + 1 for 32 pixels -- 64 distinct (pixel, entry) pairs per instruction, no broadcast:
//   * the chunk's records are staged once in the wave's LDS; a lane reads its entry's record with two broadcast
//     ds_reads (32 lanes per address);
//   * phase A keeps its 32 alphas per lane in registers, so phase B evaluates no exp at all;
//   * the serial per-pixel chain (forward.cu:357-380) is 32 steps of two entries: ONE v_permlane32_swap hands each half
//     the other parity's alpha, both lanes of a pixel then run the identical chain;
//   * lane (p, k) IS the A-operand layout of v_mfma_f32_32x32x2_f32 (row = pixel, k = entry of the pair) and lane (ch, k)
//     the B layout: one MFMA per step and 32 channels, no operand shuffles; the accumulators leave through one LDS
//     transposition per chunk;
//   * twice the live waves per block, each with a quarter of the instructions.
// A pair blends up to NS = 2 chunks per round (chunks pr and pr + NW/2), so a round still covers NW chunks: blocks whose
// pixels need more than NW/2 chunks (15 % of them at configs[2]) do not pay a second fill / barrier round.
// Per-pixel results are those of the reference's walk: same alpha expression, same test order, same sequential T chain.
// TWO: two workgroups of this kernel share a CU (8 waves, <= 128 registers, <= 80 KB of LDS each)
template <int F, bool FAST, bool EXACT, int NW, int CHS, bool TWO>
__global__ void __launch_bounds__(NW * 64, TWO ? 4 : 1)
coop_fwd_pairs_kernel(RenderArgs r, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      float* __restrict__ T_end, float* __restrict__ T_mid,
                      uint32_t* __restrict__ last_pos, float* __restrict__ partial, uint32_t* __restrict__ surv,
                      size_t surv_stride, uint2* __restrict__ nsurv, float* __restrict__ final_T,
                      uint32_t* __restrict__ last_chunk, float* __restrict__ out_color, float* __restrict__ out_feat,
                      uint32_t* __restrict__ round_base, uint32_t pool, uint32_t* __restrict__ flags, int nblocks,
                      uint64_t* host_status, uint32_t status_tag, unsigned long long* hs_fail_mark) {
  static_assert(CHS == 64, "a chunk is two 32-lane groups of the Gaussian-major backward");
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  constexpr int NCH = F + 3;
  constexpr bool MF = F >= 16;               // feature channels on the matrix cores
  constexpr int NT = MF ? (F + 31) / 32 : 0; // 32-channel tiles
  constexpr int NVF = MF ? 0 : F;            // feature channels blended on the VALU
  constexpr int NP = NW / 2;                 // wave pairs
  constexpr int NS = 2;                      // chunks a pair blends per round
  constexpr int NSTEP = CHS / 2;             // steps per chunk: two entries each
  constexpr int TRS = 33;                    // row stride of the transposition buffer [channel][32 pixels] (odd: conflict-free)
  constexpr int NOWN = (NCH + NW - 1) / NW;  // image channels owned by one wave (final sum)
  constexpr uint32_t ROUND = NW * CHS;       // survivors blended per round
  constexpr uint32_t FSTEP = NW * 64;        // entries examined per fill sub-step (one per thread)
  constexpr int FILLK = 3;                   // sub-steps per fill step (all loads in flight together): at BASELINE configs[2]
                                             // a block needs ~2400 list entries for its first 1024 survivors -- one step, not two
  constexpr int TRR = NT * 32 + 3;           // rows of a wave's hand-over buffer: the feature channels, then r, g, b
  __shared__ float trs[MF ? NW * TRR * TRS : 1];      // per wave: [channel][pixel] hand-over of the MFMA accumulators; after the
                                             // rounds it still holds the wave's LAST chunk: the final sum reads that one here
  __shared__ uint32_t lastcc[NW];            // ... its dense chunk index (0xffffffff: the wave blended nothing)
  // per wave, the chunk under evaluation, laid out for phase A's DOUBLE steps: the lanes of parity k evaluate entries
  // ea = 4 j + k and eb = ea + 2 in double step j, and (j, k) owns 12 consecutive floats {x_a, x_b, y_a, y_b | cx_a, cx_b, cy_a,
  // cy_b | cz_a, cz_b, o_a, o_b} (conic = (cx, cy, cz); opacity 0: no entry) -- three 16-byte reads whose register pairs are
  // the operands of the packed fp32 arithmetic as they arrive (the compiler's own pairing of two steps spent 8 of its 57
  // instructions per double step on v_mov shuffles)
  __shared__ float4 recP[NW][CHS / 4 * 2 * 3];  // indexed by the chunk's number in the round (both waves of a pair read one copy)
  __shared__ uint32_t sid[NW * CHS];         // round 0: the survivors' instance ids, in list order (written by the fill)
  __shared__ float4 rowq[NW][NS][CHS];       // per wave and chunk slot, for the blend: {r, g, b, Gaussian (bits)}
  __shared__ float rowf[NVF > 0 ? NW * NS * CHS * NVF : 1];  // ... and the feature row when it is blended on the VALU
  __shared__ float Tp[2][NW][64];            // per-chunk transmittance products [chunk of the round][pixel], double buffered
  __shared__ unsigned long long red_last[64];  // per pixel: (last visited chunk + 1) << 32 | bits of the transmittance after it
  __shared__ uint32_t cnt[2][FILLK * NW];    // survivors per (sub-step, wave) of a fill step, double buffered
  __shared__ uint32_t rbase[2];              // first chunk record of the round (pool index), double buffered over rounds
  constexpr uint32_t RBH = 16;               // rounds whose first record is also kept in LDS for the final sum
  __shared__ uint32_t rb_hist[RBH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: tell the compiler (scalar branches)
  // direct binning (no bin scatter launch): ImgView::ready[1], the preprocess's "a workgroup gave up waiting" mark, was read by
  // every workgroup of the bucket rank, the launch before this one; cleared here so that a replayed HIP graph -- same buffer,
  // same nonce -- does not see a stale failure (with a scatter launch the bucket rank clears it: nullptr here)
  if (blockIdx.x == 0 && tid == 0 && hs_fail_mark) *hs_fail_mark = 0ull;
  const int pr = w >> 1, hh = w & 1;         // my pair; my half block (pixel rows 4 hh .. 4 hh + 3)
  const int k_ = lane >> 5, pl_ = lane & 31; // my entry parity; my pixel inside the half block
  int tile, sub;
  map_block(blockIdx.x, tile, sub);
  if (tile >= r.tiles_x * r.tiles_y) return;  // padding blocks of the grid (not counted in nblocks)
  const PixBlk p = pix_blk(r, tile, sub, lane);            // lane = pixel: the fill's cull and the final image sum
  const int pixq_ = 32 * hh + pl_;                         // my pixel in the blend phases
  const int pixq = pixq_, k = k_, pl = pl_;
  const PixBlk pq = pix_blk(r, tile, sub, pixq);
  const int pixx_ = 32 * (hh ^ k) + pl;                    // the pixel whose round-to-round transmittance this lane tracks:
  const int pixx = pixx_;
  const bool insidex = pix_blk(r, tile, sub, pixx).inside; // k = 0 lanes their own, k = 1 lanes the other half's (64 in all)
  const uint2 rng = ranges[tile];
  const bool use_feat = (F > 0) && r.include_feature;
  // the feature table as a buffer resource (wave-uniform: kernel argument); rows are addressed by 32-bit byte offsets, which
  // the host side guarantees to fit (mgs_api.hip: P F 4 < 2^32)
  const auto feat_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(r.feats), 0, 0xffffffffu, 0x00020000);
  uint32_t* __restrict__ my_rounds = round_base + round_entry(rng.x, tile, sub, 0);  // entry of round k: my_rounds[4 * k]
  bool overflow = false;   // the chunk pool ran out (workgroup-uniform): stop, the host will see the flag
  uint32_t* my_surv = surv + (size_t)sub * surv_stride + rng.x;  // this block's compacted list of instance ids: at most len entries

  float Tround = 1.0f;     // transmittance of pixel pixx entering the round
  uint32_t my_vis = 0;
  float my_Tf = 1.0f;
  if (w == 0) red_last[lane] = (unsigned long long)__float_as_uint(1.0f);  // (ordered by the first round's barrier)
  uint32_t my_lastcc = 0xffffffffu;  // wave-uniform
  MGS_TRACE(0);

  // all of these are workgroup-uniform (every thread derives them from the same LDS counts)
  uint32_t next = rng.x;   // next list entry to examine
  uint32_t qhead = 0;      // survivors consumed so far
  uint32_t qtail = 0;      // survivors found so far
  uint32_t cbase = 0;      // dense index of this round's first chunk
  uint32_t fill = 0;       // fill steps done (parity selects the cnt buffer)
  uint32_t round = 0;

  const bool allocator = tid == (NW - 1) * 64;  // lane 0 of the last wave takes the rounds' chunk records from the pool
  // (workgroup- and grid-uniform) grids of at most 1024 blocks whose pool holds a first round for every block
  const bool static0 = (uint32_t)nblocks <= 1024u && (uint32_t)nblocks * (uint32_t)NW <= pool;
  const uint32_t dyn_base = static0 ? (uint32_t)nblocks * (uint32_t)NW : 0u;  // the atomically taken records lie behind the slices
  for (;; round++) {
    // (lane-derived indices pass through an opaque asm once per round: otherwise the compiler hoists the address arithmetic
    //  of every unrolled LDS / global access below out of the round loop and spills it: 152 bytes of scratch per lane)
    int k = k_, pl = pl_, pixq = pixq_, pixx = pixx_;
    asm volatile("" : "+v"(k), "+v"(pl), "+v"(pixq), "+v"(pixx));
    // This round's chunk records: NW consecutive ones (a round blends at most NW chunks; the pool is sized from what
    // forwards actually took, so rounding up costs memory, not correctness), requested BEFORE the fill so that the
    // returning atomic's round trip hides behind it.
    uint32_t b0 = 0;
    const bool more = next < rng.y || qtail > qhead;
    if (allocator && more) {
      // (round 0 of a small grid: a fixed slice of the pool, no atomic.  At 128 x 128 every one of the 256 blocks asked
      //  for its first records at the same moment, 256 returning atomics on ONE address: the L2 serves those one after
      //  the other, and the last block got its answer ~15 k cycles later -- the whole "fill" phase of the timeline)
      if (static0 && round == 0) b0 = ((uint32_t)tile * 4u + (uint32_t)sub) * (uint32_t)NW;
      else b0 = dyn_base + atomicAdd(&flags[FLAG_CHUNKS_USED], (uint32_t)NW);
    }
    // ---- fill: examine FILLK * FSTEP entries per step until a full round of survivors waits (or the list ends);
    //      survivors go, in list order, to this block's list in memory (read back below and by the backward) ----
    while (qtail - qhead < ROUND && next < rng.y) {
      uint32_t idd[FILLK], rk[FILLK];
      float4 a0[FILLK], a1[FILLK];
      {
#pragma unroll
        for (int kf = 0; kf < FILLK; kf++) {
          const uint32_t e = next + (uint32_t)kf * FSTEP + (uint32_t)tid;
          idd[kf] = e < rng.y ? point_list[e] : 0xffffffffu;
        }
#pragma unroll
        for (int kf = 0; kf < FILLK; kf++) {
          a0[kf] = make_float4(0, 0, 0, 0); a1[kf] = make_float4(0, 0, -1.f, -1.f);
          if (idd[kf] != 0xffffffffu) { a0[kf] = r.rec[2 * (size_t)idd[kf]]; a1[kf] = r.rec[2 * (size_t)idd[kf] + 1]; }
        }
#pragma unroll
        for (int kf = 0; kf < FILLK; kf++) {
          const bool sk = idd[kf] != 0xffffffffu && cull_ok<EXACT>(a0[kf], a1[kf], p);
          const unsigned long long sm = ballot(sk);
          rk[kf] = sk ? (uint32_t)__builtin_popcountll(sm & ((1ull << lane) - 1ull)) : 0xffffffffu;
          if (lane == 0) cnt[fill & 1][kf * NW + w] = (uint32_t)__builtin_popcountll(sm);
        }
      }
      __syncthreads();
      // exclusive prefix over the FILLK * NW counters (sub-step major = list order), one counter per lane
      const uint32_t v = lane < FILLK * NW ? cnt[fill & 1][lane] : 0u;
      const uint32_t incl = wave_incl_scan_add_u32(v);
      const uint32_t total = bcast_lane_u32(incl, 63);
#pragma unroll
      for (int kf = 0; kf < FILLK; kf++) {
        const uint32_t base = bcast_lane_u32(incl - v, kf * NW + w);
        if (rk[kf] != 0xffffffffu) {
          const uint32_t n = qtail + base + rk[kf];
          my_surv[n] = idd[kf];
          // Round 0 (qhead = 0): the survivor's record goes straight into the LDS buffer of its chunk, in phase A's layout, and
          // its id beside it -- phase A used to fetch both again through memory (list -> record: two dependent round trips
          // of ~4 k cycles per chunk slot, on every block's critical path).  Later rounds (18 of 256 blocks at configs[2]) and
          // survivors beyond the round's window take the old way.
          if (round == 0 && n < ROUND) {
            const uint32_t e = n & (CHS - 1);
            float* rp = reinterpret_cast<float*>(&recP[n / CHS][0]) + (((e >> 2) * 2 + (e & 1)) * 12 + ((e >> 1) & 1));
            rp[0] = a0[kf].x; rp[2] = a0[kf].y; rp[4] = a0[kf].z; rp[6] = a0[kf].w; rp[8] = a1[kf].x; rp[10] = a1[kf].y;
            sid[n] = idd[kf];
          }
        }
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

    // ---- phase A: my pair's chunks pr and pr + NP: stage the records, alpha of (my pixel, my entries), the products ----
    // alpha of entry 2s + k for pixel pixq with the reference's two skip tests folded in (forward.cu:345-356; 0 = skipped:
    // 1 - 0 = 1 exactly, a skipped entry leaves every product alone), from the records staged in this wave's LDS
    // (two at a time: steps 2 j and 2 j + 1, i.e. entries 4 j + k and 4 j + 2 + k, in packed fp32 -- the same roundings as the
    //  scalar form, mgs_selftest bits 28 / 29)
    struct Rec3 { float4 XY, CC, ZO; };
    auto rec3_of = [&](uint32_t buf, int j) -> Rec3 {
      return Rec3{recP[buf][(2 * j + k) * 3], recP[buf][(2 * j + k) * 3 + 1], recP[buf][(2 * j + k) * 3 + 2]};
    };
    auto alpha2_rec = [&](const Rec3& R) -> f32x2 {
      const float4 XY = R.XY, CC = R.CC, ZO = R.ZO;
      const f32x2 dx = f32x2{XY.x, XY.y} - pq.pxf, dy = f32x2{XY.z, XY.w} - pq.pyf;
      const f32x2 power = gauss_power2v(f32x2{CC.x, CC.y}, f32x2{CC.z, CC.w}, f32x2{ZO.x, ZO.y}, dx, dy);
      const f32x2 G = exp2_<FAST>(power);
      const f32x2 oG = f32x2{ZO.z, ZO.w} * G;
      const float a0 = fminf(0.99f, oG.x), a1 = fminf(0.99f, oG.y);
      return f32x2{((power.x > 0.0f) || (a0 < 1.0f / 255.0f)) ? 0.f : a0, ((power.y > 0.0f) || (a1 < 1.0f / 255.0f)) ? 0.f : a1};
    };
    auto alpha2_of = [&](uint32_t buf, int j) -> f32x2 { return alpha2_rec(rec3_of(buf, j)); };
    float al[NSTEP];          // the alphas of my FIRST chunk stay in registers: its blend evaluates no exp.  (The second
                              // chunk of a round -- blocks with more than NW/2 live chunks -- is staged last, so its records
                              // are still in recP when it is blended: its alphas are evaluated again there.)
    uint32_t nmy[NS];         // survivors of my chunks
#pragma unroll
    for (int q = 0; q < NS; q++) {
      const uint32_t ci = (uint32_t)pr + (uint32_t)q * NP;
      const bool hasq = ci < nchunk;  // wave-uniform
      nmy[q] = hasq ? min((uint32_t)CHS, avail - ci * CHS) : 0u;
      if (q == 0) {
#pragma unroll
        for (int s = 0; s < NSTEP; s++) al[s] = 0.f;
      }
      float tp = 1.0f;
      if (hasq) {
        // lane e stages entry e of the chunk: its record (round 0: already in place, see the fill), its colour row
        float4 rq = make_float4(0, 0, 0, __uint_as_float(0u));
        float fv[NVF > 0 ? NVF : 1];
#pragma unroll
        for (int i = 0; i < (NVF > 0 ? NVF : 1); i++) fv[i] = 0.f;
        const bool valid = (uint32_t)lane < nmy[q];
        float* rp = reinterpret_cast<float*>(&recP[ci][0]) + (((lane >> 2) * 2 + (lane & 1)) * 12 + ((lane >> 1) & 1));
        auto colour_row = [&](uint32_t id) {
          const uint32_t gid = gauss_of(r, id);
          const uint32_t cid = r.colors_per_view ? id : gid;  // colour row: per view when it comes from SH
          rq = make_float4(r.colors[(size_t)cid * 3], r.colors[(size_t)cid * 3 + 1], r.colors[(size_t)cid * 3 + 2],
                           __uint_as_float(gid));
          if constexpr (NVF > 0) {
            if (use_feat) {
#pragma unroll
              for (int i = 0; i < NVF; i++) fv[i] = r.feats[(size_t)gid * F + i];
            }
          }
        };
        if (round == 0) {  // (workgroup-uniform)
          if (valid) {
            colour_row(sid[ci * CHS + (uint32_t)lane]);
          } else {
            // entries past the chunk's end: opacity 0 => alpha 0 => skipped (no bounds test per step), finite everywhere (both
            // waves of the pair write these same zeros, each before its own reads); their feature row is the chunk's first
            // Gaussian's -- a row the blend reads anyway, here with weight 0
            rp[0] = 0.f; rp[2] = 0.f; rp[4] = 0.f; rp[6] = 0.f; rp[8] = 0.f; rp[10] = 0.f;
            rq.w = __uint_as_float(gauss_of(r, sid[ci * CHS]));
          }
        } else {
          float4 g0 = make_float4(0, 0, 0, 0), g1 = make_float4(0, 0, -1.f, -1.f);
          if (valid) {
            const uint32_t id = my_surv[qhead + ci * CHS + (uint32_t)lane];
            g0 = r.rec[2 * (size_t)id]; g1 = r.rec[2 * (size_t)id + 1];
            colour_row(id);
          } else {
            rq.w = __uint_as_float(gauss_of(r, my_surv[qhead + ci * CHS]));
          }
          // (both waves of the pair write the same values to the chunk's buffer, each before its own reads; the other readers of
          //  this buffer -- the previous round's phase B -- are behind the list barrier)
          rp[0] = g0.x; rp[2] = g0.y; rp[4] = g0.z; rp[6] = g0.w; rp[8] = g1.x;
          rp[10] = valid ? g1.y : 0.f;
        }
        wave_lds_sync();
        if (q == 0) MGS_TRACE(3 + 8 * round);
        // straight-line: no per-step branch (an entry past the chunk's end has opacity 0 => alpha 0), so that the
        // compiler batches the LDS reads of several steps ahead of their use
        f32x2 tp2 = {1.0f, 1.0f};
        Rec3 Rn = rec3_of(ci, 0);  // the records of a double step are read one double step ahead of their use
#pragma unroll
        for (int j = 0; j < NSTEP / 2; j++) {
          const Rec3 Rc = Rn;
          if (j + 1 < NSTEP / 2) Rn = rec3_of(ci, j + 1);
          const f32x2 a2 = alpha2_rec(Rc);
          if (q == 0) { al[2 * j] = a2.x; al[2 * j + 1] = a2.y; }
          tp2 = tp2 * (1.0f - a2);
        }
        tp = tp2.x * tp2.y;
        // the colour rows are needed by the blend only: written now, their round trip lay behind the alphas
        rowq[w][q][lane] = rq;
        if constexpr (NVF > 0) {
#pragma unroll
          for (int i = 0; i < NVF; i++) rowf[(((size_t)w * NS + q) * CHS + lane) * NVF + i] = fv[i];
        }
        float t0 = tp, t1 = tp;
        swap32(t0, t1);       // t0: the even entries' product, t1: the odd entries', in both lanes of the pixel
        tp = t0 * t1;
      }
      if (k == 0) Tp[round & 1][ci][pixq] = tp;  // (1 for a chunk slot the round does not use)
    }
    MGS_TRACE(4 + 8 * round);
    __syncthreads();
    // every load of this round so far has been consumed; saying so (s_waitcnt vmcnt(0), free here) keeps the compiler from
    // protecting registers it believes still awaited further down -- behind this round's stores, whose drain that wait would
    // then include (one in-order counter for loads and stores)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    MGS_TRACE(5 + 8 * round);
    const uint32_t rb = rbase[round & 1];
    if (rb + (uint32_t)NW > pool) { overflow = true; break; }  // uniform: every thread reads the same word
    // ---- prefix in chunk order (identical arithmetic in every wave): transmittance of pixel pixx entering my chunks ----
    float Tin[NS], Tnext = Tround;
#pragma unroll
    for (int q = 0; q < NS; q++) Tin[q] = Tround;
#pragma unroll
    for (int c2 = 0; c2 < NW; c2++) {
      const float t2 = Tp[round & 1][c2][pixx];
#pragma unroll
      for (int q = 0; q < NS; q++) Tin[q] = (c2 < pr + q * NP) ? Tin[q] * t2 : Tin[q];
      Tnext *= t2;
    }
    // ---- phase B: blend my chunks ----
#pragma unroll
    for (int q = 0; q < NS; q++) {
      const uint32_t ci = (uint32_t)pr + (uint32_t)q * NP;
      const bool hasq = ci < nchunk;
      // the k = 1 lanes tracked the other half's pixel: they take my pixel's transmittance from their k = 0 partner
      const float Tpart = __uint_as_float(lane_xor<32>(__float_as_uint(Tin[q]), lane));
      float T = k ? Tpart : Tin[q];
      const bool live = hasq && pq.inside && !(T < 0.0001f);
      if (ballot(live) != 0) {
        const uint32_t n_my = nmy[q];
        float Cc[3] = {0.f, 0.f, 0.f};
        float Cv[NVF > 0 ? NVF : 1];
#pragma unroll
        for (int i = 0; i < (NVF > 0 ? NVF : 1); i++) Cv[i] = 0.f;
        f32x16 acc[NT > 0 ? NT : 1];
#pragma unroll
        for (int t = 0; t < (NT > 0 ? NT : 1); t++)
#pragma unroll
          for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
        // B operand of step s: lane (k, ch = pl) holds feat[entry 2s + k][32 t + ch]: one coalesced 128-B row per entry and
        // tile, a ring of BD steps in flight
        constexpr int BD = 4;
        float Bq[NT > 0 ? NT : 1][BD];
        // The ring holds the RAW loaded values (the select that discards rows past the chunk's end sits at the use, so that
        // the wait for a load happens BD steps after its issue, not at it); the row's Gaussian comes from LDS one step
        // before its load is issued.
        uint32_t gnext = __float_as_uint(rowq[w][q][k].w);  // Gaussian of entry 2s + k for the next load_B(s)
        // (buffer loads: one 32-bit offset per row -- the flat form spent four VALU instructions per step on a 64-bit address;
        //  an entry past the chunk's end names the chunk's FIRST Gaussian, see the staging: a row that is read anyway, weight 0)
        auto load_B = [&](int s) {
          const uint32_t gide = gnext;
          if (s + 1 < NSTEP) gnext = __float_as_uint(rowq[w][q][2 * (s + 1) + k].w);
#pragma unroll
          for (int t = 0; t < (NT > 0 ? NT : 1); t++) {
            const int ch = 32 * t + pl;
            if constexpr (MF)
              Bq[t][s % BD] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                  feat_rsrc, gide * (uint32_t)(F * 4) + (uint32_t)((ch < F ? ch : 0) * 4), 0, 0));
          }
        };
        if constexpr (MF) {
#pragma unroll
          for (int s = 0; s < BD; s++) load_B(s);
        }
        // The walking transmittance Tc is zeroed when the pixel terminates: from then on every test_T is 0 < 1e-4, i.e. every later
        // entry "terminates" again and is not blended -- no separate alive flag, and the chain per entry is sub, mul, compare,
        // select (rounds 3-4: seven dependent instructions).  T keeps the value the reference's loop breaks with.
        float Tc = live ? T : 0.f;
        uint32_t last = 0;
        float Tm = T;  // transmittance entering the chunk's second group of 32
        bool stop = false;  // wave-uniform: the chunk is exhausted or every pixel has terminated
        constexpr int GS = 4;  // steps per straight-line group (8 entries): one stop test and one branch per group
#pragma unroll
        for (int g = 0; g < NSTEP / GS; g++) {
          if (g * GS == NSTEP / 2) Tm = T;
          stop = stop || (uint32_t)(2 * g * GS) >= n_my || ballot(Tc != 0.f) == 0;
          __builtin_amdgcn_sched_barrier(0);
          if (!stop) {
            f32x2 a2q = {0.f, 0.f};  // the second chunk's alphas of a double step (evaluated at its even step)
#pragma unroll
            for (int u4 = 0; u4 < GS; u4++) {
              const int s = g * GS + u4;
              if (q != 0 && (u4 & 1) == 0) a2q = alpha2_of(ci, s >> 1);
              float a0 = (q == 0) ? al[s] : ((u4 & 1) ? a2q.y : a2q.x), a1 = a0;
              swap32(a0, a1);  // a0 = alpha of entry 2s, a1 = alpha of entry 2s + 1 for my pixel, in both of its lanes
              // two entries of the reference's per-pixel walk (forward.cu:357-380).  A live pixel always has T >= 1e-4 (it
              // entered so, and a blend only happens when the new T stays above), hence alpha == 0 (a skipped entry) can
              // never trip the stop test and needs no test of its own; T * (1 - a) is the reference's test_T bit for bit.
              float wq[2];
              const float aa[2] = {a0, a1};
#pragma unroll
              for (int u = 0; u < 2; u++) {
                const float test_T = Tc * (1.0f - aa[u]);
                const bool term = test_T < 0.0001f;
                wq[u] = (term ? 0.f : aa[u]) * Tc;   // alpha T if the entry is blended for this pixel, else 0
                Tc = term ? 0.f : test_T;            // (= T (1 - alpha), the same rounding)
                T = term ? T : test_T;
                last = wq[u] > 0.f ? (uint32_t)(2 * s + u) + 1u : last;
              }
              const float wk = k ? wq[1] : wq[0];  // the weight of MY entry
              const float4 rq = rowq[w][q][2 * s + k];
              Cc[0] += rq.x * wk; Cc[1] += rq.y * wk; Cc[2] += rq.z * wk;
              if constexpr (NVF > 0) {
#pragma unroll
                for (int i = 0; i < NVF; i++) Cv[i] += rowf[(((size_t)w * NS + q) * CHS + 2 * s + k) * NVF + i] * wk;
              }
              if constexpr (MF) {
#pragma unroll
                for (int t = 0; t < NT; t++) {
                  // (an entry past the chunk's end has weight 0 and a real row: no select.  F is the template's width, i.e.
                  //  features ARE rendered; only a last, partial tile of channels needs its padding lanes zeroed.)
                  float bv = Bq[t][s % BD];
                  if constexpr (F % 32 != 0) bv = (32 * t + pl) < F ? bv : 0.f;
                  acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wk, bv, acc[t], 0, 0, 0);
                }
                if (s + BD < NSTEP) load_B(s + BD);  // refill the ring slot just used
              }
            }
          }
        }
        if constexpr (MF) {
          // A walk that stops early leaves ring loads in flight whose registers are reused later: the compiler then waits for
          // them THERE -- after this chunk's stores, and (one in-order counter) for those stores as well: every block drained
          // its stores at the barrier before the final sum.  Touching the ring here puts that wait before the stores.
#pragma unroll
          for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < BD; i++) asm volatile("" ::"v"(Bq[t][i]));
        }
        if (last <= 32u) Tm = T;  // nothing of the second group was blended for this pixel (Tm is then never used)
        // the two parities' colour sums -> both lanes of the pixel
#pragma unroll
        for (int i = 0; i < 3; i++) Cc[i] += __uint_as_float(lane_xor<32>(__float_as_uint(Cc[i]), lane));
#pragma unroll
        for (int i = 0; i < NVF; i++) Cv[i] += __uint_as_float(lane_xor<32>(__float_as_uint(Cv[i]), lane));
        MGS_TRACE(6 + 8 * round);
        const size_t slot = (size_t)rb + (size_t)ci;
        float* pp = partial + slot * NCH * 64 + pixq;
        if (k == 0) {
          T_end[slot * 64 + pixq] = T;
          T_mid[slot * 64 + pixq] = Tm;
          last_pos[slot * 64 + pixq] = last;
#pragma unroll
          for (int i = 0; i < 3; i++) pp[i * 64] = Cc[i];
#pragma unroll
          for (int i = 0; i < NVF; i++) pp[(3 + i) * 64] = Cv[i];
        }
        if constexpr (MF) {
          // accumulators (col = channel pl, row = pixel (i & 3) + 8 (i >> 2) + 4 k of the half block) -> [channel][pixel];
          // lane (pixel pl, k) then stores the channels k F/2 .. k F/2 + F/2 - 1 of its pixel (32 consecutive floats per row)
          float* tb = trs + (size_t)w * TRR * TRS;
#pragma unroll
          for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < 16; i++)
              tb[(32 * t + pl) * TRS + (i & 3) + 8 * (i >> 2) + 4 * k] = acc[t][i];
          if (k == 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) tb[(NT * 32 + i) * TRS + pl] = Cc[i];
          }
          wave_lds_sync();
#pragma unroll
          for (int c = 0; c < F / 2; c++) {
            const int ch = k * (F / 2) + c;
            pp[(3 + ch) * 64] = tb[ch * TRS + pl];
          }
          wave_lds_sync();  // the next chunk's writes come after these reads
          my_lastcc = cbase + ci;
        }
        if (live) { my_vis = cbase + ci + 1; my_Tf = T; }
      }
    }
    MGS_TRACE(7 + 8 * round);
    Tround = Tnext;
    qhead += min(avail, nchunk * CHS);
    cbase += nchunk;
    if (ballot(insidex && !(Tround < 0.0001f)) == 0) break;  // all 64 pixels (each wave tracks them all): uniform
  }

  MGS_TRACE(TRACE_EVENTS - 3);
  // exactly one wave owns a pixel's last visited chunk: the 64-bit maximum carries its transmittance along
  if (my_vis > 0 && k == 0)
    atomicMax(&red_last[pixq], ((unsigned long long)my_vis << 32) | (unsigned long long)__float_as_uint(my_Tf));
  if constexpr (MF) { if (lane == 0) lastcc[w] = my_lastcc; }
  __syncthreads();  // also: every wave's partial sums are written (workgroup scope)
  const unsigned long long rl = red_last[lane];   // from here on: lane = pixel of the whole block
  const uint32_t vis = (uint32_t)(rl >> 32);
  const float Tf = __uint_as_float((uint32_t)rl);
  // ---- image = sum of the visited chunks' partial colours, in chunk order; wave w owns channels w, w + NW, ... ----
  float img[NOWN];
#pragma unroll
  for (int kk = 0; kk < NOWN; kk++) img[kk] = 0.f;
  const uint32_t vmax = wave_umax(vis);
  constexpr int NFLY = 8;  // records whose loads are in flight together (a block visits ~7 chunks at BASELINE configs[2])
  static_assert(NW % NFLY == 0, "a batch of the final sum lies inside one round");
  for (uint32_t c0 = 0; c0 < vmax; c0 += NFLY) {  // the sum stays in chunk order
    float v[NFLY][NOWN];
    // the round's first record, ONE wave-uniform read per batch and decided by a branch: a per-chunk select between the
    // LDS copy and the one in memory compiles to a flat load with a full wait per chunk, which serialises the batch
    const uint32_t rr = c0 / NW;
    uint32_t rb0 = rb_hist[rr < RBH ? rr : 0u];                              // (unconditional LDS read ...
    if (__builtin_expect(rr >= RBH, 0))                                      //  ... and a rarely taken branch; the atomic
      rb0 = __hip_atomic_load(&my_rounds[4 * (size_t)rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // load is not merged)
    rb0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rb0);
    // A chunk that was the LAST one of the wave that blended it (at BASELINE configs[2]: every chunk of 85 % of the blocks)
    // is still in that wave's hand-over buffer: it is read from LDS, and only the others come from memory -- where the
    // loads queue behind this block's own stores of a moment ago.  Same values, same order either way.
    bool inl[NFLY];
    unsigned long long needg = 0ull;
#pragma unroll
    for (int u = 0; u < NFLY; u++) {
      const uint32_t cc = c0 + u;
      inl[u] = false;
      if constexpr (MF) {
        const uint32_t ws = 2u * ((cc % NW) % NP) + (uint32_t)(lane >> 5);  // the wave that blended my pixel's half of chunk cc
        inl[u] = lastcc[ws] == cc;
      }
      needg |= ballot(cc < vis && !inl[u]);
#pragma unroll
      for (int kk = 0; kk < NOWN; kk++) v[u][kk] = 0.f;
    }
    if (needg != 0ull) {  // (wave-uniform: the wait for these loads, and with it for the stores ahead of them, is in here)
#pragma unroll
      for (int u = 0; u < NFLY; u++) {
        const uint32_t cc = c0 + u;
        const bool gl = cc < vis && !inl[u];
        const size_t slot = gl ? (size_t)rb0 + (cc % NW) : 0;
        const float* pp = partial + slot * NCH * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < NOWN; kk++) {
          const int ch = w + kk * NW;
          if (gl && ch < NCH && (ch < 3 || use_feat)) v[u][kk] = pp[ch * 64];
        }
      }
#pragma unroll
      for (int u = 0; u < NFLY; u++)
#pragma unroll
        for (int kk = 0; kk < NOWN; kk++) asm volatile("" : "+v"(v[u][kk]));  // (the wait stays inside the branch)
    }
    if constexpr (MF) {
#pragma unroll
      for (int u = 0; u < NFLY; u++) {
        const uint32_t cc = c0 + u;
        const uint32_t ws = 2u * ((cc % NW) % NP) + (uint32_t)(lane >> 5);
        const float* tb = trs + (size_t)ws * TRR * TRS + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < NOWN; kk++) {
          const int ch = w + kk * NW;
          const int row = ch < 3 ? NT * 32 + ch : ch - 3;  // hand-over rows: features first, then r, g, b
          if (cc < vis && inl[u] && ch < NCH && (ch < 3 || use_feat)) v[u][kk] = tb[row * TRS];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NFLY; u++)
#pragma unroll
      for (int kk = 0; kk < NOWN; kk++) img[kk] += v[u][kk];
  }
  MGS_TRACE(TRACE_EVENTS - 2);
  // This block has taken its last chunk records: count it (after the sum: the wait for the returning ticket would
  // otherwise hold wave 0, and with it everyone, at the barrier above); whoever draws the last ticket reports the pool
  // usage to the host.
  uint32_t ticket = 0;
  if (tid == 0) {
    // (device-scope atomics served by the L2; the OR's returned value feeds the ticket, so it has been performed when the
    //  ticket is counted -- no fence: a fence here costs every block ~3 us)
    const uint32_t dep = overflow ? (atomicOr(&flags[FLAG_PREFILTERED], 0x100u) & 0u) : 0u;
    ticket = atomicAdd(&flags[FLAG_BLOCKS_DONE], 1u + dep) + 1u;
  }
  const size_t HW = (size_t)r.Hv * r.W;  // one image plane of one view
  if (p.inside) {
#pragma unroll
    for (int kk = 0; kk < NOWN; kk++) {
      const int ch = w + kk * NW;
      if (ch < 3) out_color[((size_t)p.v * 3 + ch) * HW + p.pixl] = img[kk] + Tf * r.bg[ch];
      else if (ch < NCH && use_feat) out_feat[((size_t)p.v * F + (ch - 3)) * HW + p.pixl] = img[kk];
    }
  }
  if (w == 0) {
    last_chunk[((size_t)tile * 4 + sub) * 64 + lane] = vis;
    if (p.inside) final_T[p.pixa] = Tf;
    if (lane == 0) {
      nsurv[(size_t)tile * 4 + sub] = make_uint2(qtail, round > 0 || vis > 0 ? rb_hist[0] : 0u);  // + round 0's first record
      // the workgroup that drew the last ticket reports {tag, overflow, chunk records used} to the host (mapped pinned memory)
      if (ticket == (uint32_t)nblocks && host_status) {
        const uint32_t used = dyn_base + __hip_atomic_load(&flags[FLAG_CHUNKS_USED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  hipLaunchKernelGGL((coop_fwd_pairs_kernel<F, FAST, EXACT, NW, CHUNK, TWO>), dim3(grid), dim3(NW * 64), 0, s, r,      \
                     im.ranges, b.point_list, cv.T_end, cv.T_mid, cv.last_pos, cv.partial, cv.surv,                   \
                     cv.surv_stride, cv.nsurv, im.final_T, cv.last_chunk, oc, of, cv.round_base, cv.pool, im.flags,   \
                     4 * T, st.host, st.tag, (im.direct_keys && im.ready) ? im.ready + 1 : nullptr)
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
