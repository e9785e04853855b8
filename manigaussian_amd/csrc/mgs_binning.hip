// mgs_binning.hip -- tile binning (K3-K6): from per-Gaussian tile rects to per-tile, depth-ordered instance lists.
//
// Result contract (per tile: instances ordered by view-depth bits, ties by Gaussian index -- what the reference's stable
// radix sort of (tile<<32 | depth) keys yields, RAST/cuda_rasterizer/rasterizer_impl.cu:70-138,280-320), reached another
// way: tile histogram (built by the forward preprocess) -> scatter of (depth|id) keys into per-tile slices -> bitonic sort
// of <= SEG-entry segments in LDS -> rank merge across a tile's segments + emission.  Keys (depth bits, id) are unique, so
// the order is deterministic and equals the stable sort's order.  No library calls.
//
// The scatter has two forms (the sort and the merge are shared):
//   LDS tables (bin_mode 1 and T <= LDS_TILES tiles)  the preprocess reserved every workgroup's part of every slice; the
//                         scatter keeps slice starts, reservations and cursors in LDS.  3 launches.
//   tables in memory (any tile count; bin_mode 0)     the preprocess counted instances per tile with plain atomics; one
//                         workgroup scans the histogram into ranges + the segment table, the scatter takes slots with one
//                         atomic per instance on per-tile cursors.  4 launches.
#include "mgs_common.h"
#include "mgs_device.h"

namespace mgs {

// ================================ scatter -> segment sort -> rank merge ===================================
//
// preprocess (mgs_preprocess.hip)  tile_hist[t] = instances of tile t; blk_base[b][t] = offset reserved by
//                                  preprocess workgroup b inside tile t's slice; flags[1] = R
// bin_scatter_kernel               key (depth bits << 32 | id) of every instance -> its tile slice (unordered)
// bin_segsort_kernel<SEG>          a tile's slice of L keys = ceil(L/SEG) equal segments; one workgroup sorts one
//                                  segment in LDS (bitonic network on 64-bit keys)
// bin_merge_emit_kernel<SEG>       final position of a key = its index + its lower-bound rank in the tile's other
//                                  segments (keys are unique); writes point_list and the packed instance records

__device__ __forceinline__ uint32_t div_up_u(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

// Exclusive scan of v[0..n) (LDS) in place; *total (LDS) receives the sum.  All threads of the block call it.
// tmp: LDS scratch of >= 128 entries.  blockDim.x <= 1024 (16 waves).
__device__ void block_exclusive_scan(uint32_t* v, int n, uint32_t* tmp, uint32_t* total) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = (nt + 63) >> 6;
  if (n <= 64) {  // (workgroup-uniform) one wave, one DPP scan, one barrier: the 64 tiles of a 128 x 128 image
    if (wv == 0) {
      const uint32_t x = lane < n ? v[lane] : 0u;
      const uint32_t incl = wave_incl_scan_add_u32(x);
      if (lane < n) v[lane] = incl - x;
      if (lane == 63) *total = incl;
    }
    __syncthreads();
    return;
  }
  const int per = (n + nt - 1) / nt;
  const int b = tid * per, e = min(n, b + per);
  uint32_t sum = 0;
  for (int i = b; i < e; i++) sum += v[i];
  // wave-level inclusive scan of the per-thread sums, wave totals through LDS, one more wave-level scan: three barriers
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
    incl += lane >= d ? up : 0u;
  }
  if (lane == 63) tmp[wv] = incl;
  __syncthreads();
  if (wv == 0) {
    const uint32_t w = lane < nw ? tmp[lane] : 0u;
    uint32_t wi = w;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)wi, d, 64);
      wi += lane >= d ? up : 0u;
    }
    if (lane < nw) tmp[64 + lane] = wi - w;  // exclusive prefix of the wave totals
    if (lane == nw - 1) *total = wi;
  }
  __syncthreads();
  uint32_t run = tmp[64 + wv] + incl - sum;  // exclusive prefix of this thread's slice
  for (int i = b; i < e; i++) { const uint32_t x = v[i]; v[i] = run; run += x; }
  __syncthreads();
}

// A tile's slice of L keys is cut into ns = ceil(L/seg) equal segments of seglen = ceil(L/ns) keys.
// seg_desc[s] = {first key of segment s (absolute), its key count, tile slice start, tile slice length}.

// Workgroups [0, nblk) scatter the keys of blockDim.x = pre_block() Gaussians each (the partition the preprocess used);
// workgroup nblk publishes ranges and the segment table for the next two kernels.
// T = tiles = sort slices.
__global__ void __launch_bounds__(PRE_BLOCK) bin_scatter_kernel(int Pg, int T, int tiles_x, int nblk,
                                                                uint32_t seg,
                                                                uint32_t capacity, const uint32_t* __restrict__ flags,
                                                                uint64_t* host_status, uint32_t status_tag,
                                                                const uint32_t* __restrict__ ref_count,
                                                                const uint2* __restrict__ rect,
                                                                const float* __restrict__ depths,
                                                                const uint32_t* __restrict__ tile_hist,
                                                                const uint32_t* __restrict__ blk_base,
                                                                uint64_t* __restrict__ keys_unsorted,
                                                                uint2* __restrict__ ranges,
                                                                uint32_t* __restrict__ seg_base,
                                                                uint4* __restrict__ seg_desc,
                                                                unsigned long long* ready, unsigned long long nonce) {
  extern __shared__ uint32_t lds_u[];
  const int S = T;
  uint32_t* s_start = lds_u;          // [S] exclusive scan of the histogram
  uint32_t* s_cnt = lds_u + S;        // [S] write cursor of this workgroup inside its reservation
  uint32_t* s_base = lds_u + 2 * S;   // [S] this workgroup's reserved offset inside the slice
  __shared__ uint32_t tmp[PRE_BLOCK];
  __shared__ uint32_t total;
  const int tid = threadIdx.x;
  const bool tables = (int)blockIdx.x == nblk;
  // Everything this thread reads from memory is requested NOW, in one batch: the kernel is a chain of round trips otherwise
  // (instance count -> histogram -> reservation row -> rect -> depth: five, each ~0.7 us of a 6.8 us kernel).
  // Same (view, Gaussian) partition as the forward preprocess: workgroups never straddle views.
  const int bpv = (Pg + (int)blockDim.x - 1) / (int)blockDim.x;
  const int v = (int)blockIdx.x / bpv;
  const int gi = ((int)blockIdx.x - v * bpv) * (int)blockDim.x + tid;
  const bool work = !tables && gi < Pg;
  const int idx = work ? v * Pg + gi : 0;  // (virtual) instance owner
  const uint2 r_pre = rect[idx];
  const float depth_pre = depths[idx];
  const uint32_t* __restrict__ row = blk_base + (size_t)blockIdx.x * T;
  const uint32_t hist_pre = tile_hist[tid < T ? tid : 0];
  const uint32_t row_pre = tables ? 0u : row[tid < T ? tid : 0];
  // ready[1] == this forward's nonce: a preprocess workgroup gave up waiting for the zeroed tables (see there).  The
  // histogram is then incomplete: nothing is binned, and flag bit 1 (MGS_FLAG_HANDSHAKE) tells the host why.
  const bool hs_failed = ready != nullptr && nonce != 0ull && ready[1] == nonce;
  const uint32_t R = hs_failed ? 0xffffffffu : flags[FLAG_NUM_RENDERED];
  if (tables && tid == 0 && ready) *ready = 0ull;  // the preprocess's hand-shake word: "not ready" for the next launch on this buffer
  // (ready[1], the failure mark, is read by EVERY workgroup of this launch: the segment sort, next in the chain, clears it)
  if (tables && tid == 0 && host_status) {  // report to the host (mapped pinned memory): word 2 = {tag, the reference's
    // 3-sigma-rect instance count} first, then word 0 = {tag, flags, R} with release order -- a host that sees word 0 sees word 2
    __hip_atomic_store(host_status + 2, ((uint64_t)(status_tag & 0xffffu) << 48) | (uint64_t)(hs_failed ? 0u : *ref_count),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_status, ((uint64_t)(status_tag & 0xffffu) << 48) |
                       ((uint64_t)((flags[FLAG_PREFILTERED] & 0xfffdu) | (hs_failed ? 2u : 0u)) << 32) | (hs_failed ? 0u : R),
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (R > capacity) {  // the workspace cannot hold the lists (or the hand-shake failed): publish "nothing binned", the caller retries
    if (tables) {
      for (int t = tid; t < T; t += blockDim.x) ranges[t] = make_uint2(0u, 0u);
      for (int t = tid; t < S; t += blockDim.x) seg_base[t] = 0u;
      if (tid == 0) seg_base[S] = 0u;
    }
    return;
  }
  for (int t = tid; t < S; t += blockDim.x) {
    const bool first = t == tid;  // (the first blockDim.x tiles came with the batch above)
    s_start[t] = first ? hist_pre : tile_hist[t];
    s_cnt[t] = 0;
    s_base[t] = first ? row_pre : (tables ? 0u : row[t]);  // only entries of slices this workgroup contributed to are meaningful
  }
  __syncthreads();
  block_exclusive_scan(s_start, S, tmp, &total);
  if (tables) {
    for (int t = tid; t < T; t += blockDim.x) ranges[t] = make_uint2(s_start[t], s_start[t] + tile_hist[t]);
    if (seg == 0u) return;  // the bucket rank reads the ranges only: no segment table
    for (int t = tid; t < S; t += blockDim.x) s_base[t] = div_up_u(tile_hist[t], seg);
    __syncthreads();
    block_exclusive_scan(s_base, S, tmp, &total);
    for (int t = tid; t < S; t += blockDim.x) {
      const uint32_t L = tile_hist[t], ns = div_up_u(L, seg), sb = s_base[t];
      const uint32_t seglen = ns ? div_up_u(L, ns) : 0u;
      seg_base[t] = sb;
      for (uint32_t k = 0; k < ns; k++)
        seg_desc[sb + k] = make_uint4(s_start[t] + k * seglen, min(seglen, L - k * seglen), s_start[t], L);
    }
    if (tid == 0) seg_base[S] = total;
    return;
  }
  if (!work) return;
  const uint2 r = r_pre;
  const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16);
  const int y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
  if (x1 <= x0 || y1 <= y0) return;
  const float depth = depth_pre;
  const uint64_t key = ((uint64_t)__float_as_uint(depth) << 32) | (uint32_t)idx;
  for (int y = y0; y < y1; y++)
    for (int x = x0; x < x1; x++) {
      const int t = y * tiles_x + x;
      const uint32_t slot = s_start[t] + s_base[t] + atomicAdd(&s_cnt[t], 1u);
      keys_unsorted[slot] = key;
    }
}

// ---- tables in memory: any tile count ----------------------------------------------------------------------
// One workgroup: exclusive scan of the tile histogram, LDS_TILES tiles at a time -> ranges, seg_base, seg_desc; reports
// {tag, flags, R} to the host; zeroes the scatter's cursors.  R > capacity: "nothing binned" (the histogram stays: the
// caller's retry with a larger workspace starts here again).
__global__ void __launch_bounds__(1024) bin_tables_kernel(int T, uint32_t seg, uint32_t capacity, const uint32_t* __restrict__ flags,
                                                          uint64_t* host_status, uint32_t status_tag,
                                                          const uint32_t* __restrict__ ref_count,
                                                          const uint32_t* __restrict__ tile_hist,
                                                          uint32_t* __restrict__ cursor, uint2* __restrict__ ranges,
                                                          uint32_t* __restrict__ seg_base, uint4* __restrict__ seg_desc) {
  __shared__ uint32_t s_cnt[LDS_TILES], s_start[LDS_TILES], s_seg[LDS_TILES];
  __shared__ uint32_t tmp[1024];
  __shared__ uint32_t total;
  const int tid = threadIdx.x;
  const uint32_t R = flags[FLAG_NUM_RENDERED];
  if (tid == 0 && host_status) {  // (word 2 = the reference's count first, then word 0 with release order: see bin_scatter_kernel)
    __hip_atomic_store(host_status + 2, ((uint64_t)(status_tag & 0xffffu) << 48) | (uint64_t)*ref_count, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_status, ((uint64_t)(status_tag & 0xffffu) << 48) | ((uint64_t)(flags[FLAG_PREFILTERED] & 1u) << 32) | R,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const bool over = R > capacity;
  uint32_t key_carry = 0, seg_carry = 0;
  for (int t0 = 0; t0 < T; t0 += LDS_TILES) {
    const int n = min(LDS_TILES, T - t0);
    for (int i = tid; i < n; i += blockDim.x) {
      const uint32_t c = over ? 0u : tile_hist[t0 + i];
      s_cnt[i] = c; s_start[i] = c; s_seg[i] = div_up_u(c, seg);
      cursor[t0 + i] = 0u;
    }
    __syncthreads();
    block_exclusive_scan(s_start, n, tmp, &total);
    const uint32_t keys_here = total;
    __syncthreads();
    block_exclusive_scan(s_seg, n, tmp, &total);
    const uint32_t segs_here = total;
    for (int i = tid; i < n; i += blockDim.x) {
      const uint32_t L = s_cnt[i], st = key_carry + s_start[i], sb = seg_carry + s_seg[i];
      const uint32_t ns = div_up_u(L, seg), seglen = ns ? div_up_u(L, ns) : 0u;
      ranges[t0 + i] = make_uint2(st, st + L);
      seg_base[t0 + i] = sb;
      for (uint32_t k = 0; k < ns; k++) seg_desc[sb + k] = make_uint4(st + k * seglen, min(seglen, L - k * seglen), st, L);
    }
    key_carry += keys_here; seg_carry += segs_here;
    __syncthreads();
  }
  if (tid == 0) seg_base[T] = seg_carry;
}

// One thread per (virtual) Gaussian: slot = slice start + one atomic on the tile's cursor (order inside a slice is
// arbitrary: the sort that follows makes it deterministic).
__global__ void __launch_bounds__(256) bin_scatter_global_kernel(int n, int tiles_x, uint32_t capacity,
                                                                 const uint32_t* __restrict__ flags,
                                                                 const uint2* __restrict__ rect,
                                                                 const float* __restrict__ depths,
                                                                 const uint2* __restrict__ ranges,
                                                                 uint32_t* __restrict__ cursor,
                                                                 uint64_t* __restrict__ keys_unsorted) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint2 r = rect[idx];
  const float depth = depths[idx];
  if (flags[FLAG_NUM_RENDERED] > capacity) return;
  const int x0 = (int)(r.x & 0xffffu), x1 = (int)(r.x >> 16);
  const int y0 = (int)(r.y & 0xffffu), y1 = (int)(r.y >> 16);
  if (x1 <= x0 || y1 <= y0) return;
  const uint64_t key = ((uint64_t)__float_as_uint(depth) << 32) | (uint32_t)idx;
  for (int y = y0; y < y1; y++)
    for (int x = x0; x < x1; x++) {
      const int t = y * tiles_x + x;
      keys_unsorted[ranges[t].x + atomicAdd(&cursor[t], 1u)] = key;
    }
}

// ---- bitonic sort of one segment: two keys per lane in registers ---------------------------------------
// Element e of the segment lives in thread e/2, slot e%2.  Compare-exchange distance j: 1 = inside the thread,
// 2..64 = lane distance j/2 inside the wave (DPP / permlane swaps, no LDS), >= 128 = across waves through LDS.
template <int D>
__device__ __forceinline__ void ce_lane(uint64_t& k0, uint64_t& k1, bool up, int lane) {
  const bool keep_min = ((lane & D) == 0) == up;
  const uint64_t p0 = lane_xor64<D>(k0, lane), p1 = lane_xor64<D>(k1, lane);
  k0 = ((k0 < p0) == keep_min) ? k0 : p0;
  k1 = ((k1 < p1) == keep_min) ? k1 : p1;
}

template <int SEGN>
__global__ void __launch_bounds__(SEGN / 2) bin_segsort_kernel(const uint32_t* __restrict__ n_seg,
                                                                const uint4* __restrict__ seg_desc,
                                                                const uint64_t* __restrict__ keys_unsorted,
                                                                uint64_t* __restrict__ keys,
                                                                uint32_t* __restrict__ point_list,
                                                                unsigned long long* hs_fail_mark) {
  __shared__ uint64_t sk[SEGN];
  const uint32_t tid = threadIdx.x;
  const int lane = (int)(tid & 63u);
  const uint4 d = seg_desc[blockIdx.x];  // surplus workgroups read an unused (in-bounds) entry
  // ImgView::ready[1], the preprocess's "a workgroup gave up waiting" mark: every workgroup of the bin scatter has read it
  // (kernel boundary); cleared here so that a replayed HIP graph -- same buffer, same nonce -- does not see a stale failure
  if (blockIdx.x == 0 && tid == 0 && hs_fail_mark) *hs_fail_mark = 0ull;
  if (blockIdx.x >= *n_seg) return;
  const uint32_t cnt = d.y;
  const size_t base = d.x;
  uint32_t n = 128;
  while (n < cnt) n <<= 1;
  const uint32_t e0 = 2u * tid;
  uint64_t k0 = e0 < cnt ? keys_unsorted[base + e0] : ~0ull;
  uint64_t k1 = e0 + 1u < cnt ? keys_unsorted[base + e0 + 1u] : ~0ull;
  for (uint32_t k = 2; k <= n; k <<= 1) {
    const bool up = (e0 & k) == 0;  // same for both slots (k >= 2)
    uint32_t j = k >> 1;
    if (j >= 128) {
      sk[e0] = k0; sk[e0 + 1] = k1;
      for (; j >= 128; j >>= 1) {
        __syncthreads();
        if (tid < n / 2) {
          const uint32_t a = ((tid & ~(j - 1u)) << 1) | (tid & (j - 1u));
          const uint32_t b = a | j;
          const bool upa = (a & k) == 0;
          const uint64_t x = sk[a], y = sk[b];
          if ((x > y) == upa) { sk[a] = y; sk[b] = x; }
        }
      }
      __syncthreads();
      k0 = sk[e0]; k1 = sk[e0 + 1];
    }
    switch (j) {
      case 64: ce_lane<32>(k0, k1, up, lane); [[fallthrough]];
      case 32: ce_lane<16>(k0, k1, up, lane); [[fallthrough]];
      case 16: ce_lane<8>(k0, k1, up, lane); [[fallthrough]];
      case 8: ce_lane<4>(k0, k1, up, lane); [[fallthrough]];
      case 4: ce_lane<2>(k0, k1, up, lane); [[fallthrough]];
      case 2: ce_lane<1>(k0, k1, up, lane); [[fallthrough]];
      default: break;
    }
    {  // j == 1
      const uint64_t lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0;
      k0 = up ? lo : hi;
      k1 = up ? hi : lo;
    }
  }
  if (cnt == d.w) {  // the slice is this one segment: sorted ids go straight out, the merge kernel skips it
    if (e0 < cnt) point_list[base + e0] = (uint32_t)k0;
    if (e0 + 1u < cnt) point_list[base + e0 + 1u] = (uint32_t)k1;
    return;
  }
  if (e0 < cnt) keys[base + e0] = k0;
  if (e0 + 1u < cnt) keys[base + e0 + 1u] = k1;
}

// ---- bitonic sort of one 4096-key segment: FOUR keys per lane (MgsOptions.seg = 4096, an option) -------------
// Element e lives in thread e / 4, slot e % 4.  Compare-exchange distance j: 1, 2 = inside the thread; 4 .. 128 = lane
// distance j / 4 inside the wave; >= 256 = through LDS, TWO stages per round trip: a thread picks up the elements
// {x, x+J, x+2J, x+3J} (x without the bits J and 2J), which hold the pairs of stage 2J and of stage J.
// Round 4 measured this family against the two-keys-per-lane kernel above (profiles/r04_exp_segsort.log): four keys per lane
// on 512 threads for 2048-key segments: 15.4 us against 13.2 at BASELINE configs[2], 77 us against 65 at the configs[4]
// shape (the network is latency-bound: more live waves hide more of it); two keys per lane with paired LDS round trips:
// 14.3 / 74 us.  Neither replaced the kernel above; 4096-key segments halve the rank merge's work on long lists (58 -> 38 us
// at the configs[4] shape) for a slower sort (65 -> 77 us) -- 115 us against 123 in all, kept as an option.
// The index arithmetic was checked against a CPU emulation for every segment-length class.
__device__ __forceinline__ void ce_regs(uint64_t& a, uint64_t& b, bool up) {
  const bool lt = a < b;
  const uint64_t lo = lt ? a : b, hi = lt ? b : a;
  a = up ? lo : hi;
  b = up ? hi : lo;
}
template <int D, int KPT>
__device__ __forceinline__ void ce_lanes(uint64_t (&k)[KPT], bool up, int lane) {
  const bool keep_min = ((lane & D) == 0) == up;
#pragma unroll
  for (int s = 0; s < KPT; s++) {
    const uint64_t p = lane_xor64<D>(k[s], lane);
    k[s] = ((k[s] < p) == keep_min) ? k[s] : p;
  }
}

template <int SEGN, int KPT>
__global__ void __launch_bounds__(SEGN / KPT) bin_segsort4_kernel(const uint32_t* __restrict__ n_seg,
                                                                  const uint4* __restrict__ seg_desc,
                                                                  const uint64_t* __restrict__ keys_unsorted,
                                                                  uint64_t* __restrict__ keys,
                                                                  uint32_t* __restrict__ point_list,
                                                                  unsigned long long* hs_fail_mark) {
  static_assert(KPT == 2 || KPT == 4, "two or four keys per lane");
  constexpr uint32_t NT = SEGN / KPT;
  constexpr uint32_t LDS_MIN = 64u * KPT;  // smallest compare-exchange distance that crosses waves
  __shared__ uint64_t sk[SEGN];
  const uint32_t tid = threadIdx.x;
  const int lane = (int)(tid & 63u);
  const uint4 d = seg_desc[blockIdx.x];  // surplus workgroups read an unused (in-bounds) entry
  const uint32_t nseg = *n_seg;
  if (blockIdx.x == 0 && tid == 0 && hs_fail_mark) *hs_fail_mark = 0ull;  // (see bin_segsort_kernel)
  if (blockIdx.x >= nseg) return;
  const uint32_t cnt = d.y;
  const size_t base = d.x;
  uint32_t n = LDS_MIN;
  while (n < cnt) n <<= 1;
  const uint32_t e0 = (uint32_t)KPT * tid;
  const bool mine = e0 < n;  // threads past the padded length hold padding only
  uint64_t k[KPT];
#pragma unroll
  for (int s = 0; s < KPT; s++) k[s] = e0 + (uint32_t)s < cnt ? keys_unsorted[base + e0 + (uint32_t)s] : ~0ull;
  for (uint32_t kk = 2; kk <= n; kk <<= 1) {
    uint32_t j = kk >> 1;
    if (j >= LDS_MIN) {
      if (mine) {
#pragma unroll
        for (int s = 0; s < KPT; s++) sk[e0 + (uint32_t)s] = k[s];
      }
      while (j >= LDS_MIN) {
        const bool two = (j >> 1) >= LDS_MIN;
        const uint32_t J = two ? (j >> 1) : j;
        __syncthreads();
        if (two) {  // stages 2J and J (kk = 4J: one direction for the four elements of a group)
          for (uint32_t g = tid; g < (n >> 2); g += NT) {
            const uint32_t x = ((g & ~(J - 1u)) << 2) | (g & (J - 1u));
            uint64_t a0 = sk[x], a1 = sk[x + J], a2 = sk[x + 2u * J], a3 = sk[x + 3u * J];
            const bool up = (x & kk) == 0u;
            ce_regs(a0, a2, up); ce_regs(a1, a3, up);
            ce_regs(a0, a1, up); ce_regs(a2, a3, up);
            sk[x] = a0; sk[x + J] = a1; sk[x + 2u * J] = a2; sk[x + 3u * J] = a3;
          }
        } else {    // stage J alone
          for (uint32_t c = tid; c < (n >> 1); c += NT) {
            const uint32_t x = ((c & ~(J - 1u)) << 1) | (c & (J - 1u));
            uint64_t a0 = sk[x], a1 = sk[x + J];
            ce_regs(a0, a1, (x & kk) == 0u);
            sk[x] = a0; sk[x + J] = a1;
          }
        }
        j = J >> 1;
      }
      __syncthreads();
      if (mine) {
#pragma unroll
        for (int s = 0; s < KPT; s++) k[s] = sk[e0 + (uint32_t)s];
      }
    }
    if (kk >= 2u * KPT) {  // lane distances j / KPT = 32 .. 1
      const bool up = (e0 & kk) == 0u;  // one direction for the thread's slots
      switch (j / (uint32_t)KPT) {
        case 32: ce_lanes<32, KPT>(k, up, lane); [[fallthrough]];
        case 16: ce_lanes<16, KPT>(k, up, lane); [[fallthrough]];
        case 8: ce_lanes<8, KPT>(k, up, lane); [[fallthrough]];
        case 4: ce_lanes<4, KPT>(k, up, lane); [[fallthrough]];
        case 2: ce_lanes<2, KPT>(k, up, lane); [[fallthrough]];
        case 1: ce_lanes<1, KPT>(k, up, lane); [[fallthrough]];
        default: break;
      }
    }
    // distances below KPT: inside the thread
    if constexpr (KPT == 4) {
      if (kk >= 4u) {
        const bool up = (e0 & kk) == 0u;
        ce_regs(k[0], k[2], up); ce_regs(k[1], k[3], up);
        ce_regs(k[0], k[1], up); ce_regs(k[2], k[3], up);
      } else {  // kk == 2: elements 4t, 4t+1 ascending, 4t+2, 4t+3 descending
        ce_regs(k[0], k[1], true); ce_regs(k[2], k[3], false);
      }
    } else {
      ce_regs(k[0], k[1], (e0 & kk) == 0u);
    }
  }
  if (cnt == d.w) {  // the slice is this one segment: sorted ids go straight out, the merge kernel skips it
#pragma unroll
    for (int s = 0; s < KPT; s++)
      if (e0 + (uint32_t)s < cnt) point_list[base + e0 + (uint32_t)s] = (uint32_t)k[s];
    return;
  }
#pragma unroll
  for (int s = 0; s < KPT; s++)
    if (e0 + (uint32_t)s < cnt) keys[base + e0 + (uint32_t)s] = k[s];
}

// lower bounds of two keys in the sorted LDS array a[0..len), len <= SEGN and WAVE-UNIFORM (every lane ranks in the same
// segment).  Bounded branch-free search: the first probe, at the largest power of two P <= len, picks the window a[0..P) or
// a[len-P..len) (everything before it is then known to be smaller); log2 P halvings and one last probe finish inside the
// window -- no probe ever needs a bounds test.  (The previous version clamped every probe with min(q, len) and combined
// `q <= len && v < key` per step: two scalar mask operations per step and key on the CU's one scalar unit, which is what
// bound this kernel at long lists: 56.9 M SALU against 36.7 M VALU instructions per launch at the configs[4] shape,
// profiles/r02_sq_counters_c5shape.json.)
template <int SEGN>
__device__ __forceinline__ void lds_lower_bound2(const uint64_t* a, uint32_t len, uint64_t k0, uint64_t k1, uint32_t& q0,
                                                 uint32_t& q1) {
  q0 = 0u; q1 = 0u;
  if (len == 0u) return;
  const uint32_t P = 1u << (31 - __builtin_clz(len));
  {
    const uint64_t v = a[P - 1u];
    q0 = v < k0 ? len - P : 0u;
    q1 = v < k1 ? len - P : 0u;
  }
#pragma unroll
  for (uint32_t h = SEGN / 2; h >= 1u; h >>= 1) {
    if (h < P) {  // (wave-uniform)
      const uint64_t v0 = a[q0 + h - 1u], v1 = a[q1 + h - 1u];
      q0 += v0 < k0 ? h : 0u;
      q1 += v1 < k1 ? h : 0u;
    }
  }
  {
    const uint64_t v0 = a[q0], v1 = a[q1];
    q0 += v0 < k0 ? 1u : 0u;
    q1 += v1 < k1 ? 1u : 0u;
  }
}

// K6': final position of a key = its index + its lower-bound rank in the tile's other segments (keys are unique),
// staged through LDS in groups of whole segments (<= CAP keys); then emission of the sorted ids.  One workgroup per
// segment, KPT keys per thread.  (A one-workgroup-per-tile variant that emits in output order with fully coalesced stores
// was measured at 45 us against 19 us for this one at C3: 64 tiles cannot keep 256 CUs busy.)
template <int SEGN, int KPT>  // KPT: keys ranked per thread
__global__ void __launch_bounds__(SEGN / KPT) bin_merge_emit_kernel(const uint32_t* __restrict__ n_seg,
                                                                     const uint4* __restrict__ seg_desc,
                                                                     const uint64_t* __restrict__ keys,
                                                                     uint32_t* __restrict__ point_list) {
  constexpr uint32_t NT = SEGN / KPT;
  constexpr uint32_t CAP = 8192;  // keys staged per group (64 KB)
  __shared__ uint64_t sk[CAP];
  const uint32_t tid = threadIdx.x;
  // Workgroup -> segment, XCD-aware: workgroups go to the 8 XCDs round-robin by index and every workgroup stages its WHOLE
  // tile slice, so the segments of one tile should share an L2.  Runs of 8 consecutive segments (one tile's, mostly) go to
  // one XCD, the runs round-robin over the XCDs (a contiguous eighth of the list per XCD is unbalanced: dense tiles are
  // neighbours).  With the identity map the 7 segments of a configs[4] tile ran on 7 XCDs and the kernel fetched 296 MB for
  // 29 MB of keys (FETCH_SIZE, profiles/r02_sq_counters_c5shape.json).
  const uint32_t xj = blockIdx.x >> 3;
  const uint32_t seg_i = ((xj >> 3) * 8u + (blockIdx.x & 7u)) * 8u + (xj & 7u);
  const uint4 d = seg_desc[seg_i];  // surplus workgroups read an unused (in-bounds: the table has gridDim.x entries) entry
  const uint32_t nseg = *n_seg;     // ... in the same round trip as the count that tells them so
  if (seg_i >= nseg) return;
  const uint32_t cnt = d.y, start = d.z, L = d.w;
  if (cnt == L) return;  // single-segment slice: bin_segsort_kernel wrote its ids already
  const uint32_t ns = div_up_u(L, (uint32_t)SEGN), seglen = div_up_u(L, ns);
  const uint32_t self = (d.x - start) / seglen;
  const uint64_t* __restrict__ tk = keys + start;  // the tile's slice
  uint64_t key[KPT];
  uint32_t rank[KPT];
#pragma unroll
  for (int e = 0; e < KPT; e++) {
    const uint32_t i = tid + e * NT;
    key[e] = i < cnt ? keys[(size_t)d.x + i] : ~0ull;
    rank[e] = i;
  }
  const uint32_t per_group = CAP / seglen;  // >= 2 whole segments (4 for SEGN <= 2048)
  for (uint32_t s0 = 0; s0 < ns; s0 += per_group) {
    const uint32_t s1 = min(ns, s0 + per_group);
    if (s1 - s0 == 1 && s0 == self) continue;  // the group holds only this workgroup's own segment
    const uint32_t k0 = s0 * seglen, nk = min(L, s1 * seglen) - k0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < nk; i0 += NT * 8) {  // 8 loads in flight per thread
      uint64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + u * NT + tid;
        v[u] = i < nk ? tk[k0 + i] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + u * NT + tid;
        if (i < nk) sk[i] = v[u];
      }
    }
    __syncthreads();
    for (uint32_t s2 = s0; s2 < s1; s2++) {
      if (s2 == self) continue;
      const uint32_t o2 = (s2 - s0) * seglen, len = min(seglen, L - s2 * seglen);
#pragma unroll
      for (int e = 0; e < KPT; e += 2) {
        uint32_t q0, q1;
        lds_lower_bound2<SEGN>(sk + o2, len, key[e], key[e + 1], q0, q1);
        rank[e] += q0;
        rank[e + 1] += q1;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < KPT; e++) {
    const uint32_t i = tid + e * NT;
    if (i < cnt) point_list[(size_t)start + rank[e]] = (uint32_t)key[e];
  }
}

// ================================ bucket rank (round 6): ONE launch instead of segment sort + rank merge ==================
// `split` workgroups of 1024 threads per tile (the host picks split so that every CU gets one: 4 at 64 tiles); the
// tile's keys never leave the CU between "unsorted slice" and "sorted ids".  Every workgroup of a tile
//   A. takes the tile's slice into REGISTERS (<= 16 keys per thread; longer slices are re-streamed from L2 instead) and finds
//      BOUNDS of its keys: min / max of the depth words (32-bit DPP reductions; the index words only if all depths are equal);
//   B. maps every key to t = the top 24 significant bits of (key - lower bound): a monotone function of the 64-bit key
//      (depth bits << 32 | id) -- so is floor(t * scale) for any scale > 0, with or without clamping: every key of bucket b
//      precedes every key of bucket b + 1.  LEVEL 1: 256 such buckets, histogram in LDS, scanned by one wave, which also
//      cuts THIS workgroup's share out of the sorted order -- the consecutive buckets holding keys [q L / split,
//      (q + 1) L / split), cut at bucket boundaries -- into rounds of consecutive buckets holding <= 8192 keys;
//   C. per round, LEVEL 2: the round's keys (still in their registers) onto <= 1024 fine buckets (about four keys each; the
//      level-1 table serves as the keys' CDF, so the fine buckets follow the depth density): count, scan, a returning LDS atomic on the bucket's start hands out its slots (keys
//      grouped by bucket in LDS); then every BUCKET's owner thread reads its keys once, ranks them in registers (keys
//      are unique) and puts their ids in order into an LDS id array, which leaves with coalesced stores.
// A slice of <= 1024 keys skips level 1 (one workgroup, one round, uniform fine buckets).  The order is that of a stable sort by (depth bits, id):
// the reference's (RAST/cuda_rasterizer/rasterizer_impl.cu:306-320: radix sort of tile << 32 | depth keys, stable in the id).
// Rounds 2-5 ran 66 dependent compare-exchange stages + a rank merge.
// Depth distributions that defeat the buckets are bounded, never wrong: a fine bucket of 5 .. 8 keys is ranked in registers
// too, 9 .. 64 by a wave (one lane per key); more than that (hundreds of equal depths among spread-out ones) and the round takes
// a bitonic network over the same LDS array; a level-1 bucket that alone exceeds a round (more than 8192 keys of one tile
// within 1/256 of its key range) is ranked against the streamed slice directly -- slow, correct.
constexpr int BK_THREADS = 1024;
constexpr int BK_KPT = 16;                       // keys a thread keeps in registers
constexpr int BK_INREG = BK_THREADS * BK_KPT;    // longest slice held in registers
constexpr int BK_NB1 = 256;                      // level-1 buckets (four per lane of the wave that scans them)
constexpr int BK_CAP2 = 8192;                    // keys per round (a 1 024-thread workgroup at ~126 registers fills a CU by itself:
                                                 // the LDS a second one would need is free)
constexpr int BK_SMALL = 1024;                   // slices up to this length skip level 1 (uniform fine buckets: no CDF)
constexpr int BK_NB2 = 2048;                     // level-2 buckets (four keys each on average: one or two buckets per owner thread)
constexpr uint32_t BK_LOOP_MAX = 64;             // largest fine bucket ranked by comparisons (one lane of a wave per key)
constexpr int BK_BIG_LIST = 512;                 // fine buckets of 9 .. 64 keys handed to waves per round

// Phase timeline (diagnostic, MgsOptions.dbg = 256): s_memtime stamps per (workgroup, event), thread 0 / first round only
constexpr int BK_TRACE_EVENTS = 16;
__device__ unsigned long long g_bk_trace[1024 * BK_TRACE_EVENTS];
#define MGS_BKTRACE(ev)                                                                                   \
  do {                                                                                                    \
    if (dbg && tid == 0 && blockIdx.x < 1024) g_bk_trace[(size_t)blockIdx.x * BK_TRACE_EVENTS + (ev)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

// t(key): the top 24 significant bits of key - kmin (kmin <= every key <= kmax: bounds, not necessarily attained)
struct BkMap { uint64_t kmin; uint32_t shift, tmax; };
__device__ __forceinline__ BkMap bk_map(uint64_t kmin, uint64_t kmax) {
  BkMap m;
  m.kmin = kmin;
  const uint64_t r = kmax - kmin;
  const int bits = r ? 64 - __builtin_clzll(r) : 0;
  m.shift = bits > 24 ? (uint32_t)(bits - 24) : 0u;
  m.tmax = (uint32_t)(r >> m.shift);
  return m;
}
__device__ __forceinline__ uint32_t bk_t(const BkMap& m, uint64_t key) {  // < 2^24
  const uint64_t d = key - m.kmin;
  // (one v_alignbit_b32 or one 32-bit shift -- m.shift is wave-uniform -- instead of a 64-bit shift)
  return m.shift < 32u ? __builtin_amdgcn_alignbit((uint32_t)(d >> 32), (uint32_t)d, m.shift) : ((uint32_t)(d >> 32) >> (m.shift - 32u));
}
// bucket of t inside the interval [t0, ...] at `scale` buckets per unit, clamped to [0, nb): monotone in t whatever the bounds
__device__ __forceinline__ uint32_t bk_bucket(uint32_t t, uint32_t t0, float scale, uint32_t nb) {
  const uint32_t d = t > t0 ? t - t0 : 0u;
  const uint32_t b = (uint32_t)((float)d * scale);  // d < 2^24: exact as a float; rounding and truncation keep the order
  return b < nb ? b : nb - 1u;
}
__device__ __forceinline__ uint32_t wave_umin(uint32_t v) { return ~wave_umax(~v); }

// A slice longer than the registers hold is streamed from L2 -- eight keys per thread in flight (a plain loop waits out one
// load latency per key).  f(key) is called for every key of the slice.
#define BK_STREAM(src, L, tid, kbuf, BODY)                                               \
  for (uint32_t e0_ = 0; e0_ < (L); e0_ += 8u * BK_THREADS) {                             \
    _Pragma("unroll") for (int u_ = 0; u_ < 8; u_++) {                                    \
      const uint32_t e_ = e0_ + (uint32_t)u_ * BK_THREADS + (uint32_t)(tid);              \
      kbuf[4 + u_] = e_ < (L) ? (src)[e_] : ~0ull;                                        \
    }                                                                                     \
    _Pragma("unroll") for (int u_ = 0; u_ < 8; u_++) {                                    \
      const uint32_t e_ = e0_ + (uint32_t)u_ * BK_THREADS + (uint32_t)(tid);              \
      const bool in_ = e_ < (L);                                                          \
      const uint64_t key_ = kbuf[4 + u_];                                                 \
      BODY                                                                                \
    }                                                                                     \
  }

static_assert(LDS_TILES <= BK_CAP2 && BK_BIG_LIST >= 64 + 16, "direct binning scans the tile histogram in BkShared::sid / biglist");
struct BkShared {
  uint64_t sk[BK_CAP2 + 8];       // the round's keys grouped by level-2 bucket (+ 8: an owner reads 8 slots from its start)
  uint32_t sid[BK_CAP2];          // ... their ids in order
  uint32_t cnt[BK_NB2 + 1];       // level-2 counts -> exclusive starts -> (after the placement) ends
  uint32_t c1[BK_NB1 + 1];        // level-1 counts -> exclusive starts; c1[BK_NB1] = L
  uint32_t rr[2 * BK_NB1];        // this workgroup's rounds: level-1 buckets [rr[2i], rr[2i+1])
  uint32_t tmp[64];
  uint32_t biglist[BK_BIG_LIST];  // fine buckets of 9 .. 64 keys, ranked by a wave each
  uint32_t hmin, hmax, lmin, lmax, nrounds, big, gath, nbig;
};

// Exclusive scan of cnt[0 .. nb) in place (nb a power of two, 256 .. 2048: one or two counts per thread), two barriers; *big is
// set if a count exceeds BK_LOOP_MAX.  All 1024 threads call it.
__device__ __forceinline__ void bk_scan(BkShared& S, uint32_t nb, int tid, int lane, int wv) {
  const bool two = nb > (uint32_t)BK_THREADS;  // (workgroup-uniform) 2 048 buckets: two consecutive counts per thread
  const uint32_t i0 = two ? 2u * (uint32_t)tid : (uint32_t)tid;
  const uint32_t c0 = i0 < nb ? S.cnt[i0] : 0u, c1 = two ? S.cnt[i0 + 1u] : 0u;
  if (c0 > BK_LOOP_MAX || c1 > BK_LOOP_MAX) S.big = 1u;  // (same value from everybody who writes)
  const uint32_t c = c0 + c1;
  const uint32_t incl = wave_incl_scan_add_u32(c);
  if (lane == 63) S.tmp[wv] = incl;
  __syncthreads();
  // every wave scans the 16 wave totals itself (no third barrier)
  const uint32_t wt = lane < 16 ? S.tmp[lane] : 0u;
  const uint32_t wi = wave_incl_scan_add_u32(wt);
  const uint32_t wave_off = bcast_lane_u32(wi - wt, wv);
  if (i0 < nb) S.cnt[i0] = wave_off + incl - c;
  if (two) S.cnt[i0 + 1u] = wave_off + incl - c + c0;
  __syncthreads();
}

// Rank c <= N keys of one bucket in registers: rank = how many of the others are smaller (unique keys).  Slots >= c hold ~0.
template <int N>
__device__ __forceinline__ void bk_owner_sort(BkShared& S, uint32_t b0, uint32_t c) {
  uint64_t kk[N];
  uint32_t r[N];
#pragma unroll
  for (int i = 0; i < N; i++) { kk[i] = (uint32_t)i < c ? S.sk[b0 + i] : ~0ull; r[i] = (uint32_t)(N - 1 - i); }
  // rank of slot i = (earlier slots that are smaller) + (later slots that are not larger); the padding slots (~0) are larger
  // than every key, so the ranks of the c keys are a permutation of 0 .. c - 1
#pragma unroll
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int j = i + 1; j < N; j++) {
      const uint32_t lt = kk[i] < kk[j] ? 1u : 0u;
      r[j] += lt;
      r[i] -= lt;
    }
#pragma unroll
  for (int i = 0; i < N; i++)
    if ((uint32_t)i < c) S.sid[b0 + r[i]] = (uint32_t)kk[i];
}

// Direct binning (mgs_common.h, ImgView::direct_keys): the preprocess wrote tile t's keys at keys[t * stride ...]; there was no
// bin scatter launch, so this kernel does what that kernel's table workgroup did -- every workgroup scans the tile histogram
// for its own slice of the compact id list, workgroup 0 publishes `ranges` and reports {tag, flags, R} and the reference's count
// to the host (same words, same order: see bin_scatter_kernel) and resets the preprocess's hand-shake word.
struct BkDirect {
  const uint64_t* keys;  // nullptr: the scatter kernel ran; ranges / keys_unsorted are its outputs
  uint32_t stride;
  uint32_t capacity;
  const uint32_t* tile_hist;
  const uint32_t* flags;
  const uint32_t* ref_count;
  uint2* ranges_out;
  uint64_t* host_status;
  uint32_t status_tag;
  unsigned long long* ready;
  unsigned long long nonce;
};

__global__ void __launch_bounds__(BK_THREADS) bin_bucket_emit_kernel(int T, int split, int dbg, const uint2* __restrict__ ranges,
                                                                     const uint64_t* __restrict__ keys_unsorted,
                                                                     uint32_t* __restrict__ point_list,
                                                                     unsigned long long* hs_fail_mark, BkDirect d) {
  __shared__ BkShared S;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (tile, part): the parts of a tile share the XCD (= blockIdx % 8 under round-robin dispatch) on which the
  // render kernels read the tile's list (map_block in mgs_render_common.h: tile % 8)
  const int bid = (int)blockIdx.x;
  const int tile = (bid / (8 * split)) * 8 + (bid & 7);
  const uint32_t q = (uint32_t)((bid >> 3) % split);
  // ImgView::ready[1], the preprocess's "a workgroup gave up waiting" mark: every workgroup of the bin scatter has read it
  // (kernel boundary); cleared here so that a replayed HIP graph -- same buffer, same nonce -- does not see a stale failure
  if (blockIdx.x == 0 && tid == 0 && hs_fail_mark) *hs_fail_mark = 0ull;
  if (dbg && tid < BK_TRACE_EVENTS && blockIdx.x < 1024) g_bk_trace[(size_t)blockIdx.x * BK_TRACE_EVENTS + tid] = 0ull;
  uint2 rng;
  if (d.keys) {  // (grid-uniform)
    // a preprocess workgroup gave up waiting for the zeroed tables (ready[1] == this forward's nonce): the histogram is
    // incomplete, nothing is binned, flag bit 1 tells the host why.  (The mark is read by every workgroup of this launch; the
    // render forward, next in the chain, clears it.)
    // (everything this prologue reads is requested at once: the mark, the count and -- for the 64 tiles of a 128 x 128 image --
    //  this lane's word of the histogram; read one after the other they are two dependent round trips in front of every workgroup)
    const uint32_t x64 = (T <= 64 && lane < T) ? d.tile_hist[lane] : 0u;
    const unsigned long long mark = (d.ready != nullptr && d.nonce != 0ull) ? d.ready[1] : 0ull;
    const uint32_t R_dev = d.flags[FLAG_NUM_RENDERED];
    const bool hs_failed = d.ready != nullptr && d.nonce != 0ull && mark == d.nonce;
    const uint32_t R = hs_failed ? 0xffffffffu : R_dev;
    const bool wg0 = blockIdx.x == 0;
    if (wg0 && tid == 0) {
      if (d.ready) *d.ready = 0ull;  // the hand-shake word: "not ready" for the next launch on this buffer
      if (d.host_status) {           // word 2 = {tag, the reference's count} first, then word 0 = {tag, flags, R} with release order
        __hip_atomic_store(d.host_status + 2, ((uint64_t)(d.status_tag & 0xffffu) << 48) | (uint64_t)(hs_failed ? 0u : *d.ref_count),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(d.host_status, ((uint64_t)(d.status_tag & 0xffffu) << 48) |
                           ((uint64_t)((d.flags[FLAG_PREFILTERED] & 0xfffdu) | (hs_failed ? 2u : 0u)) << 32) | (hs_failed ? 0u : R),
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (R > d.capacity) {  // the id list cannot hold the instances (or the hand-shake failed): "nothing binned", the caller retries
      if (wg0) for (int t = tid; t < T; t += BK_THREADS) d.ranges_out[t] = make_uint2(0u, 0u);
      return;
    }
    if (tile >= T) return;
    if (T <= 64) {  // (grid-uniform) the 64 tiles of a 128 x 128 image: every wave scans the histogram in its own registers
      const uint32_t x = x64;
      const uint32_t incl = wave_incl_scan_add_u32(x);
      if (wg0 && wv == 0 && lane < T) d.ranges_out[lane] = make_uint2(incl - x, incl);
      rng = make_uint2(bcast_lane_u32(incl - x, tile), bcast_lane_u32(incl, tile));
    } else {
      uint32_t* const sc = S.sid;  // (T <= LDS_TILES <= BK_CAP2 words; the array's own use starts many barriers later)
      for (int t = tid; t < T; t += BK_THREADS) sc[t] = d.tile_hist[t];
      __syncthreads();
      block_exclusive_scan(sc, T, S.biglist, &S.gath);
      if (wg0) for (int t = tid; t < T; t += BK_THREADS) d.ranges_out[t] = make_uint2(sc[t], sc[t] + d.tile_hist[t]);
      const uint32_t start = sc[tile];
      rng = make_uint2(start, start + d.tile_hist[tile]);
      __syncthreads();  // (every read of sc is done: S.biglist / S.gath / S.sid go back to their own uses)
    }
  } else {
    if (tile >= T) return;
    rng = ranges[tile];
  }
  MGS_BKTRACE(0);
  const uint32_t L = rng.y - rng.x;
  if (L == 0u) return;
  const bool small = L <= (uint32_t)BK_SMALL;   // one workgroup, one round, no level 1
  if (small && q != 0u) return;
  MGS_BKTRACE(1);
  const uint64_t* __restrict__ src = d.keys ? d.keys + (size_t)tile * d.stride : keys_unsorted + rng.x;
  uint32_t* __restrict__ dst = point_list + rng.x;
  const bool inreg = L <= (uint32_t)BK_INREG;   // (workgroup-uniform)
  const int kpt = inreg ? (int)((L + BK_THREADS - 1) / BK_THREADS) : 0;
  uint64_t k[BK_KPT];
  uint32_t have = 0u;  // bit i: register slot i holds a key
#pragma unroll
  for (int i = 0; i < BK_KPT; i++) {
    k[i] = ~0ull;
    if (i < kpt) {  // (uniform)
      const uint32_t e = (uint32_t)tid + 1024u * i;
      if (e < L) { k[i] = src[e]; have |= 1u << i; }
    }
  }
  // ---- A. bounds of the keys
  if (tid == 0) { S.hmin = ~0u; S.hmax = 0u; S.lmin = ~0u; S.lmax = 0u; S.nrounds = 0u; S.big = 0u; S.nbig = 0u; }
  if (tid <= BK_NB1) S.c1[tid] = 0u;
  S.cnt[tid] = 0u; S.cnt[tid + BK_THREADS] = 0u;
  if (tid == 0) S.cnt[BK_NB2] = 0u;
  uint32_t hlo = ~0u, hhi = 0u;
  if (inreg) {
#pragma unroll
    for (int i = 0; i < BK_KPT; i++)
      if (i < kpt && ((have >> i) & 1u)) { const uint32_t h = (uint32_t)(k[i] >> 32); hlo = min(hlo, h); hhi = max(hhi, h); }
  } else {
    BK_STREAM(src, L, tid, k, { if (in_) { const uint32_t h = (uint32_t)(key_ >> 32); hlo = min(hlo, h); hhi = max(hhi, h); } })
  }
  MGS_BKTRACE(2);
  __syncthreads();
  hlo = wave_umin(hlo); hhi = wave_umax(hhi);
  if (lane == 0) { atomicMin(&S.hmin, hlo); atomicMax(&S.hmax, hhi); }
  __syncthreads();
  uint64_t kmin = (uint64_t)S.hmin << 32, kmax = ((uint64_t)S.hmax << 32) | 0xffffffffull;
  if (S.hmin == S.hmax) {  // (uniform; rare) all depths equal: the index words carry the order
    uint32_t llo = ~0u, lhi = 0u;
    if (inreg) {
#pragma unroll
      for (int i = 0; i < BK_KPT; i++)
        if (i < kpt && ((have >> i) & 1u)) { llo = min(llo, (uint32_t)k[i]); lhi = max(lhi, (uint32_t)k[i]); }
    } else {
      BK_STREAM(src, L, tid, k, { if (in_) { llo = min(llo, (uint32_t)key_); lhi = max(lhi, (uint32_t)key_); } })
    }
    llo = wave_umin(llo); lhi = wave_umax(lhi);
    if (lane == 0) { atomicMin(&S.lmin, llo); atomicMax(&S.lmax, lhi); }
    __syncthreads();
    kmin = ((uint64_t)S.hmin << 32) | S.lmin; kmax = ((uint64_t)S.hmin << 32) | S.lmax;
  }
  MGS_BKTRACE(3);
  const BkMap m = bk_map(kmin, kmax);
  const float scale1 = (float)BK_NB1 / ((float)m.tmax + 1.0f);
  uint32_t nrounds = 1u;
  if (!small) {
    // ---- B. level-1 histogram; one wave scans it and cuts out this workgroup's rounds
    if (inreg) {
#pragma unroll
      for (int i = 0; i < BK_KPT; i++)
        if (i < kpt && ((have >> i) & 1u)) atomicAdd(&S.c1[bk_bucket(bk_t(m, k[i]), 0u, scale1, BK_NB1)], 1u);
    } else {
      BK_STREAM(src, L, tid, k, { if (in_) atomicAdd(&S.c1[bk_bucket(bk_t(m, key_), 0u, scale1, BK_NB1)], 1u); })
    }
    __syncthreads();
    MGS_BKTRACE(4);
    if (wv == 0) {
      uint32_t c[4], e[4];
#pragma unroll
      for (int j = 0; j < 4; j++) c[j] = S.c1[4 * lane + j];
      const uint32_t sum = c[0] + c[1] + c[2] + c[3];
      e[0] = wave_incl_scan_add_u32(sum) - sum;
      e[1] = e[0] + c[0]; e[2] = e[1] + c[1]; e[3] = e[2] + c[2];
#pragma unroll
      for (int j = 0; j < 4; j++) S.c1[4 * lane + j] = e[j];
      if (lane == 0) S.c1[BK_NB1] = L;
      // this part: [first bucket that starts at or behind q L / split, first bucket at or behind (q + 1) L / split)
      // (split is a power of two: shifts, not the 64-bit software division two quotients would cost this one wave)
      const int lsplit = 31 - __builtin_clz((uint32_t)split);
      const uint32_t xlo = (uint32_t)(((uint64_t)q * L) >> lsplit), xhi = (uint32_t)(((uint64_t)(q + 1u) * L) >> lsplit);
      uint32_t cl = BK_NB1, ch = BK_NB1;
#pragma unroll
      for (int j = 3; j >= 0; j--) {
        if (e[j] >= xlo) cl = 4u * lane + j;
        if (e[j] >= xhi) ch = 4u * lane + j;
      }
      const uint32_t blo = wave_umin(cl);
      const uint32_t bhi = (q + 1u < (uint32_t)split) ? wave_umin(ch) : (uint32_t)BK_NB1;
      // rounds: consecutive buckets [r0, r1) holding <= BK_CAP2 keys (at least one bucket)
      uint32_t r0 = blo, nr = 0u;
      while (r0 < bhi) {  // (wave-uniform)
        // first = start of r0: held by lane r0 / 4, slot r0 % 4
        uint32_t mine = 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) mine = (4u * lane + j == r0) ? e[j] : mine;
        const uint32_t first = wave_umax(mine);
        uint32_t cand = r0 + 1u;  // the largest b in (r0, bhi] whose keys [first, start(b)) fit a round
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t bb = 4u * lane + j;
          if (bb > r0 && bb <= bhi && e[j] - first <= (uint32_t)BK_CAP2) cand = max(cand, bb);
        }
        if (bhi == (uint32_t)BK_NB1 && L - first <= (uint32_t)BK_CAP2) cand = BK_NB1;
        const uint32_t r1 = wave_umax(cand);
        if (lane == 0) { S.rr[2u * nr] = r0; S.rr[2u * nr + 1u] = r1; }
        nr++;
        r0 = r1;
      }
      if (lane == 0) S.nrounds = nr;
    }
    __syncthreads();
    nrounds = S.nrounds;
    MGS_BKTRACE(5);
  }
  for (uint32_t rd = 0; rd < nrounds; rd++) {  // (workgroup-uniform)
    const uint32_t r0 = small ? 0u : S.rr[2u * rd], r1 = small ? (uint32_t)BK_NB1 : S.rr[2u * rd + 1u];
    const uint32_t first = small ? 0u : S.c1[r0];
    const uint32_t n = small ? L : S.c1[r1] - first;
    if (rd > 0u) {  // (the first round's tables were zeroed in the prologue)
      __syncthreads();
      S.cnt[tid] = 0u; S.cnt[tid + BK_THREADS] = 0u;
      if (tid == 0) { S.cnt[BK_NB2] = 0u; S.big = 0u; S.nbig = 0u; }
      __syncthreads();
    }
    if (n == 0u) continue;
    uint32_t* __restrict__ out = dst + first;
    if (n > (uint32_t)BK_CAP2) {
      // one level-1 bucket holds more than a round: rank each of its keys against the streamed slice (slow, correct)
      for (uint32_t e = tid; e < L; e += BK_THREADS) {
        const uint64_t v = src[e];
        if (bk_bucket(bk_t(m, v), 0u, scale1, BK_NB1) != r0) continue;
        uint32_t r = 0u;
        for (uint32_t j = 0; j < L; j++) {
          const uint64_t u = src[j];
          r += (u < v && bk_bucket(bk_t(m, u), 0u, scale1, BK_NB1) == r0) ? 1u : 0u;
        }
        out[r] = (uint32_t)v;
      }
      continue;
    }
    // ---- C. level 2.  Fine bucket of a key: the level-1 table is the keys' CDF at 256 points; inside a level-1 bucket its keys
    //         are spread linearly over as many fine buckets as the bucket has keys -- about one key per fine bucket wherever
    //         the depth density goes up or down inside the tile.  Monotone: level-1 buckets are ordered, and inside one the
    //         fraction of t * scale1 is.  (A slice without level 1: the t interval onto nb uniform buckets.)
    // Four keys per fine bucket on average: ONE owner thread per bucket ranks its <= 8 keys in registers in one batch (a
    // bucket per key costs four dependent passes per thread).
    uint32_t nb = 256;
    while (4u * nb < n) nb <<= 1;   // n / 4 <= nb < n / 2 (or 256): 256 .. 2048
    const float scale2 = (float)nb / ((float)m.tmax + 1.0f);
    // a slice that is not in registers is streamed and the round's keys take the first eight register slots in arrival order
    uint32_t inr = have;
    if (!inreg) {
      inr = 0u;
      if (tid == 0) S.gath = 0u;
      __syncthreads();
      // (one LDS atomic per wave and batch element, not per key: 1 024 same-address atomics serialise)
      BK_STREAM(src, L, tid, k, {
        const uint32_t b_ = in_ ? bk_bucket(bk_t(m, key_), 0u, scale1, BK_NB1) : 0xffffffffu;
        const bool take_ = b_ >= r0 && b_ < r1;
        const unsigned long long mk_ = ballot(take_);
        if (mk_ != 0ull) {
          uint32_t base_ = 0u;
          if (lane == 0) base_ = atomicAdd(&S.gath, (uint32_t)__builtin_popcountll(mk_));
          base_ = bcast_lane_u32(base_, 0);
          if (take_) S.sk[base_ + (uint32_t)__builtin_popcountll(mk_ & ((1ull << lane) - 1ull))] = key_;
        }
      })
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; i++) {  // (n <= 8 192: eight slots; the streaming buffer, slots 4 .. 11, is free again)
        const uint32_t e = (uint32_t)tid + 1024u * i;
        k[i] = ~0ull;
        if (e < n) { k[i] = S.sk[e]; inr |= 1u << i; }
      }
      __syncthreads();
    }
    uint32_t pk[BK_KPT / 2];  // the keys' fine buckets, two per register (0xffff: not a key of this round)
#pragma unroll
    for (int i = 0; i < BK_KPT / 2; i++) pk[i] = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < BK_KPT; i++)
      if ((inr >> i) & 1u) {
        const uint32_t t = bk_t(m, k[i]);
        uint32_t b2 = 0xffffu;
        if (small) {
          b2 = bk_bucket(t, 0u, scale2, nb);
        } else {
          const float x = (float)t * scale1;
          uint32_t b1 = (uint32_t)x;
          b1 = b1 < (uint32_t)BK_NB1 ? b1 : (uint32_t)BK_NB1 - 1u;  // (= bk_bucket(t, 0, scale1, BK_NB1): the histogram's bucket)
          if (b1 >= r0 && b1 < r1) {
            const uint32_t st = S.c1[b1], c = S.c1[b1 + 1u] - st;  // c >= 1: this key is one of them
            const uint32_t j = (uint32_t)((x - (float)b1) * (float)c);
            b2 = (st - first + (j < c ? j : c - 1u)) >> 2;
          }
        }
        if (b2 != 0xffffu) {
          atomicAdd(&S.cnt[b2], 1u);
          pk[i >> 1] = (i & 1) ? ((pk[i >> 1] & 0x0000ffffu) | (b2 << 16)) : ((pk[i >> 1] & 0xffff0000u) | b2);
        }
      }
    __syncthreads();
    if (rd == 0u) MGS_BKTRACE(8);
    bk_scan(S, nb, tid, lane, wv);
    if (rd == 0u) MGS_BKTRACE(9);
    const bool skewed = S.big != 0u;
    // the keys go to LDS grouped by bucket: a returning atomic on the bucket's START hands out its slots, so that afterwards
    // cnt[b] is the END of bucket b (= the start of b + 1) -- no arrival index is kept in registers
    uint32_t n2 = 128;
    if (skewed) {
      while (n2 < n) n2 <<= 1;
      for (uint32_t e = n + tid; e < n2; e += BK_THREADS) S.sk[e] = ~0ull;  // padding of the bitonic network below
    }
#pragma unroll
    for (int i = 0; i < BK_KPT; i++) {
      const uint32_t b2 = (pk[i >> 1] >> (16 * (i & 1))) & 0xffffu;
      if (b2 != 0xffffu) S.sk[atomicAdd(&S.cnt[b2], 1u)] = k[i];
    }
    __syncthreads();
    if (rd == 0u) MGS_BKTRACE(10);
    if (!skewed) {
      // every bucket's owner ranks its keys (each key is read once)
      for (uint32_t b = (uint32_t)tid; b < nb; b += BK_THREADS) {  // (whole waves: nb is a multiple of 64)
        const uint32_t b0 = b ? S.cnt[b - 1u] : 0u, c = S.cnt[b] - b0;
        if (ballot(c > 4u) == 0ull) {
          bk_owner_sort<4>(S, b0, c);
        } else if (c <= 8u) {
          bk_owner_sort<8>(S, b0, c);
        } else {
          // 9 .. 64 keys (the depth density peaks here): handed to a whole WAVE below -- one lane per key, c broadcast
          // reads each -- instead of c^2 reads by this one thread
          const uint32_t slot = atomicAdd(&S.nbig, 1u);
          if (slot < (uint32_t)BK_BIG_LIST) {
            S.biglist[slot] = b;
          } else {
            for (uint32_t i = 0; i < c; i++) {
              const uint64_t v = S.sk[b0 + i];
              uint32_t r = 0u;
              for (uint32_t j = 0; j < c; j++) r += S.sk[b0 + j] < v ? 1u : 0u;
              S.sid[b0 + r] = (uint32_t)v;
            }
          }
        }
      }
      __syncthreads();
      {
        const uint32_t nbig = min(S.nbig, (uint32_t)BK_BIG_LIST);
        for (uint32_t w = (uint32_t)wv; w < nbig; w += BK_THREADS / 64) {  // (wave-uniform)
          const uint32_t b = S.biglist[w];
          const uint32_t b0 = b ? S.cnt[b - 1u] : 0u, c = S.cnt[b] - b0;  // c <= BK_LOOP_MAX = 64 lanes
          const uint64_t v = (uint32_t)lane < c ? S.sk[b0 + lane] : ~0ull;
          const uint32_t vlo = (uint32_t)v, vhi = (uint32_t)(v >> 32);
          uint32_t r = 0u;
          for (uint32_t j = 0; j < c; j++) {  // key j, broadcast from lane j (scalar registers: no LDS round trip per step)
            const uint64_t u = ((uint64_t)bcast_lane_u32(vhi, (int)j) << 32) | bcast_lane_u32(vlo, (int)j);
            r += u < v ? 1u : 0u;
          }
          if ((uint32_t)lane < c) S.sid[b0 + r] = (uint32_t)v;
        }
        if (nbig) __syncthreads();  // (workgroup-uniform)
      }
      if (rd == 0u) MGS_BKTRACE(11);
      for (uint32_t j = tid; j < n; j += BK_THREADS) out[j] = S.sid[j];
      if (rd == 0u) MGS_BKTRACE(12);
    } else {
      // ---- bitonic network over the LDS array (padded to a power of two with ~0 above)
      for (uint32_t kk = 2; kk <= n2; kk <<= 1)
        for (uint32_t j = kk >> 1; j >= 1u; j >>= 1) {
          for (uint32_t c = tid; c < (n2 >> 1); c += BK_THREADS) {
            const uint32_t a = ((c & ~(j - 1u)) << 1) | (c & (j - 1u)), b = a | j;
            const uint64_t x = S.sk[a], y = S.sk[b];
            if ((x > y) == ((a & kk) == 0u)) { S.sk[a] = y; S.sk[b] = x; }
          }
          __syncthreads();
        }
      for (uint32_t j = tid; j < n; j += BK_THREADS) out[j] = (uint32_t)S.sk[j];
    }
  }
  MGS_BKTRACE(15);
}

template <int SEGN>
static void launch_sort_or_merge(int which, const BinView& b, const ImgView& im, int R, int T, hipStream_t s) {
  const int n_segments = R / SEGN + T;  // upper bound of sum_t ceil(L_t / SEGN); surplus workgroups exit at once
  constexpr int KPT = SEGN > 2048 ? 4 : 2;
  if (which == 1) {
    if constexpr (KPT == 4)
      hipLaunchKernelGGL((bin_segsort4_kernel<SEGN, 4>), dim3(n_segments), dim3(SEGN / 4), 0, s, im.seg_base + T, b.seg_desc,
                         b.keys_unsorted, b.keys, b.point_list, im.ready ? im.ready + 1 : nullptr);
    else
      hipLaunchKernelGGL(bin_segsort_kernel<SEGN>, dim3(n_segments), dim3(SEGN / 2), 0, s, im.seg_base + T, b.seg_desc,
                         b.keys_unsorted, b.keys, b.point_list, im.ready ? im.ready + 1 : nullptr);
  } else
    hipLaunchKernelGGL((bin_merge_emit_kernel<SEGN, KPT>), dim3((n_segments + 63) / 64 * 64), dim3(SEGN / KPT), 0, s, im.seg_base + T,  // (XCD map)
                       b.seg_desc, b.keys, b.point_list);
}

hipError_t launch_bin_segsort(int which, bool lds_tables, bool bucket, const GeomView& g, const BinView& b, const ImgView& im, int Pg,
                              int V, int capacity, int tiles_x, int tiles_y, int seg, int dbg, StatusSink status, hipStream_t s) {
  const int R = capacity;  // sizes the segment grids (upper bound)
  if (Pg <= 0) return hipSuccess;
  const int T = tiles_x * tiles_y;  // atlas tiles (tiles_y counts the rows of all V views)
  const int pb = pre_block((size_t)Pg);
  const int nblk = V * ((Pg + pb - 1) / pb);
  if (which == 0 && !lds_tables) {
    hipLaunchKernelGGL(bin_tables_kernel, dim3(1), dim3(1024), 0, s, T, (uint32_t)seg, (uint32_t)capacity, im.flags, status.host,
                       status.tag, im.ref_count, im.tile_hist, im.cursor, im.ranges, im.seg_base, b.seg_desc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n = V * Pg;
    hipLaunchKernelGGL(bin_scatter_global_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, tiles_x, (uint32_t)capacity, im.flags,
                       g.rect, g.depths, im.ranges, im.cursor, b.keys_unsorted);
    return hipGetLastError();
  }
  const bool direct = bucket && lds_tables && im.direct_keys != nullptr;  // the preprocess wrote the keys: no scatter launch
  if (which == 0 && direct) return hipSuccess;
  if (which == 0) {
    // the extra workgroup publishes ranges (all-empty when R == 0) and the segment table
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(nblk + 1), dim3(pb), 3 * sizeof(uint32_t) * (size_t)T, s, Pg, T,
                       tiles_x, nblk, bucket ? 0u : (uint32_t)seg, (uint32_t)capacity, g.flags, status.host, status.tag, im.ref_count, g.rect, g.depths,
                       im.tile_hist, g.blk_base, b.keys_unsorted, im.ranges, im.seg_base, b.seg_desc, im.ready, im.nonce);
    return hipGetLastError();
  }
  if (bucket) {  // which == 1: the bucket rank does the work of segment sort + rank merge; which == 2: nothing left to do
    if (which == 1) {
      // parts per tile: a power of two <= 4 that keeps the grid at ONE workgroup per CU (1 024 threads at ~126 registers fill a
      // CU's register file: a second workgroup waits for the first) -- 4 at the 64 tiles of a 128 x 128 image, 1 from 256 tiles
      // on (measured, scripts/diag/quick_split.py: 128 x 128 15.1 us with 4 parts against 18.9 / 16.6 / 25.1 with 1 / 2 / 8;
      // 500 000 Gaussians at 256 x 256 69.3 with 1 against 81.1 / 85.3 with 2 / 4)
      int split = 1;
      while (split < 4 && T * split * 2 <= 256) split *= 2;
      if (dbg & 0x7000) split = 1 << (((dbg >> 12) & 7) - 1);  // (experiments: MgsOptions.dbg bits 12-14 = 1 + log2 of the parts)
      const int grid = ((T + 7) / 8) * 8 * split;
      BkDirect d{};
      if (direct)
        d = BkDirect{im.direct_keys, im.direct_stride, (uint32_t)capacity, im.tile_hist, g.flags, im.ref_count, im.ranges,
                     status.host, status.tag, im.ready, im.nonce};
      // (direct: the failure mark is still being read by this launch's workgroups -- the render forward clears it)
      hipLaunchKernelGGL(bin_bucket_emit_kernel, dim3(grid), dim3(BK_THREADS), 0, s, T, split, dbg & 256, im.ranges, b.keys_unsorted,
                         b.point_list, (im.ready && !direct) ? im.ready + 1 : nullptr, d);
    }
    return hipGetLastError();
  }
  switch (seg) {
    case 512: launch_sort_or_merge<512>(which, b, im, R, T, s); break;
    case 1024: launch_sort_or_merge<1024>(which, b, im, R, T, s); break;
    case 4096: launch_sort_or_merge<4096>(which, b, im, R, T, s); break;
    default: launch_sort_or_merge<2048>(which, b, im, R, T, s); break;
  }
  return hipGetLastError();
}

}  // namespace mgs

// diagnostic: copy the bucket rank's phase timeline out (count = 1024 * 16 uint64)
extern "C" int mgs_debug_read_trace_bin(unsigned long long* host, size_t count) {
  const size_t n = sizeof(mgs::g_bk_trace) / sizeof(unsigned long long);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mgs::g_bk_trace), (count < n ? count : n) * sizeof(unsigned long long));
}
