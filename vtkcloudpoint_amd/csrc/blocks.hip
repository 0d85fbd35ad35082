// blocks.hip -- the reference's block-partitioned clustering ("v2.0 multithread") on MI355X.
//
//   begin   = MainForm.getClusterFromMotor, FrmMain.cs:1214-1291: bounds, order by
//             max(x-xmin, y-ymin), first ptsInCell points -> block size, (lo,hi] rectangle blocks
//             (Tools.getListByScale2, BaseClass/Tools.cs:510-513), block-major list (blockpart.hip)
//   cluster = StartCode, FrmMain.cs:2782-2794: one DBImproved(cf=0) per block -- here, over a contiguous range of
//             blocks (the unit of multi-GPU sharding): the blocks of up to 1024 points by an all-pairs kernel in LDS
//             (k_block_brute), the larger ones in ONE grouped launch of the DBSCAN engine
//   finish  = CompleteWork3, FrmMain.cs:1442-1520: per block stable order by local id, global renumber,
//             demotion of clusters of <= small_max points (with the reference's clusLen quirks), one
//             global DBImproved over all noise with cf preset, final clusForMerge order
//
// Declared deviations from the C# (same as the oracle): List.Sort's unstable tie order is
// replaced by a stable order; a block-0 point is never also filed under a rectangle; clusterSum is
// summed deterministically.  The partition (blockpart.hip) sorts nothing; rocPRIM's stable radix sort remains for the
// rare form of CompleteWork3's order when a block has more cluster ids than the LDS table of the counting sort.
#include <string.h>  // rocprim's texture_cache_iterator.hpp calls ::memset without including it

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "blocks_state.hpp"

namespace {
constexpr int BT = 256;
constexpr uint32_t NONE32 = 0xFFFFFFFFu;

int ens(vcp_ctx* ctx, DevBuf& b, size_t bytes) { return vcp_blocks_ens(ctx, b, bytes); }

template <class K, class V>
int sort_pairs(vcp_ctx* ctx, BlocksState* s, K* kin, K* kout, V* vin, V* vout, size_t n, int bits) {
  size_t tb = 0;
  VCP_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, 0, bits, ctx->stream));
  VCP_TRY(ens(ctx, s->sorttmp, tb + 64));
  VCP_HIP(ctx, rocprim::radix_sort_pairs(s->sorttmp.p, tb, kin, kout, vin, vout, n, 0, bits, ctx->stream));
  return VCP_OK;
}

int bits_for(uint64_t maxval) {
  int b = 1;
  while (b < 64 && (maxval >> b)) b++;
  return b;
}

// ---- finish --------------------------------------------------------------------------------------
// (One lane per position with the lanes of a block combined by ballot and ONE atomicMax / atomicAdd per wave and block
// was measured: 443 us against 96 -- the per-block words of neighbouring blocks share cache lines and the atomics of
// 156 k waves serialise on them.)
// per block: K_b = max local id, Z_b = number of noise points.  One wave per block walks the block's slice of
// the block-major label list (coalesced) and reduces in registers: no atomics.
// NW = 1: one wave per block (blocks of up to BIG_BLOCK positions); NW = 16: one workgroup per block of the host's list of
// large blocks, like k_block_order below.
constexpr uint32_t BIG_BLOCK = VCP_BIG_BLOCK;
template <int NW>
__global__ __launch_bounds__(NW == 1 ? BT : 64 * NW) void k_block_stats(const int32_t* __restrict__ local,
                                                                       const uint32_t* __restrict__ blockstart,
                                                                       int64_t nblocks, const uint32_t* __restrict__ biglist,
                                                                       uint32_t* __restrict__ kb, uint32_t* __restrict__ zb,
                                                                       uint32_t* __restrict__ dmisc, uint32_t b_lo) {
  // (blockstart, kb, zb, ... are indexed from the first block of this context's share: b_lo is taken off the global ids
  // of the list of large blocks only)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (NW == 1 && blockIdx.x == 0 && threadIdx.x < 8) {
    // the finish stage's counters and flags ([2] err, [4] some block has more ids than the LDS table) and the two words
    // behind kb that the scan reads: cleared here, by the first kernel of the stage, instead of by memsets
    dmisc[threadIdx.x] = 0u;
    if (threadIdx.x < 2) kb[nblocks + threadIdx.x] = 0u;
  }
  int64_t b;
  if (NW == 1) {
    b = (int64_t)blockIdx.x * (BT / 64) + w;
    if (b >= nblocks) return;
  } else {
    b = biglist[blockIdx.x] - b_lo;
  }
  const uint32_t s0 = blockstart[b], s1 = blockstart[b + 1];
  if (NW == 1 && s1 - s0 > BIG_BLOCK) return;
  uint32_t K = 0, Z = 0;
  for (uint32_t t = s0 + (NW == 1 ? lane : threadIdx.x); t < s1; t += 64 * NW) {
    const int32_t l = local[t];
    if (l == 0) Z++;
    else K = max(K, (uint32_t)l);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    K = max(K, (uint32_t)__shfl_xor((int)K, d, 64));
    Z += (uint32_t)__shfl_xor((int)Z, d, 64);
  }
  if (NW > 1) {
    __shared__ uint32_t sk[NW], sz[NW];
    if (lane == 0) {
      sk[w] = K;
      sz[w] = Z;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int k = 1; k < NW; k++) {
      K = max(K, sk[k]);
      Z += sz[k];
    }
  }
  if (lane == 0) {
    kb[b] = K;
    zb[b] = Z;
  }
}
// cluster sizes: one wave per block counts its local ids in LDS (a block holds ~ptsInCell points, so few ids);
// blocks with more ids than the LDS table fall back to global atomics
constexpr int CS_CAP = 512;
__global__ __launch_bounds__(BT) void k_cluster_sizes(const int32_t* __restrict__ local, const uint32_t* __restrict__ blockstart,
                                                     int64_t nblocks, const uint32_t* __restrict__ kb,
                                                     const uint32_t* __restrict__ cstart, uint32_t* __restrict__ csize) {
  __shared__ uint32_t cnt[BT / 64][CS_CAP];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * (BT / 64) + w;
  if (b >= nblocks) return;  // whole waves leave together; no workgroup barrier below
  const uint32_t K = kb[b];
  if (K == 0) return;
  const uint32_t c0 = cstart[b];
  if (K <= (uint32_t)CS_CAP) {
    for (uint32_t k = lane; k < K; k += 64) cnt[w][k] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = blockstart[b] + lane; t < blockstart[b + 1]; t += 64) {
      const int32_t l = local[t];
      if (l > 0) atomicAdd(&cnt[w][l - 1], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k = lane; k < K; k += 64) csize[c0 + k] = cnt[w][k];
  } else {
    for (uint32_t t = blockstart[b] + lane; t < blockstart[b + 1]; t += 64) {
      const int32_t l = local[t];
      if (l > 0) atomicAdd(&csize[c0 + (uint32_t)l - 1u], 1u);
    }
  }
}
// Final order inside a block = stable by local id (noise, id 0, first): a counting sort per block.  Ids are counted in
// LDS, scanned, then the block's positions are placed in order -- per chunk of 64 positions the lanes that hold the same
// id take consecutive slots (one ballot per distinct id in the chunk).  Replaces two library radix sorts over all m
// positions (by local id, then by block) when no block has more than CS_CAP ids.  One wave per block, four blocks per
// workgroup: blocks of up to BIG_BLOCK positions (a block holds ~ptsInCell points); the larger ones: k_block_order_big.
__global__ __launch_bounds__(BT) void k_block_order(const int32_t* __restrict__ local, const uint32_t* __restrict__ blockstart,
                                                   int64_t nblocks, const uint32_t* __restrict__ kb,
                                                   const uint32_t* __restrict__ cstart, uint32_t* __restrict__ csize,
                                                   uint32_t* __restrict__ order, uint32_t* __restrict__ ovf) {
  constexpr int NG = BT / 64;  // blocks per workgroup
  __shared__ uint32_t cnt[NG][CS_CAP + 1];
  const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * NG + g;
  if (b >= nblocks) return;  // whole waves leave together; no workgroup barrier below
  const uint32_t s0 = blockstart[b], s1 = blockstart[b + 1];
  if (s0 == s1 || s1 - s0 > BIG_BLOCK) return;
  const uint32_t K = kb[b];  // ids 0..K
  if (K > (uint32_t)CS_CAP || K == 0u) {  // (uniform per wave)
    if (K != 0u && lane == 0) *ovf = 1u;  // more ids than the LDS table: the host repeats the stage in the library-sort form
    for (uint32_t t = s0 + lane; t < s1; t += 64) order[t] = t;  // nothing but noise (most blocks of a scan's background): the list's order
    return;
  }
  for (uint32_t k = lane; k <= K; k += 64) cnt[g][k] = 0;
  __builtin_amdgcn_wave_barrier();
  for (uint32_t t = s0 + lane; t < s1; t += 64) atomicAdd(&cnt[g][local[t]], 1u);
  __builtin_amdgcn_wave_barrier();
  {  // the counts of ids 1..K are the cluster sizes CompleteWork3's demotion rule needs (k_keep)
    const uint32_t c0 = cstart[b];
    for (uint32_t k = 1 + lane; k <= K; k += 64) csize[c0 + k - 1] = cnt[g][k];
  }
  __builtin_amdgcn_wave_barrier();
  {  // exclusive scan of cnt[0..K], 64 entries per trip
    uint32_t carry = s0;
    for (uint32_t k0 = 0; k0 <= K; k0 += 64) {
      const uint32_t k = k0 + lane;
      const uint32_t v = k <= K ? cnt[g][k] : 0u;
      uint32_t inc = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
      }
      if (k <= K) cnt[g][k] = carry + inc - v;  // first final position of id k
      carry += __shfl(inc, 63, 64);
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t t0 = s0; t0 < s1; t0 += 64) {
    const uint32_t t = t0 + lane;
    const bool mine = t < s1;
    const uint32_t id = mine ? (uint32_t)local[t] : 0xFFFFFFFFu;
    unsigned long long todo = __ballot(mine);
    while (todo) {
      const int first = __ffsll((long long)todo) - 1;
      const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)id, first);  // (first is uniform: a scalar lane select)
      const unsigned long long same = __ballot(mine && id == cur);
      const uint32_t base = cnt[g][cur];
      if (mine && id == cur) order[base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = t;
      __builtin_amdgcn_wave_barrier();
      if (lane == first) cnt[g][cur] = base + (uint32_t)__popcll(same);
      __builtin_amdgcn_wave_barrier();
      todo &= ~same;
    }
  }
}
// Large blocks (the host's list), second form: one workgroup of 16 waves per block, wave w owns the w-th contiguous segment
// of the block.  Counts per (id, wave) in LDS, ONE flat exclusive scan in (id, wave) order -- which is the stable order --
// and every wave places its own segment with its own cursors: each position is read twice by one wave (the form before it
// had every wave read the whole block and place the ids congruent to its number: 146 us for the thousand large blocks of
// the 10 M-point cloud, 70 us now).
constexpr int OBW = 16;
constexpr int OBS = OBW + 1;  // row stride of the counters: the lanes of a wave (same w, different ids) hit different banks
__global__ __launch_bounds__(64 * OBW) void k_block_order_big(const int32_t* __restrict__ local,
                                                              const uint32_t* __restrict__ blockstart,
                                                              const uint32_t* __restrict__ kb,
                                                              const uint32_t* __restrict__ biglist,
                                                              const uint32_t* __restrict__ cstart, uint32_t* __restrict__ csize,
                                                              uint32_t* __restrict__ order, uint32_t* __restrict__ ovf,
                                                              uint32_t b_lo) {
  constexpr uint32_t NT = 64 * OBW;
  constexpr uint32_t EMAX = (uint32_t)(CS_CAP + 1) * OBS;
  constexpr uint32_t PER = (EMAX + NT - 1) / NT;
  __shared__ uint32_t cnt[PER * NT + 1];
  __shared__ uint32_t wsum[OBW];
  const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
  const int64_t b = (int64_t)biglist[blockIdx.x] - (int64_t)b_lo;
  const uint32_t s0 = blockstart[b], s1 = blockstart[b + 1];
  const uint32_t K = kb[b];  // ids 0..K
  if (K > (uint32_t)CS_CAP || K == 0u) {  // (uniform over the workgroup)
    if (K != 0u && tid == 0) *ovf = 1u;   // more ids than the LDS table: the host repeats the stage in the library-sort form
    for (uint32_t t = s0 + tid; t < s1; t += NT) order[t] = t;  // nothing but noise: the order is the list's
    return;
  }
  const uint32_t E = (K + 1u) * OBS;
  for (uint32_t k = tid; k < PER * NT + 1u; k += NT) cnt[k] = 0u;
  const uint32_t seg = (((s1 - s0) + OBW - 1u) / OBW + 63u) & ~63u;
  const uint32_t my0 = min(s0 + w * seg, s1), my1 = min(my0 + seg, s1);
  __syncthreads();
  for (uint32_t t = my0 + lane; t < my1; t += 64u) atomicAdd(&cnt[(uint32_t)local[t] * OBS + w], 1u);
  __syncthreads();
  {  // flat exclusive scan of cnt[0..E] (+ s0): thread t owns PER consecutive entries
    uint32_t v[PER], loc = 0u;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      v[k] = cnt[tid * PER + k];
      loc += v[k];
    }
    uint32_t inc = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t x = __shfl_up(inc, d, 64);
      if ((int)lane >= d) inc += x;
    }
    if (lane == 63u) wsum[w] = inc;
    __syncthreads();
    uint32_t cum = s0 + inc - loc;
    for (uint32_t k = 0; k < w; k++) cum += wsum[k];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      cnt[tid * PER + k] = cum;
      cum += v[k];
    }
  }
  __syncthreads();
  {  // the sizes of the clusters 1..K (k_keep): the distance between the first cursors of consecutive ids
    const uint32_t c0 = cstart[b];
    for (uint32_t k = 1u + tid; k <= K; k += NT) csize[c0 + k - 1u] = cnt[(k + 1u) * OBS] - cnt[k * OBS];
  }
  __syncthreads();
  (void)E;
  uint32_t idn = my0 + lane < my1 ? (uint32_t)local[my0 + lane] : 0xFFFFFFFFu;
  for (uint32_t t0 = my0; t0 < my1; t0 += 64u) {
    const uint32_t t = t0 + lane;
    const uint32_t id = idn;
    idn = t + 64u < my1 ? (uint32_t)local[t + 64u] : 0xFFFFFFFFu;  // (the next chunk's ids are on their way)
    const bool mine = t < my1;
    unsigned long long todo = __ballot(mine);
    while (todo) {
      const int first = __ffsll((long long)todo) - 1;
      const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)id, first);  // (first is uniform: a scalar lane select)
      const unsigned long long same = __ballot(mine && id == cur);
      const uint32_t base = cnt[cur * OBS + w];
      if (mine && id == cur) order[base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = t;
      __builtin_amdgcn_wave_barrier();
      if ((int)lane == first) cnt[cur * OBS + w] = base + (uint32_t)__popcll(same);
      __builtin_amdgcn_wave_barrier();
      todo &= ~same;
    }
  }
}
// keep[c] for cluster entry c = cstart[b] + k - 1 (FrmMain.cs:1479-1495): a cluster is demoted when the next
// id shows up and clusLen <= small_max; clusLen over-counts the first cluster of a block without noise by
// one (:1461-1465); the last cluster of a block is never checked.
__global__ __launch_bounds__(BT) void k_keep(int64_t nblocks, const uint32_t* __restrict__ cstart,
                                            const uint32_t* __restrict__ kb, const uint32_t* __restrict__ zb,
                                            const uint32_t* __restrict__ blockstart, const uint32_t* __restrict__ csize,
                                            int small_max, uint32_t* __restrict__ keep, uint32_t* __restrict__ victim_of,
                                            uint32_t* __restrict__ err, uint32_t b_lo) {
  int64_t b = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (b == 0) keep[cstart[nblocks]] = keep[cstart[nblocks] + 1] = 0u;  // the scan of keep reads one entry more (also of a share without blocks)
  if (b >= nblocks) return;
  uint32_t K = kb[b];
  victim_of[b] = NONE32;
  for (uint32_t k = 1; k <= K; k++) {
    uint32_t c = cstart[b] + k - 1;
    uint32_t eff = csize[c] + ((zb[b] == 0 && k == 1) ? 1u : 0u);
    bool demoted = (k < K) && eff <= (uint32_t)small_max;
    keep[c] = demoted ? 0u : 1u;
    if (demoted && zb[b] == 0 && k == 1) {
      // the extra clusForMerge entry that gets zeroed is the last entry of the previous non-empty block
      long long pb = b - 1;
      while (pb >= 0 && blockstart[pb + 1] == blockstart[pb]) pb--;
      if (pb >= 0) victim_of[b] = (uint32_t)pb;
      else if (b_lo == 0) atomicAdd(err, 1u);  // clusForMerge[-1]: ArgumentOutOfRangeException
      else err[3] = 1u;  // ([5] of the stage's words) the entry lies in an earlier rank's share: asked for over the exchange
    }
  }
}
__global__ __launch_bounds__(BT) void k_newlab(const int32_t* __restrict__ local, const uint32_t* __restrict__ blk_t,
                                              int64_t m, const uint32_t* __restrict__ cstart,
                                              const uint32_t* __restrict__ keep, const uint32_t* __restrict__ keeprank,
                                              int32_t* __restrict__ newlab, uint32_t b_lo) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t >= m) return;
  int32_t l = local[t];
  int32_t out = 0;
  if (l > 0) {
    uint32_t c = cstart[blk_t[t] - b_lo] + (uint32_t)l - 1u;
    if (keep[c]) out = (int32_t)keeprank[c] + 1;
  }
  newlab[t] = out;
}
__global__ __launch_bounds__(BT) void k_iota(uint32_t* __restrict__ v, int64_t m) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t < m) v[t] = (uint32_t)t;
}
__global__ __launch_bounds__(BT) void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx,
                                                  int64_t m, uint32_t* __restrict__ dst) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t < m) dst[t] = src[idx[t]];
}
// zero the victims: last entry (in final order) of block victim_of[b]
__global__ __launch_bounds__(BT) void k_victims(int64_t nblocks, const uint32_t* __restrict__ victim_of,
                                               const uint32_t* __restrict__ blockstart, const uint32_t* __restrict__ order,
                                               int32_t* __restrict__ newlab) {
  int64_t b = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (b >= nblocks) return;
  uint32_t pb = victim_of[b];
  if (pb == NONE32) return;
  newlab[order[blockstart[pb + 1] - 1]] = 0;
}
// The zero list (FrmMain.cs:1510-1515) and, inside it, the ACTIVE points of the noise pass.
// The noise pass (:1507-1516) is one DBImproved over the zero list S only.  A point of S that was NOISE inside its block
// had fewer than minPts neighbours there, and in S it has even fewer -- unless it has neighbours in OTHER blocks, i.e.
// lies within eps of its rectangle's boundary.  So a core point of the noise pass is (a) within eps of a block boundary,
// or (b) a point that carried a label inside its block and lost it (a demoted cluster, or the entry the clusLen quirk
// zeroes): the S-neighbours of such a point, if it was core in its block, are labelled-and-demoted points too.  Every
// member of a noise-pass cluster is within eps of one of its core points, hence within 2 eps of a block boundary or of
// kind (b) itself.  A = {zero-list points not more than 2 eps (+ rounding) inside their rectangle} + {kind (b)} is closed
// under "neighbour of a core candidate", so DBImproved over A alone -- same list order -- gives every point of A the
// label the pass over S would, and everybody else keeps 0.  A is an eighth of S at the reference defaults (blocks of
// ~4 x 4 units, eps 0.07): the noise pass drops from 0.68 to 0.2 ms.  (Partition on other coordinates than the
// clustering -- getClusterFromList -- has no such geometry: all_active.)
struct BandP {
  double x_Min, x_Max, y_Min, y_Max, cell_x, cell_y, r2;
  int rows, cols, all_active;
};
// Both sets leave the kernel as BITMAPS over the final order u (one word per 32 positions) plus the words' populations, whose
// exclusive scans (m / 32 entries instead of m) give every position its rank: zcnt / acnt [nw + 2], zbits / abits [nw].
__global__ __launch_bounds__(BT) void k_zero_flag(const int32_t* __restrict__ newlab, const uint32_t* __restrict__ order,
                                                 int64_t m, uint32_t* __restrict__ zcnt, uint32_t* __restrict__ zbits,
                                                 const int32_t* __restrict__ local, const double* __restrict__ motor_bm,
                                                 const uint32_t* __restrict__ blk_t, BandP B, uint32_t* __restrict__ acnt,
                                                 uint32_t* __restrict__ abits) {
  const int64_t u = (int64_t)blockIdx.x * BT + threadIdx.x;
  const int64_t nw = (m + 31) / 32;
  if (u == 0) {  // the scans read one entry more
    zcnt[nw] = zcnt[nw + 1] = 0u;
    acnt[nw] = acnt[nw + 1] = 0u;
  }
  bool z = false, act = false;
  if (u < m) {
    const uint32_t t = order[u];
    z = newlab[t] == 0;
    act = z;
    if (z && !B.all_active && local[t] == 0) {  // noise inside its block: active only near the rectangle's boundary
      const double2 v = *reinterpret_cast<const double2*>(motor_bm + 2 * (size_t)t);
      const int b = (int)blk_t[t], p = b / B.cols, q = b - p * B.cols;
      // the rectangle as FrmMain.cs:1262-1285 evaluates it (last row / column stretched to the max)
      const double lox = B.x_Min + (double)q * B.cell_x, hix = q == B.cols - 1 ? B.x_Max : B.x_Min + (double)(q + 1) * B.cell_x;
      const double loy = B.y_Min + (double)p * B.cell_y, hiy = p == B.rows - 1 ? B.y_Max : B.y_Min + (double)(p + 1) * B.cell_y;
      act = !(v.x - lox > B.r2 && hix - v.x > B.r2 && v.y - loy > B.r2 && hiy - v.y > B.r2);
    }
  }
  const unsigned long long mz = __ballot(z), ma = __ballot(act);
  const int lane = threadIdx.x & 63;
  if ((lane & 31) == 0) {  // lanes 0 and 32 store the wave's two words
    const int64_t wd = u >> 5;
    if (wd < nw) {
      const uint32_t wz = (uint32_t)(mz >> lane), wa = (uint32_t)(ma >> lane);
      zbits[wd] = wz;
      abits[wd] = wa;
      zcnt[wd] = (uint32_t)__popc(wz);
      acnt[wd] = (uint32_t)__popc(wa);
    }
  }
}
// merge_order = non-zero entries in final order, then the zero list (FrmMain.cs:1510-1520)
__global__ __launch_bounds__(BT) void k_compact(const uint32_t* __restrict__ zpre, const uint32_t* __restrict__ zbits,
                                               const uint32_t* __restrict__ order, const uint32_t* __restrict__ bl,
                                               const double* __restrict__ motor_bm, int64_t m, uint32_t Z,
                                               uint32_t* __restrict__ zrank, double* __restrict__ zcoords,
                                               int64_t* __restrict__ merge_order, int swap_xy,
                                               const uint32_t* __restrict__ apre, const uint32_t* __restrict__ abits) {
  int64_t u = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (u >= m) return;
  uint32_t t = order[u];
  const uint32_t wd = (uint32_t)(u >> 5), below = (1u << (u & 31)) - 1u;
  const uint32_t wz = zbits[wd], wa = abits[wd];
  const uint32_t zl = zpre[wd] + (uint32_t)__popc(wz & below);  // rank in the zero list (the merge order's)
  const uint32_t zr = apre[wd] + (uint32_t)__popc(wa & below);  // rank among the ACTIVE points: the noise pass's list
  if ((wz >> (u & 31)) & 1u) {
    const bool act = (wa >> (u & 31)) & 1u;
    zrank[t] = act ? zr : NONE32;  // where the noise pass will leave this point's label (NONE: stays 0)
    if (merge_order) merge_order[(m - Z) + zl] = (int64_t)bl[t];
    if (!act) return;
    // the coordinates in block-major order: t stays inside the point's block, the original index does not
    // (swap_xy: as (y, x) -- the shares of the ranks are bands in y, and the exact multi-GPU DBSCAN cuts along its first axis)
    const double2 v = *reinterpret_cast<const double2*>(motor_bm + 2 * (size_t)t);
    *reinterpret_cast<double2*>(zcoords + 2 * (size_t)zr) = swap_xy ? make_double2(v.y, v.x) : v;
  } else if (merge_order) {
    merge_order[u - zl] = (int64_t)bl[t];
  }
}
// every label by original index, in ONE pass after the noise pass: block-major positions < m carry the renumbered id or,
// for noise / demoted points, what the global noise pass gave them (zlab at their rank in the zero list); the rest of
// the list (points in no block) 0 -- the whole array is written, nothing has to be cleared first
__global__ __launch_bounds__(BT) void k_final_labels(const int32_t* __restrict__ newlab, const int32_t* __restrict__ zlab,
                                                    const uint32_t* __restrict__ zrank, const uint32_t* __restrict__ bl,
                                                    int64_t m, int64_t n, int32_t* __restrict__ labels) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t >= n) return;
  int32_t v = 0;
  if (t < m) {
    v = newlab[t];
    if (v == 0 && zlab && zrank[t] != NONE32) v = zlab[zrank[t]];
  }
  labels[bl[t]] = v;
}

// ---- a rank's share (sharded pipeline) -------------------------------------------------------------------------
// last non-empty block of the share: is there one, and the entry an EARLIER rank's first block would zero through the
// clusLen off-by-one (FrmMain.cs:1461-1465 with :1485-1488) -- the last entry, in final order, of that block
__global__ __launch_bounds__(BT) void k_last_entry(int64_t nbl, const uint32_t* __restrict__ blockstart,
                                                  const uint32_t* __restrict__ order, const int32_t* __restrict__ newlab,
                                                  uint32_t* __restrict__ dmisc) {
  __shared__ int best;
  if (threadIdx.x == 0) best = -1;
  __syncthreads();
  int loc = -1;
  for (int64_t lb = threadIdx.x; lb < nbl; lb += BT)
    if (blockstart[lb + 1] > blockstart[lb]) loc = (int)lb;
  if (loc >= 0) atomicMax(&best, loc);
  __syncthreads();
  if (threadIdx.x == 0) {
    dmisc[6] = best >= 0 ? 1u : 0u;
    dmisc[7] = dmisc[8] = 0u;
    if (best >= 0) {
      const uint32_t pos = order[blockstart[best + 1] - 1];
      dmisc[7] = newlab[pos] != 0 ? 1u : 0u;
      dmisc[8] = pos;
    }
  }
}
__global__ void k_zero_one(int32_t* __restrict__ newlab, const uint32_t* __restrict__ dmisc) { newlab[dmisc[8]] = 0; }
// (original index, final label) of every point of the share: kept clusters shifted by the clusters kept in earlier shares,
// noise / demoted points with what the global noise pass gave them, points in no block 0
__global__ __launch_bounds__(BT) void k_pairs(const int32_t* __restrict__ newlab, const int32_t* __restrict__ zlab,
                                             const uint32_t* __restrict__ zrank, const uint32_t* __restrict__ bl, int64_t m,
                                             int64_t n_loc, int32_t kept_off, int64_t* __restrict__ pairs) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t >= n_loc) return;
  int32_t v = 0;
  if (t < m) {
    v = newlab[t];
    if (v > 0) v += kept_off;
    else if (zlab && zrank[t] != NONE32) v = zlab[zrank[t]];
  }
  pairs[t] = (int64_t)(((unsigned long long)bl[t] << 32) | (unsigned long long)(uint32_t)v);
}
__global__ __launch_bounds__(BT) void k_scatter_pairs(const int64_t* __restrict__ pairs, int64_t cnt, int64_t n,
                                                     int32_t* __restrict__ labels, uint32_t* __restrict__ bad) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t >= cnt) return;
  const unsigned long long p = (unsigned long long)pairs[t];
  const uint32_t i = (uint32_t)(p >> 32);
  if ((int64_t)i < n) labels[i] = (int32_t)(uint32_t)p;
  else *bad = 1u;
}

unsigned nblk(int64_t n) { return vcp_blocks(n, BT); }

// key_in: the coordinates the PARTITION reads -- (motor_x, motor_y) in getClusterFromMotor (FrmMain.cs:1214-1291,
// Tools.getListByScale2), (X, Y) in its twin getClusterFromList (:1136-1213, Tools.getListByScale :507-509); the
// per-block DBImproved and the noise pass always cluster on motor (StartCode :2785-2786, BC/DBImproved.cs:16-21).
// NULL = the motor coordinates themselves.
int blocks_begin(vcp_ctx* ctx, const double* d_motor_in, bool from_host, const double* h_motor, const double* key_in,
                 int64_t n, double eps, int min_pts, int pts_in_cell, int small_max, int32_t* rows_o, int32_t* cols_o,
                 int64_t* m_o) {
  if (n < 0) return vcp_fail(ctx, VCP_ERR_ARG, "n < 0");
  if (n == 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "rawData.Min() on an empty list throws (FrmMain.cs:1224)");
  if (pts_in_cell <= 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "Take(0) then cell.Max() throws (FrmMain.cs:1255)");
  if (n >= 0x7FFFFFF0LL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "n beyond 32-bit indexing");
  VCP_TRY(vcp_bind(ctx));
  hipStream_t st = ctx->stream;
  if (!ctx->blocks) ctx->blocks = new BlocksState();
  BlocksState* s = ctx->blocks;
  s->ready = false;
  s->n = n;
  s->eps = eps;
  s->min_pts = min_pts;
  s->small_max = small_max;
  // host entry points upload into the state's own buffers; device entry points are read in place: the caller keeps
  // d_motor (and d_key_xy) valid and unchanged until the finish stage has returned (include/vcp.h)
  const double* motor_own = d_motor_in;
  if (from_host) {
    VCP_TRY(ens(ctx, s->motor, (size_t)n * 16));
    VCP_HIP(ctx, hipMemcpyAsync(s->motor.p, h_motor, (size_t)n * 16, hipMemcpyHostToDevice, st));
    motor_own = s->motor.as<double>();
  }
  s->motor_ptr = motor_own;
  if (key_in) {
    if (from_host) {
      VCP_TRY(ens(ctx, s->pkey, (size_t)n * 16));
      VCP_HIP(ctx, hipMemcpyAsync(s->pkey.p, key_in, (size_t)n * 16, hipMemcpyHostToDevice, st));
    }
  }
  const double* motor = key_in ? (from_host ? s->pkey.as<double>() : key_in) : motor_own;  // what the partition reads
  VCP_TRY(vcp_blocks_partition(ctx, s, motor, motor_own, n, pts_in_cell));
  s->cov_ok = true;
  s->cov_lo = s->cov_hi = s->b_lo;
  s->totalC_acc = 0;
  if (rows_o) *rows_o = s->rows;
  if (cols_o) *cols_o = s->cols;
  if (m_o) *m_o = s->m;
  const int64_t nb1 = s->nblocks + 1;
  VCP_TRY(ens(ctx, s->gtwice, (size_t)nb1 * 4));
  VCP_TRY(ens(ctx, s->gnclus, (size_t)nb1 * 4));
  s->ready = true;
  return VCP_OK;
}

// ---- cluster: small blocks ---------------------------------------------------------------------------------------
// A block of a few hundred points needs no grid: its coordinates fit in LDS and DBImproved's three facts (BaseClass/
// DBImproved.cs: core test :33-54, transitive expansion :56-90, border rule :87) come from all-pairs passes over them --
// for the sparse background of a scan (most blocks, no core point at all) that is one pass of n_b^2 predicate evaluations
// on binary64 registers against a broadcast LDS read, where the grid engine bins, sorts and tables every point over tens
// of empty cells each.  One workgroup per block of at most BRUTE_CAP points; larger blocks (the dense ones) take the
// grouped engine.  Same canonical formulation as the engine (dbscan.hip): a cluster's seed is its core point of smallest
// list position, clusters are numbered by seed, a border point takes the largest adjacent cluster and counts as "queried
// twice" when it precedes the seed of the smallest one.
__global__ void k_zero_words(uint32_t* __restrict__ p, uint32_t n) {
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) p[k] = 0u;
}
constexpr int BRT = 256;
constexpr uint32_t BRUTE_CAP = VCP_BRUTE_MAX;

__device__ __forceinline__ uint32_t brute_find(volatile uint32_t* par, uint32_t x) {
  for (;;) {
    const uint32_t y = par[x];
    if (y == x) return x;
    x = y;
  }
}
// link the trees of a and b (the smaller index becomes the root: a component's root is its seed)
__device__ __forceinline__ void brute_unite(uint32_t* par, uint32_t a, uint32_t b) {
  for (;;) {
    a = brute_find(par, a);
    b = brute_find(par, b);
    if (a == b) return;
    const uint32_t hi = max(a, b), lo = min(a, b);
    if (atomicCAS(&par[hi], hi, lo) == hi) return;
  }
}

// d = max(x - x_Min, y - y_Min), the partition's sort key (blockpart.hip: dkey)
__device__ __forceinline__ double brute_d(const double2 v, double x_Min, double y_Min) {
  const double a = v.x - x_Min, b = v.y - y_Min;
  return a > b ? a : b;
}

// The window of candidates.  d is 1-Lipschitz in the maximum norm, so two points within eps (L1) of each other have d values
// within eps -- and the block's points are stored in the order of d (FrmMain.cs:1229-1251, inside a block the order of the
// sorted list): the neighbours of the wave's 64 consecutive points lie between the first position whose d reaches
// d(first) - eps and the last whose d stays below d(last) + eps (a hair of slack for the rounding of the subtractions; the
// predicate itself stays the exact one).  On the sparse background that is a few percent of the block.  The kernel does not
// TRUST the order: it checks that d is non-decreasing over the block and takes every position otherwise (a block ordered
// by other coordinates: vcp_blocks_begin_keyed).
struct BruteWin {
  uint32_t lo, hi;
};
// The window of the GROUP of BRG consecutive points the calling lane's point p belongs to (all lanes of the wave call
// together; the BRG lanes of a group get the same answer).  It reaches a few positions beyond the group's own: the lanes of
// the group walk outwards BRG positions at a time, one LDS read and one ballot per step.
constexpr uint32_t BRG = 8;
__device__ __forceinline__ BruteWin brute_window(const double2* pt, uint32_t ng, uint32_t p, double eps, double x_Min,
                                                 double y_Min, bool sorted) {
  const uint32_t lane = threadIdx.x & 63u, sub = lane & (BRG - 1u), sh = lane & ~(BRG - 1u);
  const uint32_t first = p & ~(BRG - 1u);
  const bool active = first < ng;
  BruteWin wn{0u, active ? ng : 0u};
  const uint32_t last = active ? min(first + BRG - 1u, ng - 1u) : 0u;
  const double df = brute_d(pt[active ? first : 0u], x_Min, y_Min), dl = brute_d(pt[last], x_Min, y_Min);
  const double lo_v = df - eps - (fabs(df) + eps) * 0x1p-40, hi_v = dl + eps + (fabs(dl) + eps) * 0x1p-40;
  // eps a NaN (nothing is within it) or negative, or a block in another order: no shortcut
  bool walk = active && sorted && lo_v <= df && hi_v >= dl;
  uint32_t a = first;  // -> first position in [0, first] whose d >= lo_v
  bool more = walk;
  while (__ballot(more) != 0ull) {
    const bool in = more && a > sub && brute_d(pt[a - 1u - sub], x_Min, y_Min) >= lo_v;
    const uint32_t bits = (uint32_t)(__ballot(in) >> sh) & ((1u << BRG) - 1u);
    const uint32_t run = (uint32_t)__ffs((int)(~bits)) - 1u;  // bits has BRG bits: a zero bit at BRG at the latest
    if (more) {
      a -= min(run, BRG);
      more = run >= BRG;
    }
  }
  if (walk) wn.lo = a;
  a = last + 1u;  // -> first position in (last, ng] whose d > hi_v
  more = walk;
  while (__ballot(more) != 0ull) {
    const bool in = more && a + sub < ng && brute_d(pt[min(a + sub, ng - 1u)], x_Min, y_Min) <= hi_v;
    const uint32_t bits = (uint32_t)(__ballot(in) >> sh) & ((1u << BRG) - 1u);
    const uint32_t run = (uint32_t)__ffs((int)(~bits)) - 1u;
    if (more) {
      a += min(run, BRG);
      more = run >= BRG;
    }
  }
  if (walk) wn.hi = a;
  return wn;
}

// CAP = points of LDS: blocks of (above, CAP'] points, CAP' = min(CAP, thr_small).  Two instances: most blocks hold a few
// hundred points and their workgroups should not reserve the LDS of the largest (6 KB against 24: eight workgroups per CU)
template <uint32_t CAP>
__device__ __forceinline__ void brute_block(const uint32_t b, const double* __restrict__ motor_bm,
                                            const uint32_t* __restrict__ blockstart, uint32_t above, uint32_t thr_small, double eps,
                                            int min_pts, double x_Min, double y_Min, int32_t* __restrict__ d_local,
                                            unsigned long long* __restrict__ counters) {
  const uint32_t s0 = blockstart[b], ng = blockstart[b + 1] - s0;
  constexpr int BRP = (int)(CAP / BRT);  // points per thread
  if (ng > thr_small || ng > CAP || (above > 0u && ng <= above)) return;
  if (ng == 0) return;
  __shared__ double2 pt[CAP];
  __shared__ uint32_t par[CAP];   // core points: parent in the forest; NONE32 otherwise
  __shared__ uint32_t rnk[CAP];   // at roots: the cluster's rank among the block's clusters
  __shared__ uint32_t wsum[BRT / 64];
  __shared__ uint32_t s_twice, s_any, s_unsorted;
  const uint32_t t = threadIdx.x;
  const int lane = t & 63, w = t >> 6;
  const double2* src = reinterpret_cast<const double2*>(motor_bm) + s0;
  for (uint32_t j = t; j < ng; j += BRT) pt[j] = src[j];
  if (t == 0) s_twice = s_any = s_unsorted = 0u;
  __syncthreads();
  double ax[BRP], ay[BRP];
  uint32_t c[BRP];
  {
    bool bad = false;
#pragma unroll
    for (int u = 0; u < BRP; u++) {
      const uint32_t p = t + (uint32_t)u * BRT;
      const double2 a = p < ng ? pt[p] : make_double2(NAN, NAN);  // (a NaN is within eps of nothing)
      ax[u] = a.x;
      ay[u] = a.y;
      c[u] = 0u;
      if (p > 0u && p < ng && !(brute_d(pt[p - 1u], x_Min, y_Min) <= brute_d(a, x_Min, y_Min))) bad = true;
    }
    if (__ballot(bad) != 0ull && lane == 0) s_unsorted = 1u;
  }
  __syncthreads();
  const bool sorted = s_unsorted == 0u;
  // the candidate windows of the lane's points, one per round of 256 points (shared by the BRG lanes of a group)
  BruteWin wn[BRP];
#pragma unroll
  for (int u = 0; u < BRP; u++) {
    wn[u].lo = wn[u].hi = 0u;
    if ((uint32_t)u * BRT + (uint32_t)w * 64u < ng)  // (uniform over the wave)
      wn[u] = brute_window(pt, ng, t + (uint32_t)u * BRT, eps, x_Min, y_Min, sorted);
  }
  // 1. neighbours within eps (itself included)
#pragma unroll
  for (int u = 0; u < BRP; u++) {
    uint32_t cc = 0u;
    const double x = ax[u], y = ay[u];
#pragma unroll 4
    for (uint32_t j = wn[u].lo; j < wn[u].hi; j++) {
      const double2 q = pt[j];
      cc += (fabs(x - q.x) + fabs(y - q.y) <= eps) ? 1u : 0u;
    }
    c[u] = cc;
  }
  bool core[BRP];
  uint32_t anycore = 0u;
#pragma unroll
  for (int u = 0; u < BRP; u++) {
    const uint32_t p = t + (uint32_t)u * BRT;
    core[u] = p < ng && (int)c[u] >= min_pts;
    if (p < ng) par[p] = core[u] ? p : NONE32;
    anycore |= core[u] ? 1u : 0u;
  }
  if (__ballot(anycore) != 0ull && lane == 0) s_any = 1u;
  __syncthreads();
  if (s_any == 0u) {  // no core point: every point is noise, no cluster (uniform over the workgroup)
    for (uint32_t j = t; j < ng; j += BRT) d_local[s0 + j] = 0;
    if (t == 0) atomicAdd(&counters[2 * (b & 63u)], (unsigned long long)ng * (unsigned long long)ng);
    return;
  }
  // from here on the LDS copy serves as the list of CORE points: the others get a NaN abscissa, within eps of nothing
  // (the windows were taken from the unchanged copy above)
#pragma unroll
  for (int u = 0; u < BRP; u++) {
    const uint32_t p = t + (uint32_t)u * BRT;
    if (p < ng && !core[u]) pt[p].x = NAN;
  }
  __syncthreads();
  // 2. components of the core points: a core point links to every core point before it
#pragma unroll
  for (int u = 0; u < BRP; u++) {
    if (__ballot(core[u]) == 0ull) continue;  // (uniform over the wave)
    const uint32_t p = t + (uint32_t)u * BRT;
    const uint32_t wend = core[u] ? min(wn[u].hi, p) : 0u;  // the core points before it
    const double x = ax[u], y = ay[u];
    for (uint32_t j = wn[u].lo; j < wend; j++) {
      const double2 q = pt[j];
      if (fabs(x - q.x) + fabs(y - q.y) <= eps) brute_unite(par, p, j);
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < BRP; u++) {
    const uint32_t p = t + (uint32_t)u * BRT;
    if (core[u]) {
      const uint32_t r = brute_find(par, p);
      // (a root keeps itself; writing other entries while neighbours still walk through them is safe: every value ever
      // stored in par[p] is an ancestor of p)
      if (r != p) par[p] = r;
    }
  }
  __syncthreads();
  // ranks of the roots in index order = numbering by seed
  uint32_t run = 0u;
#pragma unroll
  for (int u = 0; u < BRP; u++) {
    if ((uint32_t)u * BRT >= ng) break;
    const uint32_t p = t + (uint32_t)u * BRT;
    const bool root = core[u] && par[p] == p;
    const unsigned long long m = __ballot(root);
    if (lane == 0) wsum[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = run;
    for (int k = 0; k < w; k++) before += wsum[k];
    if (root) rnk[p] = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (int k = 0; k < BRT / 64; k++) run += wsum[k];
    __syncthreads();
  }
  const uint32_t kg = run;
  // 3. labels: core points their component; the others the largest adjacent cluster (roots are in rank order)
  uint32_t twice = 0u;
#pragma unroll
  for (int u = 0; u < BRP; u++) {
    if ((uint32_t)u * BRT >= ng) break;
    const uint32_t p = t + (uint32_t)u * BRT;
    const bool cand = p < ng && !core[u] && c[u] > 1u;
    uint32_t mxr = 0u, mnr = NONE32;
    bool any = false;
    if (__ballot(cand) != 0ull) {
      const double x = ax[u], y = ay[u];
      for (uint32_t j = wn[u].lo, je = cand ? wn[u].hi : 0u; j < je; j++) {
        const double2 q = pt[j];
        if (fabs(x - q.x) + fabs(y - q.y) <= eps) {
          const uint32_t r = par[j];
          mxr = max(mxr, r);
          mnr = min(mnr, r);
          any = true;
        }
      }
    }
    if (p < ng) {
      int32_t lab = 0;
      if (core[u]) lab = (int32_t)rnk[par[p]] + 1;
      else if (any) {
        lab = (int32_t)rnk[mxr] + 1;
        if (p < mnr) twice++;
      }
      d_local[s0 + p] = lab;
    }
  }
  if (twice) atomicAdd(&s_twice, twice);
  __syncthreads();
  if (t == 0) {
    const uint32_t tw = s_twice;
    atomicAdd(&counters[2 * (b & 63u)], (unsigned long long)ng * ((unsigned long long)ng + kg + tw));
    atomicAdd(&counters[2 * (b & 63u) + 1], (unsigned long long)kg);
  }
}
// blocks [lo, hi): the grid is capped and walks them (one workgroup per block would wrap at 2^32 threads: 16 M blocks)
template <uint32_t CAP>
__global__ __launch_bounds__(BRT) void k_block_brute(const double* __restrict__ motor_bm, const uint32_t* __restrict__ blockstart,
                                                    uint32_t lo, uint32_t hi, uint32_t above, uint32_t thr_small, double eps,
                                                    int min_pts, double x_Min, double y_Min, int32_t* __restrict__ d_local,
                                                    unsigned long long* __restrict__ counters) {
  for (uint32_t b = lo + blockIdx.x; b < hi; b += gridDim.x) {
    brute_block<CAP>(b, motor_bm, blockstart, above, thr_small, eps, min_pts, x_Min, y_Min, d_local, counters);
    __syncthreads();  // the LDS arrays are the next block's
  }
}

int blocks_cluster(vcp_ctx* ctx, int32_t lo, int32_t hi, int32_t* d_local, int64_t* evals_o) {
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_begin has not run");
  if (hi < 0) hi = (int32_t)s->nblocks;
  if (lo < 0 || hi > s->nblocks || lo > hi) return vcp_fail(ctx, VCP_ERR_ARG, "block range");
  if (lo < hi && (lo < s->b_lo || hi > s->b_hi)) return vcp_fail(ctx, VCP_ERR_ARG, "block range outside this context's share");
  if (evals_o) *evals_o = 0;
  if (lo == hi) return VCP_OK;
  if (s->m == 0) return VCP_OK;
  // blocks of at most brute_thr points: all pairs in LDS; the others: one grouped launch of the grid engine, which sees
  // the points of the small blocks as excluded (group -1 in grp_big, written by the partition)
  const uint32_t thr = s->brute_thr;
  uint64_t pts_small = 0, pts_big = 0;
  bool any_mid = false;  // blocks beyond the small instance of the all-pairs kernel
  if (thr > 0) {
    for (int64_t b = lo; b < hi; b++) {
      const uint32_t c = s->h_blockstart[(size_t)b + 1] - s->h_blockstart[(size_t)b];
      if (c <= thr) pts_small += c;
      else pts_big += c;
      if (c <= thr && c > (uint32_t)BRT) any_mid = true;
    }
  } else {
    pts_big = 1;
  }
  // The small blocks run on a second stream beside the engine's launches (their outputs are disjoint: positions of d_local,
  // counters of their own); both are joined before the host reads anything.
  int64_t ev = 0;
  int32_t cf = 0;
  unsigned long long* hb = reinterpret_cast<unsigned long long*>(ctx->pinned) + 128;
  if (pts_small > 0) {
    if (!s->side) {
      VCP_HIP(ctx, hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
      VCP_HIP(ctx, hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    }
    VCP_TRY(ens(ctx, s->brutecnt, 128 * sizeof(unsigned long long)));
    unsigned long long* bc = s->brutecnt.as<unsigned long long>();  // [64][2]: op counter, clusters of the small blocks
    VCP_HIP(ctx, hipEventRecord(s->ev_fork, ctx->stream));
    VCP_HIP(ctx, hipStreamWaitEvent(s->side, s->ev_fork, 0));
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(BT), 0, s->side, reinterpret_cast<uint32_t*>(bc), 256u);
    const unsigned gb = (unsigned)std::min<int64_t>((int64_t)hi - lo, (int64_t)1 << 22);
    hipLaunchKernelGGL(k_block_brute<BRT>, dim3(gb), dim3(BRT), 0, s->side, s->motor_bm.as<double>(),
                       s->blockstart.as<uint32_t>(), (uint32_t)lo, (uint32_t)hi, 0u, thr, s->eps, s->min_pts, s->x_Min, s->y_Min,
                       d_local, bc);
    if (thr > (uint32_t)BRT && any_mid)
      hipLaunchKernelGGL(k_block_brute<BRUTE_CAP>, dim3(gb), dim3(BRT), 0, s->side, s->motor_bm.as<double>(),
                         s->blockstart.as<uint32_t>(), (uint32_t)lo, (uint32_t)hi, (uint32_t)BRT, thr, s->eps, s->min_pts,
                         s->x_Min, s->y_Min, d_local, bc);
    VCP_HIP(ctx, hipMemcpyAsync(hb, bc, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->side));
    VCP_HIP(ctx, hipGetLastError());
  }
  if (pts_big > 0) {
    DbscanExt ext;
    ext.d_group = thr > 0 ? s->grp_big.as<int32_t>()
                          : reinterpret_cast<const int32_t*>(s->blk_t.as<uint32_t>());  // block id per block-major position
    ext.skip_upto = thr;
    ext.d_ord = nullptr;  // list position = input index
    ext.d_groupstart = s->blockstart.as<uint32_t>();
    ext.G = (int32_t)s->nblocks;
    ext.only_lo = lo;
    ext.only_hi = hi;
    ext.d_group_twice = s->gtwice.as<uint32_t>();
    ext.d_group_nclus = s->gnclus.as<uint32_t>();
    const double bbox[6] = {s->mbox[0], s->mbox[2], 0.0, s->mbox[1], s->mbox[3], 0.0};  // the partition has seen every point
    ext.h_bbox = bbox;
    // input = the m points that fell in a block, in block-major order; positions outside [lo, hi)'s slice are left alone
    const int rc = vcp_dbscan_engine(ctx, s->motor_bm.as<double>(), s->m, 2, VCP_L1_2D, s->eps, s->min_pts, 0, nullptr, d_local,
                                     nullptr, nullptr, &cf, &ev, &ext);
    if (rc != VCP_OK) {
      if (pts_small > 0) (void)hipStreamSynchronize(s->side);
      return rc;
    }
  }
  VCP_HIP(ctx, hipGetLastError());
  if (pts_small > 0) {
    VCP_HIP(ctx, hipStreamSynchronize(s->side));
    for (int k = 0; k < 64; k++) {
      ev += (int64_t)hb[2 * k];
      cf += (int32_t)hb[2 * k + 1];
    }
  }
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (evals_o) *evals_o = ev;
  // the clusters of the blocks clustered so far (the finish stage sizes its arrays with it when every block is in)
  if (lo == s->cov_hi) {
    s->cov_hi = hi;
    s->totalC_acc += cf;
  } else {
    s->cov_ok = false;
  }
  return VCP_OK;
}

// CompleteWork3 (FrmMain.cs:1442-1520) over this context's share of the blocks, in stages:
//   local   per block: stable order by local id, renumbering, demotion of small clusters (with the clusLen quirks);
//           reads back the stage's counters: clusters, kept ones, the error / request flags
//   zero    (after the ranks have told each other who has to zero whose last entry) the zero list: Z, positions
//   zcoords the coordinates of the zero list (the input of the global noise pass) and the merge order
// The single-device call runs them back to back.
int finish_local(vcp_ctx* ctx, const int32_t* d_local, bool sharded, bool force_sort) {
  BlocksState* s = ctx->blocks;
  hipStream_t st = ctx->stream;
  const int64_t m = s->m, nb = s->b_hi - s->b_lo;  // blocks of the share
  const uint32_t b_lo = (uint32_t)s->b_lo;
  const uint32_t* blockstart = s->blockstart.as<uint32_t>() + s->b_lo;
  VCP_TRY(ens(ctx, s->kb, (size_t)(nb + 2) * 4));
  VCP_TRY(ens(ctx, s->zb, (size_t)(nb + 2) * 4));
  VCP_TRY(ens(ctx, s->cstart, (size_t)(nb + 2) * 4));
  const uint32_t* blk_t = s->blk_t.as<uint32_t>();
  uint32_t* kb = s->kb.as<uint32_t>();
  uint32_t* zb = s->zb.as<uint32_t>();
  uint32_t* cstart = s->cstart.as<uint32_t>();
  // [0] total clusters, [1] kept, [2] err, [3] Z, [4] id table overflow, [5] request to an earlier share, [6..8] k_last_entry
  uint32_t* dmisc = s->misc.as<uint32_t>();
  const unsigned nbw = (unsigned)((std::max<int64_t>(nb, 1) + BT / 64 - 1) / (BT / 64));  // one wave per block
  hipLaunchKernelGGL(k_block_stats<1>, dim3(nbw), dim3(BT), 0, st, d_local, blockstart, nb, nullptr, kb, zb, dmisc, b_lo);
  if (s->nbig)
    hipLaunchKernelGGL(k_block_stats<16>, dim3(s->nbig), dim3(1024), 0, st, d_local, blockstart, nb,
                       s->biglist.as<uint32_t>(), kb, zb, dmisc, b_lo);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, kb, cstart, nb + 1, dmisc));  // cstart[nb] = total clusters
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  // The number of clusters sizes two arrays and a scan.  When this context clustered every block of the share itself (one
  // call or a sequence of adjoining ranges) the engine has already reported it; otherwise (label slices gathered from other
  // devices) it is read back here.
  uint32_t totalC;
  if (s->cov_ok && s->cov_lo == s->b_lo && s->cov_hi == s->b_hi) {
    totalC = (uint32_t)s->totalC_acc;
  } else {
    VCP_HIP(ctx, hipMemcpyAsync(hp, dmisc, 4, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    totalC = hp[0];
  }
  VCP_TRY(ens(ctx, s->csize, (size_t)(totalC + 2) * 4));
  VCP_TRY(ens(ctx, s->keep, (size_t)(totalC + 2) * 4 * 2));
  VCP_TRY(ens(ctx, s->tmp3, (size_t)(nb + 2) * 4));
  uint32_t* csize = s->csize.as<uint32_t>();
  uint32_t* keep = s->keep.as<uint32_t>();
  uint32_t* keeprank = keep + (totalC + 2);
  uint32_t* victim_of = s->tmp3.as<uint32_t>();
  // final order inside a block: stable by local id -- a per-block counting sort (which also yields the cluster sizes),
  // or, when some block has more ids than its LDS table (flagged by the kernel, seen at the stage's one read-back), the
  // library-sort form: positions by local id, then by block
  VCP_TRY(ens(ctx, s->order, (size_t)(m + 1) * 4));
  VCP_TRY(ens(ctx, s->tmp2, (size_t)(m + 2) * 4));
  uint32_t* order = s->order.as<uint32_t>();
  VCP_TRY(ens(ctx, s->newlab, (size_t)(m + 1) * 4));
  int32_t* newlab = s->newlab.as<int32_t>();
  bool by_sort = force_sort || getenv("VCP_BLOCKS_ORDER_SORT") != nullptr;  // (the variable: test switch)
  for (;;) {
    if (m > 0 && !by_sort) {
      hipLaunchKernelGGL(k_block_order, dim3(nbw), dim3(BT), 0, st, d_local, blockstart, nb, kb, cstart, csize, order,
                         dmisc + 4);
      if (s->nbig)
        hipLaunchKernelGGL(k_block_order_big, dim3(s->nbig), dim3(64 * OBW), 0, st, d_local, blockstart, kb,
                           s->biglist.as<uint32_t>(), cstart, csize, order, dmisc + 4, b_lo);
    } else if (m > 0) {
      VCP_TRY(ens(ctx, s->tmp0, (size_t)(m + 1) * 8));
      VCP_TRY(ens(ctx, s->tmp1, (size_t)(m + 1) * 8));
      uint32_t* iota = s->tmp2.as<uint32_t>();
      uint32_t* k1 = s->tmp0.as<uint32_t>();
      uint32_t* k1o = k1 + (m + 1);
      uint32_t* v1o = s->tmp1.as<uint32_t>();
      uint32_t* k2 = v1o + (m + 1);
      VCP_HIP(ctx, hipMemsetAsync(csize, 0, (size_t)(totalC + 2) * 4, st));
      hipLaunchKernelGGL(k_cluster_sizes, dim3(nbw), dim3(BT), 0, st, d_local, blockstart, nb, kb, cstart, csize);
      hipLaunchKernelGGL(k_iota, dim3(nblk(m)), dim3(BT), 0, st, iota, m);
      VCP_HIP(ctx, hipMemcpyAsync(k1, d_local, (size_t)m * 4, hipMemcpyDeviceToDevice, st));
      VCP_TRY(sort_pairs(ctx, s, k1, k1o, iota, v1o, (size_t)m, bits_for((uint64_t)m)));  // a local id is at most m
      hipLaunchKernelGGL(k_gather_u32, dim3(nblk(m)), dim3(BT), 0, st, blk_t, v1o, m, k2);
      VCP_TRY(sort_pairs(ctx, s, k2, k1o, v1o, order, (size_t)m, bits_for((uint64_t)s->nblocks)));
    }
    hipLaunchKernelGGL(k_keep, dim3(nblk(nb)), dim3(BT), 0, st, nb, cstart, kb, zb, blockstart, csize, s->small_max, keep,
                       victim_of, dmisc + 2, b_lo);
    VCP_TRY(vcp_exclusive_scan_u32(ctx, keep, keeprank, (int64_t)totalC + 1, dmisc + 1));
    hipLaunchKernelGGL(k_newlab, dim3(nblk(m)), dim3(BT), 0, st, d_local, blk_t, m, cstart, keep, keeprank, newlab, b_lo);
    hipLaunchKernelGGL(k_victims, dim3(nblk(nb)), dim3(BT), 0, st, nb, victim_of, blockstart, order, newlab);
    if (!sharded) break;  // (the single device reads the counters once, after the zero list: finish_zero)
    hipLaunchKernelGGL(k_last_entry, dim3(1), dim3(BT), 0, st, nb, blockstart, order, newlab, dmisc);
    VCP_HIP(ctx, hipMemcpyAsync(hp, dmisc, 48, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    if (hp[4] == 0 || by_sort) break;
    by_sort = true;  // some block has more than CS_CAP cluster ids: once more, order by the library sort
    VCP_HIP(ctx, hipMemsetAsync(dmisc + 2, 0, 4, st));
    VCP_HIP(ctx, hipMemsetAsync(dmisc + 5, 0, 4, st));
  }
  s->f_by_sort = by_sort;
  s->f_totalC = totalC;
  s->f_local = d_local;
  if (sharded) {
    if (hp[0] != totalC) return vcp_fail(ctx, VCP_ERR_ARG, "the label array does not hold what vcp_blocks_cluster_dev produced");
    s->f_kept = hp[1];
    s->f_err = hp[2];
    s->f_req = hp[5];
    s->f_nonempty = hp[6];
    s->f_last_nonzero = hp[7];
  }
  return VCP_OK;
}

// returns 1 in *again when the single-device pass has to be repeated in the library-sort form
int finish_zero(vcp_ctx* ctx, bool sharded, int zero_last, bool* again) {
  BlocksState* s = ctx->blocks;
  hipStream_t st = ctx->stream;
  const int64_t m = s->m;
  uint32_t* dmisc = s->misc.as<uint32_t>();
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  int32_t* newlab = s->newlab.as<int32_t>();
  uint32_t* order = s->order.as<uint32_t>();
  if (again) *again = false;
  if (sharded && zero_last) hipLaunchKernelGGL(k_zero_one, dim3(1), dim3(1), 0, st, newlab, dmisc);
  // zero list (FrmMain.cs:1510-1515)
  const int64_t nw = (m + 31) / 32;  // words of the two bitmaps; each buffer: [nw + 2] word counts, then [nw] words
  VCP_TRY(ens(ctx, s->zflag, (size_t)(2 * nw + 4) * 4));
  VCP_TRY(ens(ctx, s->zlist, (size_t)(2 * nw + 4) * 4));
  uint32_t* zcnt = s->zflag.as<uint32_t>();
  uint32_t* acnt = s->zlist.as<uint32_t>();
  BandP B;
  B.x_Min = s->x_Min;
  B.x_Max = s->x_Max;
  B.y_Min = s->y_Min;
  B.y_Max = s->y_Max;
  B.cell_x = s->cell_x;
  B.cell_y = s->cell_y;
  B.rows = s->rows;
  B.cols = s->cols;
  // 2 eps + the roundings of the differences involved (a neighbour pair has |dx| <= eps (1 + 2^-52), see k_zero_flag)
  B.r2 = 2.0 * s->eps * (1.0 + 1.0 / 1099511627776.0);
  static const bool band_off = getenv("VCP_NOISE_ALL") != nullptr;  // test switch: the whole zero list
  B.all_active = (s->d_key != s->d_motor || !(s->eps >= 0.0) || !std::isfinite(B.r2) || band_off) ? 1 : 0;
  hipLaunchKernelGGL(k_zero_flag, dim3(nblk(m)), dim3(BT), 0, st, newlab, order, m, zcnt, zcnt + nw + 2, s->f_local,
                     s->motor_bm.as<double>(), s->blk_t.as<uint32_t>(), B, acnt, acnt + nw + 2);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, zcnt, zcnt, nw + 1, dmisc + 3));
  VCP_TRY(vcp_exclusive_scan_u32(ctx, acnt, acnt, nw + 1, dmisc + 9));
  VCP_HIP(ctx, hipMemcpyAsync(hp, dmisc, 48, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (!sharded) {
    if (hp[4] != 0 && !s->f_by_sort) {
      if (again) *again = true;
      return VCP_OK;
    }
    if (hp[0] != s->f_totalC)
      return vcp_fail(ctx, VCP_ERR_ARG, "the label array does not hold what vcp_blocks_cluster_dev produced");
    s->f_kept = hp[1];
    s->f_err = hp[2];
  }
  s->f_Z = hp[3];
  s->f_A = hp[9];
  return VCP_OK;
}

int finish_zcoords(vcp_ctx* ctx, double* d_zcoords, int64_t* d_merge_order, int swap_xy) {
  BlocksState* s = ctx->blocks;
  hipStream_t st = ctx->stream;
  const int64_t m = s->m;
  const int64_t nw = (m + 31) / 32;
  if (m > 0)
    hipLaunchKernelGGL(k_compact, dim3(nblk(m)), dim3(BT), 0, st, s->zflag.as<uint32_t>(), s->zflag.as<uint32_t>() + nw + 2,
                       s->order.as<uint32_t>(), s->bl.as<uint32_t>(), s->motor_bm.as<double>(), m, s->f_Z,
                       s->tmp2.as<uint32_t>() /* zrank: free again (it held the identity of the library-sort order) */,
                       d_zcoords, d_merge_order, swap_xy, s->zlist.as<uint32_t>(), s->zlist.as<uint32_t>() + nw + 2);
  VCP_HIP(ctx, hipGetLastError());
  return VCP_OK;
}

int blocks_finish(vcp_ctx* ctx, const int32_t* d_local, int64_t evals_blocks, int32_t* d_labels, int64_t* d_merge_order,
                  int32_t* kept_o, int32_t* del_o, int32_t* ca_o, int64_t* evals_o) {
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_begin has not run");
  if (s->b_lo != 0 || s->b_hi != s->nblocks)
    return vcp_fail(ctx, VCP_ERR_ARG, "this context holds a share of the blocks: use the vcp_blocks_finish_*_dev stages");
  hipStream_t st = ctx->stream;
  const int64_t n = s->n, m = s->m;
  bool again = false;
  VCP_TRY(finish_local(ctx, d_local, false, false));
  VCP_TRY(finish_zero(ctx, false, 0, &again));
  if (again) {  // some block has more than CS_CAP cluster ids: once more, order by the library sort
    VCP_TRY(finish_local(ctx, d_local, false, true));
    VCP_TRY(finish_zero(ctx, false, 0, nullptr));
  }
  if (s->f_err != 0)
    return vcp_fail(ctx, VCP_ERR_INDEX, "clusForMerge index -1 while demoting the first cluster (FrmMain.cs:1487)");
  const uint32_t kept = s->f_kept, Z = s->f_Z, A = s->f_A;  // A = active points of the zero list (k_zero_flag)
  const uint32_t delSum = s->f_totalC - kept;
  VCP_TRY(ens(ctx, s->zcoords, (size_t)(A + 1) * 16));
  VCP_TRY(ens(ctx, s->zlab, (size_t)(A + 1) * 4));
  VCP_TRY(finish_zcoords(ctx, s->zcoords.as<double>(), d_merge_order, 0));
  uint32_t* zrank = s->tmp2.as<uint32_t>();
  int32_t* newlab = s->newlab.as<int32_t>();
  // FrmMain.cs:1507-1516: one DBImproved over all noise, cf preset to the kept-cluster count
  int32_t cf = (int32_t)kept;
  int64_t ev = 0;
  if (A > 0) {
    DbscanExt zext;  // the noise points lie inside the cloud's box
    const double bbox[6] = {s->mbox[0], s->mbox[2], 0.0, s->mbox[1], s->mbox[3], 0.0};
    zext.h_bbox = bbox;
    // the noise points are spread over the cloud's box unless far outliers stretch it -- which shows in the block grid: a
    // first block of ptsInCell points that is a speck of the box means rows x cols far beyond n / ptsInCell
    zext.no_trim = (double)s->nblocks <= 16.0 * ((double)s->n / (double)std::max(s->take, 1) + 1.0);
    VCP_TRY(vcp_dbscan_engine(ctx, s->zcoords.as<double>(), (int64_t)A, 2, VCP_L1_2D, s->eps, s->min_pts, (int32_t)kept,
                              nullptr, s->zlab.as<int32_t>(), nullptr, nullptr, &cf, &ev, &zext));
  }
  // iritatorNum of the pass over the whole zero list: Z x (queried points + seeds + border points queried twice); the
  // engine reports A x (A + K + twice) for the active points, whose clusters and order are the same
  {
    const int64_t K = (int64_t)cf - (int64_t)kept;
    const int64_t twice = A > 0 ? ev / (int64_t)A - (int64_t)A - K : 0;
    ev = (int64_t)Z * ((int64_t)Z + K + twice);
  }
  // labels by original index: kept clusters and the noise pass result, one scatter
  hipLaunchKernelGGL(k_final_labels, dim3(nblk(n)), dim3(BT), 0, st, newlab, A > 0 ? s->zlab.as<int32_t>() : nullptr, zrank,
                     s->bl.as<uint32_t>(), m, n, d_labels);
  VCP_HIP(ctx, hipGetLastError());
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (kept_o) *kept_o = (int32_t)kept;
  if (del_o) *del_o = (int32_t)delSum;
  if (ca_o) *ca_o = cf;
  if (evals_o) *evals_o = evals_blocks + ev;
  return VCP_OK;
}

}  // namespace

extern "C" {

void vcp_blocks_state_free(vcp_ctx* ctx) {
  if (!ctx || !ctx->blocks) return;
  BlocksState* s = ctx->blocks;
  DevBuf* all[] = {&s->motor, &s->pkey, &s->blockof, &s->bl, &s->motor_bm, &s->blockstart,
                   &s->gtwice, &s->gnclus, &s->tmp0, &s->tmp1, &s->tmp2, &s->tmp3, &s->sorttmp, &s->blk_t, &s->csize,
                   &s->cstart, &s->kb, &s->zb, &s->keep, &s->order, &s->newlab, &s->zflag, &s->zlist, &s->zcoords,
                   &s->zlab, &s->misc, &s->biglist, &s->sel, &s->cand, &s->counts, &s->rec, &s->rec2, &s->stage, &s->rank, &s->binfo, &s->slicelist,
                   &s->vlist, &s->fall, &s->gcnt, &s->grp_big, &s->brutecnt, &s->counts_t};
  for (DevBuf* b : all)
    if (b->p) (void)hipFree(b->p);
  if (s->side) {
    (void)hipStreamDestroy(s->side);
    (void)hipEventDestroy(s->ev_fork);
  }
  delete s;
  ctx->blocks = nullptr;
}

int vcp_blocks_begin(vcp_ctx* ctx, const double* motor, int64_t n, double eps, int min_pts, int pts_in_cell,
                     int small_max, int32_t* rows, int32_t* cols, int64_t* nblocks, int64_t* m) {
  if (!ctx) return VCP_ERR_ARG;
  if (n > 0 && !motor) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(blocks_begin(ctx, nullptr, true, motor, nullptr, n, eps, min_pts, pts_in_cell, small_max, rows, cols, m));
  if (nblocks) *nblocks = ctx->blocks->nblocks;
  return VCP_OK;
}

int vcp_blocks_begin_keyed(vcp_ctx* ctx, const double* key_xy, const double* motor, int64_t n, double eps, int min_pts,
                           int pts_in_cell, int small_max, int32_t* rows, int32_t* cols, int64_t* nblocks, int64_t* m) {
  if (!ctx) return VCP_ERR_ARG;
  if (n > 0 && (!motor || !key_xy)) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(blocks_begin(ctx, nullptr, true, motor, key_xy, n, eps, min_pts, pts_in_cell, small_max, rows, cols, m));
  if (nblocks) *nblocks = ctx->blocks->nblocks;
  return VCP_OK;
}

int vcp_blocks_begin_keyed_dev(vcp_ctx* ctx, const double* d_key_xy, const double* d_motor, int64_t n, double eps,
                               int min_pts, int pts_in_cell, int small_max, int32_t* rows, int32_t* cols,
                               int64_t* nblocks, int64_t* m) {
  if (!ctx) return VCP_ERR_ARG;
  if (n > 0 && (!d_motor || !d_key_xy)) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(blocks_begin(ctx, d_motor, false, nullptr, d_key_xy, n, eps, min_pts, pts_in_cell, small_max, rows, cols, m));
  if (nblocks) *nblocks = ctx->blocks->nblocks;
  return VCP_OK;
}

int vcp_blocks_begin_dev(vcp_ctx* ctx, const double* d_motor, int64_t n, double eps, int min_pts, int pts_in_cell,
                         int small_max, int32_t* rows, int32_t* cols, int64_t* nblocks, int64_t* m) {
  if (!ctx) return VCP_ERR_ARG;
  if (n > 0 && !d_motor) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(blocks_begin(ctx, d_motor, false, nullptr, nullptr, n, eps, min_pts, pts_in_cell, small_max, rows, cols, m));
  if (nblocks) *nblocks = ctx->blocks->nblocks;
  return VCP_OK;
}

int vcp_blocks_share(vcp_ctx* ctx, int rank, int world, int32_t* block_lo, int32_t* block_hi, int64_t* pos_lo,
                     int64_t* pos_hi) {
  if (!ctx) return VCP_ERR_ARG;
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_begin has not run");
  if (world < 1 || rank < 0 || rank >= world) return vcp_fail(ctx, VCP_ERR_ARG, "rank/world");
  if (s->b_lo != 0 || s->b_hi != s->nblocks)
    return vcp_fail(ctx, VCP_ERR_ARG, "this context holds a share of the blocks already (vcp_blocks_build_dev)");
  // contiguous block ranges balanced on the point count (vcp_blocks_share_plan, multi.hip: the same arithmetic for the
  // multi-process ranks and for the device threads of vcp_dbscan_blocks_multi)
  std::vector<int64_t> cuts((size_t)world + 1);
  if (vcp_blocks_share_plan(s->h_blockstart.data(), s->nblocks, world, cuts.data()) != VCP_OK)
    return vcp_fail(ctx, VCP_ERR_ARG, "share plan");
  const int64_t lo = cuts[(size_t)rank], hi = cuts[(size_t)rank + 1];
  if (block_lo) *block_lo = (int32_t)lo;
  if (block_hi) *block_hi = (int32_t)hi;
  if (pos_lo) *pos_lo = s->h_blockstart[(size_t)lo];
  if (pos_hi) *pos_hi = s->h_blockstart[(size_t)hi];
  return VCP_OK;
}

int vcp_blocks_cluster_dev(vcp_ctx* ctx, int32_t block_lo, int32_t block_hi, int32_t* d_local, int64_t* evals) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  return blocks_cluster(ctx, block_lo, block_hi, d_local, evals);
}

int vcp_blocks_finish_dev(vcp_ctx* ctx, const int32_t* d_local, int64_t evals_blocks, int32_t* d_labels,
                          int32_t* d_block_of, int64_t* d_merge_order, int64_t* m_out, int32_t* kept, int32_t* del_sum,
                          int32_t* cluster_amount, int64_t* dist_evals) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_begin has not run");
  if (!d_labels || (s->m > 0 && !d_local)) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(blocks_finish(ctx, d_local, evals_blocks, d_labels, d_merge_order, kept, del_sum, cluster_amount, dist_evals));
  if (d_block_of)
    VCP_HIP(ctx, hipMemcpyAsync(d_block_of, s->blockof.p, (size_t)s->n * 4, hipMemcpyDeviceToDevice, ctx->stream));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (m_out) *m_out = s->m;
  return VCP_OK;
}

// ---- the pipeline with every stage sharded (include/vcp.h) ---------------------------------------------------------
int vcp_blocks_plan_dev(vcp_ctx* ctx, const double* d_key_xy, const double* d_motor, int64_t n, double eps, int min_pts,
                        int pts_in_cell, int small_max, int32_t* rows, int32_t* cols, int64_t* nblocks, int64_t* nsuper) {
  if (!ctx) return VCP_ERR_ARG;
  if (n > 0 && !d_motor) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  if (n < 0) return vcp_fail(ctx, VCP_ERR_ARG, "n < 0");
  if (n == 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "rawData.Min() on an empty list throws (FrmMain.cs:1224)");
  if (pts_in_cell <= 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "Take(0) then cell.Max() throws (FrmMain.cs:1255)");
  if (n >= 0x7FFFFFF0LL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "n beyond 32-bit indexing");
  VCP_TRY(vcp_bind(ctx));
  if (!ctx->blocks) ctx->blocks = new BlocksState();
  BlocksState* s = ctx->blocks;
  s->ready = false;
  s->n = n;
  s->eps = eps;
  s->min_pts = min_pts;
  s->small_max = small_max;
  s->motor_ptr = d_motor;
  VCP_TRY(vcp_blocks_plan(ctx, s, d_key_xy ? d_key_xy : d_motor, d_motor, n, pts_in_cell, true));
  if (rows) *rows = s->rows;
  if (cols) *cols = s->cols;
  if (nblocks) *nblocks = s->nblocks;
  if (nsuper) *nsuper = s->NS;
  return VCP_OK;
}

int vcp_blocks_plan_cuts(vcp_ctx* ctx, int world, int64_t* cuts) {
  if (!ctx) return VCP_ERR_ARG;
  BlocksState* s = ctx->blocks;
  if (!s || !s->planned || s->h_sbstart.empty()) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_plan_dev has not run");
  if (world < 1 || !cuts) return vcp_fail(ctx, VCP_ERR_ARG, "world / cuts");
  // shares of super-buckets balanced on the point count: rank r starts at the first super-bucket whose first point is
  // >= n * r / world -- the arithmetic of vcp_blocks_share_plan, one level up
  return vcp_blocks_share_plan(s->h_sbstart.data(), (int64_t)s->NS, world, cuts);
}

int vcp_blocks_build_dev(vcp_ctx* ctx, int64_t super_lo, int64_t super_hi, int32_t* block_lo, int32_t* block_hi, int64_t* m_loc,
                         int64_t* n_loc) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  BlocksState* s = ctx->blocks;
  if (!s || !s->planned || s->h_sbstart.empty()) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_plan_dev has not run");
  if (super_lo < 0 || super_hi > (int64_t)s->NS || super_lo > super_hi) return vcp_fail(ctx, VCP_ERR_ARG, "share of super-buckets");
  const uint32_t off0 = s->h_sbstart[(size_t)super_lo];
  const int64_t nl = (int64_t)s->h_sbstart[(size_t)super_hi] - (int64_t)off0;
  VCP_TRY(vcp_blocks_build(ctx, s, (uint32_t)super_lo, (uint32_t)super_hi, off0, nl));
  const int64_t nb1 = s->nblocks + 1;
  VCP_TRY(ens(ctx, s->gtwice, (size_t)nb1 * 4));
  VCP_TRY(ens(ctx, s->gnclus, (size_t)nb1 * 4));
  s->cov_ok = true;
  s->cov_lo = s->cov_hi = s->b_lo;
  s->totalC_acc = 0;
  s->ready = true;
  if (block_lo) *block_lo = (int32_t)s->b_lo;
  if (block_hi) *block_hi = (int32_t)s->b_hi;
  if (m_loc) *m_loc = s->m;
  if (n_loc) *n_loc = s->n_loc;
  return VCP_OK;
}

int vcp_blocks_finish_local_dev(vcp_ctx* ctx, const int32_t* d_local, int64_t info[8]) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_build_dev has not run");
  if (!info || (s->m > 0 && !d_local)) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(finish_local(ctx, d_local, true, false));
  info[0] = s->f_totalC;
  info[1] = s->f_kept;
  info[2] = s->f_err;
  info[3] = s->f_req;
  info[4] = s->f_nonempty;
  info[5] = s->f_last_nonzero;
  info[6] = s->m;
  info[7] = s->n_loc;
  return VCP_OK;
}

int vcp_blocks_finish_zero_dev(vcp_ctx* ctx, int zero_last, int64_t* z_count, int64_t* active_count) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready || !s->f_local) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_finish_local_dev has not run");
  VCP_TRY(finish_zero(ctx, true, zero_last, nullptr));
  if (z_count) *z_count = s->f_Z;
  if (active_count) *active_count = s->f_A;
  return VCP_OK;
}

int vcp_blocks_finish_zcoords_dev(vcp_ctx* ctx, int swap_xy, double* d_zcoords) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready || !s->f_local) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_finish_zero_dev has not run");
  if (s->f_A > 0 && !d_zcoords) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(finish_zcoords(ctx, d_zcoords, nullptr, swap_xy));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VCP_OK;
}

// for multi.hip (several devices from one process): the zero-list stage with the share's part of clusForMerge --
// d_merge_order [m_loc]: the non-zero entries in final order, then the share's zero list
int vcp_blocks_finish_zcoords_order(vcp_ctx* ctx, double* d_zcoords, int64_t* d_merge_order) {
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready || !s->f_local) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_finish_zero_dev has not run");
  VCP_TRY(finish_zcoords(ctx, d_zcoords, d_merge_order, 0));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VCP_OK;
}

int vcp_blocks_finish_pairs_dev(vcp_ctx* ctx, int32_t kept_offset, const int32_t* d_zlab, int64_t* d_pairs) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready || !s->f_local) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_finish_zero_dev has not run");
  if (s->n_loc > 0 && !d_pairs) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  if (s->f_A > 0 && !d_zlab) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  if (s->n_loc > 0)
    hipLaunchKernelGGL(k_pairs, dim3(nblk(s->n_loc)), dim3(BT), 0, ctx->stream, s->newlab.as<int32_t>(), d_zlab,
                       s->tmp2.as<uint32_t>(), s->bl.as<uint32_t>(), s->m, s->n_loc, kept_offset, d_pairs);
  VCP_HIP(ctx, hipGetLastError());
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VCP_OK;
}

int vcp_scatter_pairs_dev(vcp_ctx* ctx, const int64_t* d_pairs, int64_t count, int64_t n, int32_t* d_labels) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  if (count < 0 || n < 0) return vcp_fail(ctx, VCP_ERR_ARG, "count / n");
  if (count == 0) return VCP_OK;
  if (!d_pairs || !d_labels) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(vcp_ensure(ctx, ctx->b_misc, 256));
  uint32_t* bad = ctx->b_misc.as<uint32_t>();
  VCP_HIP(ctx, hipMemsetAsync(bad, 0, 4, ctx->stream));
  hipLaunchKernelGGL(k_scatter_pairs, dim3(nblk(count)), dim3(BT), 0, ctx->stream, d_pairs, count, n, d_labels, bad);
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(hp, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hp[0]) return vcp_fail(ctx, VCP_ERR_ARG, "a pair names an index beyond n");
  return VCP_OK;
}

int vcp_dbscan_blocks(vcp_ctx* ctx, const double* motor, int64_t n, double eps, int min_pts, int pts_in_cell,
                      int small_max, int32_t* labels, int32_t* block_of, int64_t* merge_order, int64_t* m_out,
                      int32_t* rows, int32_t* cols, int32_t* kept, int32_t* del_sum, int32_t* cluster_amount,
                      int64_t* dist_evals) {
  return vcp_dbscan_blocks_keyed(ctx, nullptr, motor, n, eps, min_pts, pts_in_cell, small_max, labels, block_of,
                                 merge_order, m_out, rows, cols, kept, del_sum, cluster_amount, dist_evals);
}

int vcp_dbscan_blocks_keyed(vcp_ctx* ctx, const double* key_xy, const double* motor, int64_t n, double eps, int min_pts,
                            int pts_in_cell, int small_max, int32_t* labels, int32_t* block_of, int64_t* merge_order,
                            int64_t* m_out, int32_t* rows, int32_t* cols, int32_t* kept, int32_t* del_sum,
                            int32_t* cluster_amount, int64_t* dist_evals) {
  if (!ctx) return VCP_ERR_ARG;
  if (n > 0 && (!motor || !labels)) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  int64_t m = 0, nblocks = 0;
  if (key_xy) VCP_TRY(vcp_blocks_begin_keyed(ctx, key_xy, motor, n, eps, min_pts, pts_in_cell, small_max, rows, cols, &nblocks, &m));
  else VCP_TRY(vcp_blocks_begin(ctx, motor, n, eps, min_pts, pts_in_cell, small_max, rows, cols, &nblocks, &m));
  hipStream_t st = ctx->stream;
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, (size_t)(m + 1) * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out3, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)(m + 1) * 8));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in3, (size_t)n * 4));
  int64_t ev = 0;
  VCP_TRY(blocks_cluster(ctx, 0, -1, ctx->b_out0.as<int32_t>(), &ev));
  VCP_TRY(vcp_blocks_finish_dev(ctx, ctx->b_out0.as<int32_t>(), ev, ctx->b_out3.as<int32_t>(), ctx->b_in3.as<int32_t>(),
                                merge_order ? ctx->b_in0.as<int64_t>() : nullptr, m_out, kept, del_sum, cluster_amount,
                                dist_evals));
  VCP_HIP(ctx, hipMemcpyAsync(labels, ctx->b_out3.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (block_of) VCP_HIP(ctx, hipMemcpyAsync(block_of, ctx->b_in3.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (merge_order && m > 0)
    VCP_HIP(ctx, hipMemcpyAsync(merge_order, ctx->b_in0.p, (size_t)m * 8, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  return VCP_OK;
}

}  // extern "C"
