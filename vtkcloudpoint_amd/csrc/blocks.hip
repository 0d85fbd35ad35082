// blocks.hip -- the reference's block-partitioned clustering ("v2.0 multithread") on MI355X.
//
//   begin   = MainForm.getClusterFromMotor, FrmMain.cs:1214-1291: bounds, stable sort by
//             max(x-xmin, y-ymin), first ptsInCell points -> block size, (lo,hi] rectangle blocks
//             (Tools.getListByScale2, BaseClass/Tools.cs:510-513), block-major list
//   cluster = StartCode, FrmMain.cs:2782-2794: one DBImproved(cf=0) per block -- here ONE grouped launch of
//             the DBSCAN engine over a contiguous range of blocks (the unit of multi-GPU sharding)
//   finish  = CompleteWork3, FrmMain.cs:1442-1520: per block stable order by local id, global renumber,
//             demotion of clusters of <= small_max points (with the reference's clusLen quirks), one
//             global DBImproved over all noise with cf preset, final clusForMerge order
//
// Declared deviations from the C# (same as the oracle): List.Sort's unstable tie order is
// replaced by a stable order; a block-0 point is never also filed under a rectangle; clusterSum is
// summed deterministically.  Sorting uses rocPRIM's stable LSD radix sort.
#include <string.h>  // rocprim's texture_cache_iterator.hpp calls ::memset without including it

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dbscan_engine.hpp"

struct BlocksState {
  int64_t n = 0, m = 0;
  int32_t rows = 0, cols = 0;
  int64_t nblocks = 0;
  double eps = 0;
  int min_pts = 0, small_max = 3, take = 0;
  double x_Min = 0, x_Max = 0, y_Min = 0, y_Max = 0, cell_x = 0, cell_y = 0;
  const double* motor_ptr = nullptr;  // the cloud on the device: our upload (host entry points) or the caller's array
  DevBuf motor, pkey, orand, raw, blockof, bl, motor_bm, blockstart, gtwice, gnclus, tmp0, tmp1, tmp2, tmp3, sorttmp,
      blk_t, csize, cstart, kb, zb, keep, order, newlab, zflag, zlist, zcoords, zlab, misc;
  std::vector<uint32_t> h_blockstart, h_big;
  DevBuf biglist;     // blocks of more than BIG_BLOCK points (k_block_order<16>), found on the host at begin
  uint32_t nbig = 0;
  bool ready = false;
};

namespace {
constexpr int BT = 256;
constexpr uint32_t NONE32 = 0xFFFFFFFFu;

int ens(vcp_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.cap >= bytes) return VCP_OK;
  if (b.p) {
    VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VCP_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    return vcp_fail(ctx, VCP_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  b.cap = want;
  return VCP_OK;
}

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

__device__ __forceinline__ double wmin(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmin(v, __shfl_down(v, d, 64));
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_down(v, d, 64));
  return v;
}

// out[0..3] = xmin, xmax, ymin, ymax over idx[0..cnt) (idx null = all), out[4] = #non-finite coordinates
__global__ __launch_bounds__(BT) void k_minmax2(const double* __restrict__ motor, const uint32_t* __restrict__ idx,
                                               int64_t cnt, double* __restrict__ out) {
  // single block
  double xmn = INFINITY, xmx = -INFINITY, ymn = INFINITY, ymx = -INFINITY, bad = 0;
  for (int64_t t = threadIdx.x; t < cnt; t += BT) {
    int64_t i = idx ? idx[t] : t;
    double x = motor[2 * i], y = motor[2 * i + 1];
    if (!isfinite(x) || !isfinite(y)) bad += 1.0;
    xmn = fmin(xmn, x);
    xmx = fmax(xmx, x);
    ymn = fmin(ymn, y);
    ymx = fmax(ymx, y);
  }
  __shared__ double sm[BT / 64][5];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double a = wmin(xmn), b = wmax(xmx), c = wmin(ymn), d = wmax(ymx), e = bad;
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) e += __shfl_down(e, k, 64);
  if (lane == 0) {
    sm[w][0] = a;
    sm[w][1] = b;
    sm[w][2] = c;
    sm[w][3] = d;
    sm[w][4] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < BT / 64; k++) {
      sm[0][0] = fmin(sm[0][0], sm[k][0]);
      sm[0][1] = fmax(sm[0][1], sm[k][1]);
      sm[0][2] = fmin(sm[0][2], sm[k][2]);
      sm[0][3] = fmax(sm[0][3], sm[k][3]);
      sm[0][4] += sm[k][4];
    }
    for (int k = 0; k < 5; k++) out[k] = sm[0][k];
  }
}

// multi-block version writing per-block partials [nb][5]
__global__ __launch_bounds__(BT) void k_minmax2_part(const double* __restrict__ motor, int64_t n, double* __restrict__ part) {
  double xmn = INFINITY, xmx = -INFINITY, ymn = INFINITY, ymx = -INFINITY, bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * BT + threadIdx.x; i < n; i += (int64_t)gridDim.x * BT) {
    double2 v = *reinterpret_cast<const double2*>(motor + 2 * i);
    if (!isfinite(v.x) || !isfinite(v.y)) bad += 1.0;
    xmn = fmin(xmn, v.x);
    xmx = fmax(xmx, v.x);
    ymn = fmin(ymn, v.y);
    ymx = fmax(ymx, v.y);
  }
  __shared__ double sm[BT / 64][5];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double a = wmin(xmn), b = wmax(xmx), c = wmin(ymn), d = wmax(ymx), e = bad;
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) e += __shfl_down(e, k, 64);
  if (lane == 0) {
    sm[w][0] = a;
    sm[w][1] = b;
    sm[w][2] = c;
    sm[w][3] = d;
    sm[w][4] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < BT / 64; k++) {
      sm[0][0] = fmin(sm[0][0], sm[k][0]);
      sm[0][1] = fmax(sm[0][1], sm[k][1]);
      sm[0][2] = fmin(sm[0][2], sm[k][2]);
      sm[0][3] = fmax(sm[0][3], sm[k][3]);
      sm[0][4] += sm[k][4];
    }
    for (int k = 0; k < 5; k++) part[(size_t)blockIdx.x * 5 + k] = sm[0][k];
  }
}
__global__ __launch_bounds__(BT) void k_minmax2_final(const double* __restrict__ part, int nb, double* __restrict__ out) {
  double a = INFINITY, b = -INFINITY, c = INFINITY, d = -INFINITY, e = 0;
  for (int k = threadIdx.x; k < nb; k += BT) {
    a = fmin(a, part[k * 5]);
    b = fmax(b, part[k * 5 + 1]);
    c = fmin(c, part[k * 5 + 2]);
    d = fmax(d, part[k * 5 + 3]);
    e += part[k * 5 + 4];
  }
  __shared__ double sm[BT / 64][5];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  a = wmin(a);
  b = wmax(b);
  c = wmin(c);
  d = wmax(d);
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) e += __shfl_down(e, k, 64);
  if (lane == 0) {
    sm[w][0] = a;
    sm[w][1] = b;
    sm[w][2] = c;
    sm[w][3] = d;
    sm[w][4] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < BT / 64; k++) {
      sm[0][0] = fmin(sm[0][0], sm[k][0]);
      sm[0][1] = fmax(sm[0][1], sm[k][1]);
      sm[0][2] = fmin(sm[0][2], sm[k][2]);
      sm[0][3] = fmax(sm[0][3], sm[k][3]);
      sm[0][4] += sm[k][4];
    }
    for (int k = 0; k < 5; k++) out[k] = sm[0][k];
  }
}

// FrmMain.cs:1231-1232: d = Math.Max(x - x_Min, y - y_Min); non-negative, so the IEEE bit pattern orders it
// Also reduces, per workgroup and into 32 slots, the OR and the AND of all keys: bits on which every key agrees cannot
// change the order, so the radix sort runs over the varying bit range only (coordinates on a 2^-10 grid below 2^10: 20
// significant bits -> 3 passes instead of the 8 of a 64-bit key).
__global__ __launch_bounds__(BT) void k_sortkey(const double* __restrict__ motor, int64_t n, double x_Min, double y_Min,
                                               uint64_t* __restrict__ key, uint32_t* __restrict__ idx,
                                               unsigned long long* __restrict__ orand) {
  unsigned long long ko = 0ull, ka = ~0ull;
  for (int64_t i = (int64_t)blockIdx.x * BT + threadIdx.x; i < n; i += (int64_t)gridDim.x * BT) {  // <= 2048 workgroups:
    double2 v = *reinterpret_cast<const double2*>(motor + 2 * i);                                    // 2 atomics each
    double a = v.x - x_Min, b = v.y - y_Min;
    double d = a > b ? a : b;  // Math.Max on finite values
    const uint64_t k = (uint64_t)__double_as_longlong(d + 0.0);
    key[i] = k;
    idx[i] = (uint32_t)i;
    ko |= k;
    ka &= k;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    ko |= __shfl_xor(ko, d, 64);
    ka &= __shfl_xor(ka, d, 64);
  }
  __shared__ unsigned long long so[BT / 64], sa[BT / 64];
  if ((threadIdx.x & 63) == 0) {
    so[threadIdx.x >> 6] = ko;
    sa[threadIdx.x >> 6] = ka;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < BT / 64; k++) {
      so[0] |= so[k];
      sa[0] &= sa[k];
    }
    atomicOr(&orand[blockIdx.x & 31], so[0]);
    atomicAnd(&orand[32 + (blockIdx.x & 31)], sa[0]);
  }
}

struct PartP {
  double x_Min, x_Max, y_Min, y_Max, cell_x, cell_y;
  int rows, cols, take;
};

// unique q with lo(q) < v <= hi(q) (Tools.getListByScale2: strict > on the low edge, <= on the high edge;
// last row / column stretched to the max), or -1.  lo/hi are evaluated exactly as FrmMain.cs:1262-1285 does.
__device__ __forceinline__ int find_axis(double v, double vmin, double vmax, double cellw, int cnt) {
  double g = (v - vmin) / cellw;
  long long q0 = isfinite(g) ? (long long)floor(g) : 0;
  for (long long q = q0 - 2; q <= q0 + 2; q++) {
    if (q < 0 || q >= cnt) continue;
    double lo = vmin + (double)(int)q * cellw;
    double hi = (q == cnt - 1) ? vmax : vmin + (double)((int)q + 1) * cellw;
    if (v > lo && v <= hi) return (int)q;
  }
  {
    int q = cnt - 1;
    double lo = vmin + (double)q * cellw;
    if (v > lo && v <= vmax) return q;
  }
  return -1;
}

// The first block is rawData.Take(ptsInCell) of the list sorted by d = max(x - x_Min, y - y_Min) (stable: ties keep the
// input order), i.e. exactly the points whose (d, index) is <= that of the take-th element (key_T, idx_T): no rank array.
__global__ __launch_bounds__(BT) void k_block_of(const double* __restrict__ motor, int64_t n, PartP P, uint64_t key_T,
                                                uint32_t idx_T, int32_t* __restrict__ blockof) {
  int64_t i = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (i >= n) return;
  int32_t b = -1;
  const double2 v = *reinterpret_cast<const double2*>(motor + 2 * i);
  const double da = v.x - P.x_Min, db = v.y - P.y_Min;
  const double d = da > db ? da : db;
  const uint64_t k = (uint64_t)__double_as_longlong(d + 0.0);  // k_sortkey's key
  if (k < key_T || (k == key_T && (uint32_t)i <= idx_T)) {
    b = 0;  // cells[0] = rawData.Take(ptsInCell), FrmMain.cs:1254,1260
  } else {
    int q = find_axis(v.x, P.x_Min, P.x_Max, P.cell_x, P.cols);
    int p = find_axis(v.y, P.y_Min, P.y_Max, P.cell_y, P.rows);
    if (p >= 0 && q >= 0) {
      long long index = (long long)p * P.cols + q;
      if (index != 0) b = (int32_t)index;  // rectangle 0 is skipped, FrmMain.cs:1266
    }
  }
  blockof[i] = b;
}

// keys for the block-major list: block id of raw[t] (dropped -> nblocks), values raw[t]
__global__ __launch_bounds__(BT) void k_blockkey(const uint32_t* __restrict__ raw, const int32_t* __restrict__ blockof,
                                                int64_t n, uint32_t nblocks, uint32_t* __restrict__ key) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t >= n) return;
  int32_t b = blockof[raw[t]];
  key[t] = b < 0 ? nblocks : (uint32_t)b;
}

// mark[k] = (last position of key k) + 1 in the sorted key list; an exclusive max-scan of the marks gives the
// first position of every key (no per-point atomics: global atomics execute at the memory side on this part,
// and 10 M adds into 27 k counters took 1.6 ms)
__global__ __launch_bounds__(BT) void k_mark_key_ends(const uint32_t* __restrict__ skey, int64_t n,
                                                     uint32_t* __restrict__ mark) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t >= n) return;
  const uint32_t k = skey[t];
  if (t == n - 1 || skey[t + 1] != k) mark[k] = (uint32_t)t + 1u;
}

// the cloud in block-major order: the per-block clustering then reads and writes by block-major position (the
// engine's gather becomes spatially coherent, its labels ARE the block-local ids in list order)
__global__ __launch_bounds__(BT) void k_gather_motor(const double* __restrict__ motor, const uint32_t* __restrict__ bl,
                                                    int64_t m, double* __restrict__ motor_bm) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t < m) *reinterpret_cast<double2*>(motor_bm + 2 * t) = *reinterpret_cast<const double2*>(motor + 2 * (int64_t)bl[t]);
}

// ---- finish --------------------------------------------------------------------------------------
// (One lane per position with the lanes of a block combined by ballot and ONE atomicMax / atomicAdd per wave and block
// was measured: 443 us against 96 -- the per-block words of neighbouring blocks share cache lines and the atomics of
// 156 k waves serialise on them.)
// per block: K_b = max local id, Z_b = number of noise points.  One wave per block walks the block's slice of
// the block-major label list (coalesced) and reduces in registers: no atomics.
// NW = 1: one wave per block (blocks of up to BIG_BLOCK positions); NW = 16: one workgroup per block of the host's list of
// large blocks, like k_block_order below.
constexpr uint32_t BIG_BLOCK = 1024;
template <int NW>
__global__ __launch_bounds__(NW == 1 ? BT : 64 * NW) void k_block_stats(const int32_t* __restrict__ local,
                                                                       const uint32_t* __restrict__ blockstart,
                                                                       int64_t nblocks, const uint32_t* __restrict__ biglist,
                                                                       uint32_t* __restrict__ kb, uint32_t* __restrict__ zb,
                                                                       uint32_t* __restrict__ kmax_slots) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t b;
  if (NW == 1) {
    b = (int64_t)blockIdx.x * (BT / 64) + w;
    if (b >= nblocks) return;
  } else {
    b = biglist[blockIdx.x];
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
    // largest local id over all blocks (32 slots: a single word would serialise 27 k atomics): the key width of the
    // final (block, local id) sort
    if (K > kmax_slots[blockIdx.x & 31]) atomicMax(&kmax_slots[blockIdx.x & 31], K);
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
// positions (by local id, then by block) when no block has more than CS_CAP ids.
//   NW = 1: one wave per block, four blocks per workgroup: blocks of up to BIG_BLOCK positions (a block holds
//           ~ptsInCell points);
//   NW = 16: one workgroup per block of the host's list of large blocks (the heart of a blob can put 10^4 points in
//           one rectangle; a single wave walking it set the kernel time: 415 us): every wave reads the whole block but
//           places only the ids congruent to its number, so each cursor has one owner and the order stays stable.
template <int NW>
__global__ __launch_bounds__(NW == 1 ? BT : 64 * NW) void k_block_order(const int32_t* __restrict__ local,
                                                                       const uint32_t* __restrict__ blockstart,
                                                                       int64_t nblocks, const uint32_t* __restrict__ kb,
                                                                       const uint32_t* __restrict__ biglist,
                                                                       const uint32_t* __restrict__ cstart,
                                                                       uint32_t* __restrict__ csize,
                                                                       uint32_t* __restrict__ order) {
  constexpr int NG = NW == 1 ? BT / 64 : 1;  // blocks per workgroup
  __shared__ uint32_t cnt[NG][CS_CAP + 1];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = NW == 1 ? w : 0;       // which of the workgroup's blocks
  const int me = NW == 1 ? 0 : w;      // which ids this wave places (id % NW)
  int64_t b;
  if (NW == 1) {
    b = (int64_t)blockIdx.x * NG + w;
    if (b >= nblocks) return;  // whole waves leave together; no workgroup barrier below for NW == 1
  } else {
    b = biglist[blockIdx.x];
  }
  const uint32_t s0 = blockstart[b], s1 = blockstart[b + 1];
  if (NW == 1 && (s0 == s1 || s1 - s0 > BIG_BLOCK)) return;
  const uint32_t K = kb[b];  // ids 0..K
  const uint32_t nthr = 64 * NW, tid = NW == 1 ? lane : threadIdx.x;
  for (uint32_t k = tid; k <= K; k += nthr) cnt[g][k] = 0;
  if (NW == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
  for (uint32_t t = s0 + tid; t < s1; t += nthr) atomicAdd(&cnt[g][local[t]], 1u);
  if (NW == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
  {  // the counts of ids 1..K are the cluster sizes CompleteWork3's demotion rule needs (k_keep)
    const uint32_t c0 = cstart[b];
    for (uint32_t k = 1 + tid; k <= K; k += nthr) csize[c0 + k - 1] = cnt[g][k];
  }
  if (NW == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
  if (w == 0 || NW == 1) {
    // exclusive scan of cnt[0..K] by one wave, 64 entries per trip
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
  if (NW == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
  // NW > 1: the ids come through LDS in tiles loaded by the whole workgroup (a wave walking a large block chunk by chunk
  // would wait for one global load per chunk: 400 dependent loads for 25 k points)
  constexpr uint32_t TILE = NW == 1 ? 64 : 8192;
  __shared__ uint16_t ids[NW == 1 ? 1 : TILE];
  for (uint32_t tile0 = s0; tile0 < s1; tile0 += TILE) {
    const uint32_t tend = min(tile0 + TILE, s1);
    if (NW > 1) {
      __syncthreads();  // the previous tile has been consumed
      for (uint32_t t = tile0 + tid; t < tend; t += nthr) ids[t - tile0] = (uint16_t)local[t];  // ids <= CS_CAP
      __syncthreads();
    }
    for (uint32_t t0 = tile0; t0 < tend; t0 += 64) {
      const uint32_t t = t0 + lane;
      const uint32_t id = t < tend ? (NW == 1 ? (uint32_t)local[t] : (uint32_t)ids[t - tile0]) : 0xFFFFFFFFu;
      const bool mine = t < tend && (NW == 1 || id % NW == (uint32_t)me);
      unsigned long long todo = __ballot(mine);
      while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const uint32_t cur = (uint32_t)__shfl((int)id, first, 64);
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
}
// keep[c] for cluster entry c = cstart[b] + k - 1 (FrmMain.cs:1479-1495): a cluster is demoted when the next
// id shows up and clusLen <= small_max; clusLen over-counts the first cluster of a block without noise by
// one (:1461-1465); the last cluster of a block is never checked.
__global__ __launch_bounds__(BT) void k_keep(int64_t nblocks, const uint32_t* __restrict__ cstart,
                                            const uint32_t* __restrict__ kb, const uint32_t* __restrict__ zb,
                                            const uint32_t* __restrict__ blockstart, const uint32_t* __restrict__ csize,
                                            int small_max, uint32_t* __restrict__ keep, uint32_t* __restrict__ victim_of,
                                            uint32_t* __restrict__ err) {
  int64_t b = (int64_t)blockIdx.x * BT + threadIdx.x;
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
      if (pb < 0) atomicAdd(err, 1u);  // clusForMerge[-1]: ArgumentOutOfRangeException
      else victim_of[b] = (uint32_t)pb;
    }
  }
}
__global__ __launch_bounds__(BT) void k_newlab(const int32_t* __restrict__ local, const uint32_t* __restrict__ blk_t,
                                              int64_t m, const uint32_t* __restrict__ cstart,
                                              const uint32_t* __restrict__ keep, const uint32_t* __restrict__ keeprank,
                                              int32_t* __restrict__ newlab) {
  int64_t t = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (t >= m) return;
  int32_t l = local[t];
  int32_t out = 0;
  if (l > 0) {
    uint32_t c = cstart[blk_t[t]] + (uint32_t)l - 1u;
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
__global__ __launch_bounds__(BT) void k_zero_flag(const int32_t* __restrict__ newlab, const uint32_t* __restrict__ order,
                                                 int64_t m, uint32_t* __restrict__ zflag) {
  int64_t u = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (u < m) zflag[u] = newlab[order[u]] == 0 ? 1u : 0u;
}
// merge_order = non-zero entries in final order, then the zero list (FrmMain.cs:1510-1520)
__global__ __launch_bounds__(BT) void k_compact(const uint32_t* __restrict__ zflag_scan, const int32_t* __restrict__ newlab,
                                               const uint32_t* __restrict__ order, const uint32_t* __restrict__ bl,
                                               const double* __restrict__ motor_bm, int64_t m, uint32_t Z,
                                               uint32_t* __restrict__ zrank, double* __restrict__ zcoords,
                                               int64_t* __restrict__ merge_order) {
  int64_t u = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (u >= m) return;
  uint32_t t = order[u];
  uint32_t zr = zflag_scan[u];
  if (newlab[t] == 0) {
    zrank[t] = zr;  // where the noise pass will leave this point's label
    // the coordinates in block-major order: t stays inside the point's block, the original index does not
    *reinterpret_cast<double2*>(zcoords + 2 * (size_t)zr) = *reinterpret_cast<const double2*>(motor_bm + 2 * (size_t)t);
    if (merge_order) merge_order[(m - Z) + zr] = (int64_t)bl[t];
  } else if (merge_order) {
    merge_order[u - zr] = (int64_t)bl[t];
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
    if (v == 0 && zlab) v = zlab[zrank[t]];
  }
  labels[bl[t]] = v;
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
    // a non-finite motor coordinate would reach DBImproved only; the partition's own check below covers the keys
    const int rbm = (int)vcp_blocks(n, BT, 1024);
    VCP_TRY(ens(ctx, s->misc, (size_t)(rbm * 5 + 64) * 8));
    double* partm = s->misc.as<double>();
    double* outm = partm + (size_t)rbm * 5;
    hipLaunchKernelGGL(k_minmax2_part, dim3(rbm), dim3(BT), 0, st, motor_own, n, partm);
    hipLaunchKernelGGL(k_minmax2_final, dim3(1), dim3(BT), 0, st, partm, rbm, outm);
    double* hm = reinterpret_cast<double*>(ctx->pinned);
    VCP_HIP(ctx, hipMemcpyAsync(hm, outm, 5 * 8, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    if (hm[4] != 0.0) return vcp_fail(ctx, VCP_ERR_ARG, "non-finite motor coordinates");
  }
  const double* motor = key_in ? (from_host ? s->pkey.as<double>() : key_in) : motor_own;  // what the partition reads
  // bounds (FrmMain.cs:1224-1227) and the finiteness check
  const int rb = (int)vcp_blocks(n, BT, 1024);
  VCP_TRY(ens(ctx, s->misc, (size_t)(rb * 5 + 64) * 8));
  double* part = s->misc.as<double>();
  double* out = part + (size_t)rb * 5;
  hipLaunchKernelGGL(k_minmax2_part, dim3(rb), dim3(BT), 0, st, motor, n, part);
  hipLaunchKernelGGL(k_minmax2_final, dim3(1), dim3(BT), 0, st, part, rb, out);
  double* h = reinterpret_cast<double*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(h, out, 5 * 8, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (h[4] != 0.0) return vcp_fail(ctx, VCP_ERR_ARG, "non-finite partition coordinates");
  s->x_Min = h[0];
  s->x_Max = h[1];
  s->y_Min = h[2];
  s->y_Max = h[3];
  // stable sort by max(x - x_Min, y - y_Min) (FrmMain.cs:1229-1251; ties keep the input order)
  VCP_TRY(ens(ctx, s->tmp0, (size_t)n * 8));
  VCP_TRY(ens(ctx, s->tmp1, (size_t)n * 8));
  VCP_TRY(ens(ctx, s->tmp2, (size_t)n * 4));
  VCP_TRY(ens(ctx, s->raw, (size_t)n * 4));
  VCP_TRY(ens(ctx, s->orand, 64 * 8));
  unsigned long long* orand = s->orand.as<unsigned long long>();
  VCP_HIP(ctx, hipMemsetAsync(orand, 0, 32 * 8, st));
  VCP_HIP(ctx, hipMemsetAsync(orand + 32, 0xFF, 32 * 8, st));
  hipLaunchKernelGGL(k_sortkey, dim3(vcp_blocks(n, BT, 2048)), dim3(BT), 0, st, motor, n, s->x_Min, s->y_Min, s->tmp0.as<uint64_t>(),
                     s->tmp2.as<uint32_t>(), orand);
  int bit_lo = 0, bit_hi = 64;
  {
    unsigned long long* ho = reinterpret_cast<unsigned long long*>(ctx->pinned) + 256;
    VCP_HIP(ctx, hipMemcpyAsync(ho, orand, 64 * 8, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    unsigned long long o = 0ull, a = ~0ull;
    for (int k = 0; k < 32; k++) {
      o |= ho[k];
      a &= ho[32 + k];
    }
    const unsigned long long varying = o & ~a;
    if (varying == 0ull) {
      bit_lo = 0;
      bit_hi = 1;  // all keys equal: one pass keeps the input order
    } else {
      bit_lo = __builtin_ctzll(varying);
      bit_hi = 64 - __builtin_clzll(varying);
    }
  }
  {
    size_t tb = 0;
    VCP_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, s->tmp0.as<uint64_t>(), s->tmp1.as<uint64_t>(),
                                           s->tmp2.as<uint32_t>(), s->raw.as<uint32_t>(), (size_t)n, bit_lo, bit_hi, st));
    VCP_TRY(ens(ctx, s->sorttmp, tb + 64));
    VCP_HIP(ctx, rocprim::radix_sort_pairs(s->sorttmp.p, tb, s->tmp0.as<uint64_t>(), s->tmp1.as<uint64_t>(),
                                           s->tmp2.as<uint32_t>(), s->raw.as<uint32_t>(), (size_t)n, bit_lo, bit_hi, st));
  }
  // first block -> block size (FrmMain.cs:1253-1258); its last element's (key, index) tells every point whether it is in
  s->take = (int)std::min<int64_t>(pts_in_cell, n);
  hipLaunchKernelGGL(k_minmax2, dim3(1), dim3(BT), 0, st, motor, s->raw.as<uint32_t>(), (int64_t)s->take, out);
  VCP_HIP(ctx, hipMemcpyAsync(h, out, 5 * 8, hipMemcpyDeviceToHost, st));
  uint64_t* h_keyT = reinterpret_cast<uint64_t*>(ctx->pinned) + 16;
  uint32_t* h_idxT = reinterpret_cast<uint32_t*>(h_keyT + 1);
  VCP_HIP(ctx, hipMemcpyAsync(h_keyT, s->tmp1.as<uint64_t>() + (s->take - 1), 8, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(h_idxT, s->raw.as<uint32_t>() + (s->take - 1), 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  const uint64_t key_T = *h_keyT;
  const uint32_t idx_T = *h_idxT;
  s->cell_x = h[1] - s->x_Min;
  s->cell_y = h[3] - s->y_Min;
  const double fr = (s->y_Max - s->y_Min) / s->cell_y, fc = (s->x_Max - s->x_Min) / s->cell_x;
  if (!std::isfinite(fr) || !std::isfinite(fc))
    return vcp_fail(ctx, VCP_ERR_DEGENERATE, "first block has zero extent: rows/cols undefined (FrmMain.cs:1256-1259)");
  if (fr >= 2147483646.0 || fc >= 2147483646.0) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "rows/cols overflow int");
  s->rows = (int)fr + 1;
  s->cols = (int)fc + 1;
  s->nblocks = (int64_t)s->rows * s->cols;
  if (s->nblocks > ((int64_t)1 << 26)) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "%lld blocks", (long long)s->nblocks);
  if (rows_o) *rows_o = s->rows;
  if (cols_o) *cols_o = s->cols;
  // block of every point (FrmMain.cs:1259-1285, Tools.cs:510-513)
  PartP P{s->x_Min, s->x_Max, s->y_Min, s->y_Max, s->cell_x, s->cell_y, s->rows, s->cols, s->take};
  VCP_TRY(ens(ctx, s->blockof, (size_t)n * 4));
  hipLaunchKernelGGL(k_block_of, dim3(nblk(n)), dim3(BT), 0, st, motor, n, P, key_T, idx_T, s->blockof.as<int32_t>());
  // block-major list: stable sort of the list order by block id; dropped points go last
  const int64_t nb1 = s->nblocks + 1;
  VCP_TRY(ens(ctx, s->blockstart, (size_t)(nb1 + 1) * 4));
  VCP_TRY(ens(ctx, s->bl, (size_t)n * 4));
  VCP_TRY(ens(ctx, s->motor_bm, (size_t)n * 16));
  uint32_t* bkey = s->tmp2.as<uint32_t>();
  VCP_TRY(ens(ctx, s->blk_t, (size_t)(n + 1) * 4));
  uint32_t* blk_t = s->blk_t.as<uint32_t>();  // block id per block-major position (the sorted keys)
  hipLaunchKernelGGL(k_blockkey, dim3(nblk(n)), dim3(BT), 0, st, s->raw.as<uint32_t>(), s->blockof.as<int32_t>(), n,
                     (uint32_t)s->nblocks, bkey);
  VCP_TRY(sort_pairs(ctx, s, bkey, blk_t, s->raw.as<uint32_t>(), s->bl.as<uint32_t>(), (size_t)n,
                     bits_for((uint64_t)s->nblocks)));
  VCP_HIP(ctx, hipMemsetAsync(s->blockstart.p, 0, (size_t)(nb1 + 1) * 4, st));
  hipLaunchKernelGGL(k_mark_key_ends, dim3(nblk(n)), dim3(BT), 0, st, blk_t, n, s->blockstart.as<uint32_t>());
  VCP_TRY(vcp_exclusive_max_scan_u32(ctx, s->blockstart.as<uint32_t>(), s->blockstart.as<uint32_t>(), nb1 + 1, nullptr));
  hipLaunchKernelGGL(k_gather_motor, dim3(nblk(n)), dim3(BT), 0, st, motor_own, s->bl.as<uint32_t>(), n,
                     s->motor_bm.as<double>());
  VCP_HIP(ctx, hipGetLastError());
  s->h_blockstart.resize((size_t)nb1 + 1);
  VCP_HIP(ctx, hipMemcpyAsync(s->h_blockstart.data(), s->blockstart.p, (size_t)(nb1 + 1) * 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  s->m = s->h_blockstart[(size_t)s->nblocks];
  if (m_o) *m_o = s->m;
  {
    std::vector<uint32_t>& big = s->h_big;  // a member: the copy below is asynchronous
    big.clear();
    for (int64_t b = 0; b < s->nblocks; b++)
      if (s->h_blockstart[(size_t)b + 1] - s->h_blockstart[(size_t)b] > BIG_BLOCK) big.push_back((uint32_t)b);
    s->nbig = (uint32_t)big.size();
    if (s->nbig) {
      VCP_TRY(ens(ctx, s->biglist, big.size() * 4));
      VCP_HIP(ctx, hipMemcpyAsync(s->biglist.p, big.data(), big.size() * 4, hipMemcpyHostToDevice, st));
    }
  }
  VCP_TRY(ens(ctx, s->gtwice, (size_t)nb1 * 4));
  VCP_TRY(ens(ctx, s->gnclus, (size_t)nb1 * 4));
  s->ready = true;
  return VCP_OK;
}

int blocks_cluster(vcp_ctx* ctx, int32_t lo, int32_t hi, int32_t* d_local, int64_t* evals_o) {
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_begin has not run");
  if (hi < 0) hi = (int32_t)s->nblocks;
  if (lo < 0 || hi > s->nblocks || lo > hi) return vcp_fail(ctx, VCP_ERR_ARG, "block range");
  if (evals_o) *evals_o = 0;
  if (lo == hi) return VCP_OK;
  if (s->m == 0) return VCP_OK;
  DbscanExt ext;
  ext.d_group = reinterpret_cast<const int32_t*>(s->blk_t.as<uint32_t>());  // block id per block-major position
  ext.d_ord = nullptr;                                                        // list position = input index
  ext.d_groupstart = s->blockstart.as<uint32_t>();
  ext.G = (int32_t)s->nblocks;
  ext.only_lo = lo;
  ext.only_hi = hi;
  ext.d_group_twice = s->gtwice.as<uint32_t>();
  ext.d_group_nclus = s->gnclus.as<uint32_t>();
  int64_t ev = 0;
  int32_t cf = 0;
  // input = the m points that fell in a block, in block-major order; positions outside [lo, hi)'s slice get 0
  VCP_TRY(vcp_dbscan_engine(ctx, s->motor_bm.as<double>(), s->m, 2, VCP_L1_2D, s->eps, s->min_pts, 0, nullptr, d_local,
                            nullptr, nullptr, &cf, &ev, &ext));
  VCP_HIP(ctx, hipGetLastError());
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (evals_o) *evals_o = ev;
  return VCP_OK;
}

int blocks_finish(vcp_ctx* ctx, const int32_t* d_local, int64_t evals_blocks, int32_t* d_labels, int64_t* d_merge_order,
                  int32_t* kept_o, int32_t* del_o, int32_t* ca_o, int64_t* evals_o) {
  BlocksState* s = ctx->blocks;
  if (!s || !s->ready) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_blocks_begin has not run");
  hipStream_t st = ctx->stream;
  const int64_t n = s->n, m = s->m, nb = s->nblocks;
  const uint32_t* blockstart = s->blockstart.as<uint32_t>();
  VCP_TRY(ens(ctx, s->kb, (size_t)(nb + 2) * 4));
  VCP_TRY(ens(ctx, s->zb, (size_t)(nb + 2) * 4));
  VCP_TRY(ens(ctx, s->cstart, (size_t)(nb + 2) * 4));
  const uint32_t* blk_t = s->blk_t.as<uint32_t>();
  uint32_t* kb = s->kb.as<uint32_t>();
  uint32_t* zb = s->zb.as<uint32_t>();
  uint32_t* cstart = s->cstart.as<uint32_t>();
  uint32_t* dmisc = s->misc.as<uint32_t>();  // [0] total clusters, [1] kept, [2] err, [3] Z
  const unsigned nbw = (unsigned)((nb + BT / 64 - 1) / (BT / 64));  // one wave per block
  VCP_HIP(ctx, hipMemsetAsync(dmisc, 0, 64 * 4, st));  // [8..40): slots of the largest local id
  VCP_HIP(ctx, hipMemsetAsync(kb + nb, 0, 8, st));  // the scan reads kb[nb]
  hipLaunchKernelGGL(k_block_stats<1>, dim3(nbw), dim3(BT), 0, st, d_local, blockstart, nb, nullptr, kb, zb, dmisc + 8);
  if (s->nbig)
    hipLaunchKernelGGL(k_block_stats<16>, dim3(s->nbig), dim3(1024), 0, st, d_local, blockstart, nb,
                       s->biglist.as<uint32_t>(), kb, zb, dmisc + 8);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, kb, cstart, nb + 1, dmisc));  // cstart[nb] = total clusters
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(hp, dmisc, 40 * 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  const uint32_t totalC = hp[0];
  uint32_t maxK = 0;  // largest local cluster id in any block
  for (int k = 0; k < 32; k++) maxK = std::max(maxK, hp[8 + k]);
  VCP_TRY(ens(ctx, s->csize, (size_t)(totalC + 2) * 4));
  VCP_TRY(ens(ctx, s->keep, (size_t)(totalC + 2) * 4 * 2));
  VCP_TRY(ens(ctx, s->tmp3, (size_t)(nb + 2) * 4));
  uint32_t* csize = s->csize.as<uint32_t>();
  uint32_t* keep = s->keep.as<uint32_t>();
  uint32_t* keeprank = keep + (totalC + 2);
  uint32_t* victim_of = s->tmp3.as<uint32_t>();
  // final order inside a block: stable by local id -- a per-block counting sort (which also yields the cluster sizes),
  // or, when some block has more ids than its LDS table, the library-sort form: positions by local id, then by block
  VCP_TRY(ens(ctx, s->order, (size_t)(m + 1) * 4));
  VCP_TRY(ens(ctx, s->tmp0, (size_t)(m + 1) * 8));
  VCP_TRY(ens(ctx, s->tmp1, (size_t)(m + 1) * 8));
  VCP_TRY(ens(ctx, s->tmp2, (size_t)(m + 1) * 4));
  uint32_t* iota = s->tmp2.as<uint32_t>();
  uint32_t* k1 = s->tmp0.as<uint32_t>();
  uint32_t* k1o = k1 + (m + 1);
  uint32_t* v1o = s->tmp1.as<uint32_t>();
  uint32_t* k2 = v1o + (m + 1);
  uint32_t* order = s->order.as<uint32_t>();
  const bool order_by_sort = getenv("VCP_BLOCKS_ORDER_SORT") != nullptr;  // test switch: the library-sort form
  VCP_HIP(ctx, hipMemsetAsync(keep, 0, (size_t)(totalC + 2) * 4 * 2, st));
  if (m > 0 && maxK <= (uint32_t)CS_CAP && !order_by_sort) {
    hipLaunchKernelGGL(k_block_order<1>, dim3(nbw), dim3(BT), 0, st, d_local, blockstart, nb, kb, nullptr, cstart, csize,
                       order);
    if (s->nbig)
      hipLaunchKernelGGL(k_block_order<16>, dim3(s->nbig), dim3(1024), 0, st, d_local, blockstart, nb, kb,
                         s->biglist.as<uint32_t>(), cstart, csize, order);
  } else if (m > 0) {
    VCP_HIP(ctx, hipMemsetAsync(csize, 0, (size_t)(totalC + 2) * 4, st));
    hipLaunchKernelGGL(k_cluster_sizes, dim3(nbw), dim3(BT), 0, st, d_local, blockstart, nb, kb, cstart, csize);
    hipLaunchKernelGGL(k_iota, dim3(nblk(m)), dim3(BT), 0, st, iota, m);
    VCP_HIP(ctx, hipMemcpyAsync(k1, d_local, (size_t)m * 4, hipMemcpyDeviceToDevice, st));
    VCP_TRY(sort_pairs(ctx, s, k1, k1o, iota, v1o, (size_t)m, bits_for(maxK)));
    hipLaunchKernelGGL(k_gather_u32, dim3(nblk(m)), dim3(BT), 0, st, blk_t, v1o, m, k2);
    VCP_TRY(sort_pairs(ctx, s, k2, k1o, v1o, order, (size_t)m, bits_for((uint64_t)nb)));
  }
  hipLaunchKernelGGL(k_keep, dim3(nblk(nb)), dim3(BT), 0, st, nb, cstart, kb, zb, blockstart, csize, s->small_max, keep,
                     victim_of, dmisc + 2);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, keep, keeprank, (int64_t)totalC + 1, dmisc + 1));
  VCP_TRY(ens(ctx, s->newlab, (size_t)(m + 1) * 4));
  int32_t* newlab = s->newlab.as<int32_t>();
  hipLaunchKernelGGL(k_newlab, dim3(nblk(m)), dim3(BT), 0, st, d_local, blk_t, m, cstart, keep, keeprank, newlab);
  hipLaunchKernelGGL(k_victims, dim3(nblk(nb)), dim3(BT), 0, st, nb, victim_of, blockstart, order, newlab);
  // zero list (FrmMain.cs:1510-1515) and merge order
  VCP_TRY(ens(ctx, s->zflag, (size_t)(m + 2) * 4));
  uint32_t* zflag = s->zflag.as<uint32_t>();
  VCP_HIP(ctx, hipMemsetAsync(zflag + m, 0, 8, st));  // k_zero_flag writes [0, m); the scan reads one entry more
  if (m > 0) hipLaunchKernelGGL(k_zero_flag, dim3(nblk(m)), dim3(BT), 0, st, newlab, order, m, zflag);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, zflag, zflag, m + 1, dmisc + 3));
  VCP_HIP(ctx, hipMemcpyAsync(hp, dmisc, 16, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (hp[2] != 0)
    return vcp_fail(ctx, VCP_ERR_INDEX, "clusForMerge index -1 while demoting the first cluster (FrmMain.cs:1487)");
  const uint32_t kept = hp[1], Z = hp[3];
  const uint32_t delSum = totalC - kept;
  VCP_TRY(ens(ctx, s->zlist, (size_t)(Z + 1) * 4));
  VCP_TRY(ens(ctx, s->zcoords, (size_t)(Z + 1) * 16));
  VCP_TRY(ens(ctx, s->zlab, (size_t)(Z + 1) * 4));
  uint32_t* zrank = s->tmp2.as<uint32_t>();  // [m + 1]: free again (it held the identity for the library-sort order)
  if (m > 0)
    hipLaunchKernelGGL(k_compact, dim3(nblk(m)), dim3(BT), 0, st, zflag, newlab, order, s->bl.as<uint32_t>(),
                       s->motor_bm.as<double>(), m, Z, zrank, s->zcoords.as<double>(), d_merge_order);
  VCP_HIP(ctx, hipGetLastError());
  // FrmMain.cs:1507-1516: one DBImproved over all noise, cf preset to the kept-cluster count
  int32_t cf = (int32_t)kept;
  int64_t ev = 0;
  if (Z > 0)
    VCP_TRY(vcp_dbscan_engine(ctx, s->zcoords.as<double>(), (int64_t)Z, 2, VCP_L1_2D, s->eps, s->min_pts, (int32_t)kept,
                              nullptr, s->zlab.as<int32_t>(), nullptr, nullptr, &cf, &ev, nullptr));
  // labels by original index: kept clusters and the noise pass result, one scatter
  hipLaunchKernelGGL(k_final_labels, dim3(nblk(n)), dim3(BT), 0, st, newlab, Z > 0 ? s->zlab.as<int32_t>() : nullptr, zrank,
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
  DevBuf* all[] = {&s->motor, &s->pkey, &s->orand, &s->raw, &s->blockof, &s->bl, &s->motor_bm, &s->blockstart,
                   &s->gtwice, &s->gnclus, &s->tmp0, &s->tmp1, &s->tmp2, &s->tmp3, &s->sorttmp, &s->blk_t, &s->csize,
                   &s->cstart, &s->kb, &s->zb, &s->keep, &s->order, &s->newlab, &s->zflag, &s->zlist, &s->zcoords,
                   &s->zlab, &s->misc, &s->biglist};
  for (DevBuf* b : all)
    if (b->p) (void)hipFree(b->p);
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
