// centroids.hip -- Tools.GetClusList / MergeIDByDistance / refreshCensAndClusByDictionary on MI355X.
//
// Centroids (BaseClass/Tools.cs:162-195): per cluster the binary64 mean of (X,Y,Z) and of
// (motor_x,motor_y).  The C# sums sequentially (LINQ Average); here the points are put in (label, index)
// order by a stable radix sort and every cluster is reduced by a FIXED tree: chunks of CH consecutive
// members -> one workgroup each (thread-strided partial sums, wave shuffle tree, 4 waves in order), then
// the chunk partials of a cluster are added in chunk order.  The partition depends only on the cluster
// sizes, so results are run-to-run deterministic; they differ from the sequential sum in the last bits
// (tests: 1e-12 relative).  Algorithmic bytes: 4 (label) + 24 (xyz) + 16 (motor) per point, read once.
#include <string.h>  // rocprim's texture_cache_iterator.hpp calls ::memset without including it

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "vcp_ctx.hpp"

extern "C" int vcp_dbscan_dev(vcp_ctx*, const double*, int64_t, int, int, double, int, int32_t, const uint8_t*,
                              int32_t*, uint8_t*, uint8_t*, int32_t*, int64_t*);

namespace {
constexpr int CT = 256;
constexpr int CH = 16384;  // members per chunk

__global__ __launch_bounds__(CT) void k_lab_keys(const int32_t* __restrict__ labels, int64_t n, int32_t K,
                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                uint32_t* __restrict__ bad) {
  int64_t i = (int64_t)blockIdx.x * CT + threadIdx.x;
  if (i >= n) return;
  int32_t l = labels[i];
  if (l < 0 || l > K) {  // clusList[p.clusterId - 1] out of range (Tools.cs:185)
    atomicAdd(bad, 1u);
    l = 0;
  }
  keys[i] = (uint32_t)l;
  vals[i] = (uint32_t)i;
}

// Segment bounds from the SORTED labels, no per-point atomics (global atomics run at the memory side here:
// 5 M adds into 27 k counters cost more than the sort): mark[l] = (last slot of label l) + 1; the exclusive
// max-scan of the marks is the first slot of every label; counts are the differences.
__global__ __launch_bounds__(CT) void k_lab_mark(const uint32_t* __restrict__ skey, int64_t n, uint32_t* __restrict__ mark) {
  int64_t t = (int64_t)blockIdx.x * CT + threadIdx.x;
  if (t >= n) return;
  const uint32_t k = skey[t];
  if (t == n - 1 || skey[t + 1] != k) mark[k] = (uint32_t)t + 1u;
}

// counts[k] = members of cluster k (k = 0: the noise bucket, reported as 0 chunks), nch[k] = chunks of cluster k
__global__ __launch_bounds__(CT) void k_nchunks(const uint32_t* __restrict__ segstart, int32_t K,
                                               uint32_t* __restrict__ counts, uint32_t* __restrict__ nch) {
  int k = blockIdx.x * CT + threadIdx.x;
  if (k > K) return;
  const uint32_t c = segstart[k + 1] - segstart[k];
  counts[k] = c;
  nch[k] = k == 0 ? 0u : (c + CH - 1) / CH;
}

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}

// one workgroup per chunk; chunkstart[k] = first chunk id of cluster k (k in 1..K), segstart[k] = first
// sorted slot of cluster k's members (after the noise bucket)
__global__ __launch_bounds__(CT) void k_chunk_sums(const uint32_t* __restrict__ sorted_idx,
                                                  const uint32_t* __restrict__ segstart,
                                                  const uint32_t* __restrict__ counts,
                                                  const uint32_t* __restrict__ chunkstart, int32_t K,
                                                  const double* __restrict__ xyz, const double* __restrict__ motor,
                                                  double* __restrict__ partial) {
  const uint32_t c = blockIdx.x;
  // binary search: largest k in [1,K] with chunkstart[k] <= c
  int lo = 1, hi = K;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (chunkstart[mid] <= c) lo = mid; else hi = mid - 1;
  }
  const int k = lo;
  const uint32_t within = c - chunkstart[k];
  const uint32_t beg = segstart[k] + within * CH;
  const uint32_t cnt = counts[k];
  const uint32_t end = segstart[k] + (within * CH + CH < cnt ? within * CH + CH : cnt);
  double s[5] = {0, 0, 0, 0, 0};
  for (uint32_t t = beg + threadIdx.x; t < end; t += CT) {
    const uint32_t i = sorted_idx[t];
    if (xyz) {
      s[0] += xyz[3 * (int64_t)i];
      s[1] += xyz[3 * (int64_t)i + 1];
      s[2] += xyz[3 * (int64_t)i + 2];
    }
    if (motor) {
      s[3] += motor[2 * (int64_t)i];
      s[4] += motor[2 * (int64_t)i + 1];
    }
  }
  __shared__ double sm[CT / 64][5];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 5; a++) {
    double v = wsum(s[a]);
    if (lane == 0) sm[w][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    double v = sm[0][threadIdx.x];
    for (int q = 1; q < CT / 64; q++) v += sm[q][threadIdx.x];
    partial[(size_t)c * 5 + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(CT) void k_centroid_final(const double* __restrict__ partial,
                                                      const uint32_t* __restrict__ counts,
                                                      const uint32_t* __restrict__ chunkstart, int32_t K,
                                                      int has_xyz, int has_motor, double* __restrict__ c3,
                                                      double* __restrict__ c2, int64_t* __restrict__ out_counts) {
  int k = blockIdx.x * CT + threadIdx.x + 1;
  if (k > K) return;
  const uint32_t cnt = counts[k];
  if (out_counts) out_counts[k - 1] = (int64_t)cnt;
  double s[5] = {0, 0, 0, 0, 0};
  for (uint32_t c = chunkstart[k]; c < chunkstart[k + 1]; c++)
    for (int a = 0; a < 5; a++) s[a] += partial[(size_t)c * 5 + a];
  const double nan = __longlong_as_double(0x7FF8000000000000LL);
  const double dn = (double)cnt;
  if (has_xyz && c3)
    for (int a = 0; a < 3; a++) c3[3 * (size_t)(k - 1) + a] = cnt ? s[a] / dn : nan;
  if (has_motor && c2)
    for (int a = 0; a < 2; a++) c2[2 * (size_t)(k - 1) + a] = cnt ? s[3 + a] / dn : nan;
}

// getFixedPtsCentroid (BaseClass/Tools.cs:78-111): the same fixed tree over sum(w X), sum(w Y), sum(w Z), sum(w) with
// w = 1 where the member's clusterId != 0 and duplicates are ignored, else its ptsCount (:88-101).  X * ptsCount is
// rounded before it is added, as in the C# (no FMA contraction in this library).
__global__ __launch_bounds__(CT) void k_chunk_sums_w(const uint32_t* __restrict__ sorted_idx,
                                                    const uint32_t* __restrict__ segstart,
                                                    const uint32_t* __restrict__ counts,
                                                    const uint32_t* __restrict__ chunkstart, int32_t K,
                                                    const double* __restrict__ xyz, const int32_t* __restrict__ cluster_id,
                                                    const int32_t* __restrict__ pts_count, int ignore_dup,
                                                    double* __restrict__ partial) {
  const uint32_t c = blockIdx.x;
  int lo = 1, hi = K;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (chunkstart[mid] <= c) lo = mid; else hi = mid - 1;
  }
  const int k = lo;
  const uint32_t within = c - chunkstart[k];
  const uint32_t beg = segstart[k] + within * CH;
  const uint32_t cnt = counts[k];
  const uint32_t end = segstart[k] + (within * CH + CH < cnt ? within * CH + CH : cnt);
  double s[4] = {0, 0, 0, 0};
  for (uint32_t t = beg + threadIdx.x; t < end; t += CT) {
    const uint32_t i = sorted_idx[t];
    const double x = xyz[3 * (int64_t)i], y = xyz[3 * (int64_t)i + 1], z = xyz[3 * (int64_t)i + 2];
    if (cluster_id[i] != 0 && ignore_dup) {
      s[0] += x;
      s[1] += y;
      s[2] += z;
      s[3] += 1.0;
    } else {
      const double w = (double)pts_count[i];
      s[0] += x * w;
      s[1] += y * w;
      s[2] += z * w;
      s[3] += w;  // integers below 2^53: exact
    }
  }
  __shared__ double sm[CT / 64][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 4; a++) {
    double v = wsum(s[a]);
    if (lane == 0) sm[w][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = sm[0][threadIdx.x];
    for (int q = 1; q < CT / 64; q++) v += sm[q][threadIdx.x];
    partial[(size_t)c * 4 + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(CT) void k_centroid_final_w(const double* __restrict__ partial,
                                                        const uint32_t* __restrict__ counts,
                                                        const uint32_t* __restrict__ chunkstart, int32_t K,
                                                        double* __restrict__ c3, int64_t* __restrict__ inside,
                                                        uint32_t* __restrict__ empty) {
  int k = blockIdx.x * CT + threadIdx.x + 1;
  if (k > K) return;
  if (counts[k] == 0) atomicAdd(empty, 1u);  // clusList[i].li[0] on an empty list throws (Tools.cs:106)
  double s[4] = {0, 0, 0, 0};
  for (uint32_t c = chunkstart[k]; c < chunkstart[k + 1]; c++)
    for (int a = 0; a < 4; a++) s[a] += partial[(size_t)c * 4 + a];
  if (inside) inside[k - 1] = (int64_t)s[3];
  for (int a = 0; a < 3; a++) c3[3 * (size_t)(k - 1) + a] = s[a] / s[3];  // insideNum == 0: 0/0 = NaN, like the C#
}

struct WeightArgs {
  const int32_t* cluster_id;
  const int32_t* pts_count;
  int ignore_dup;
};

int centroids_dev(vcp_ctx* ctx, const double* d_xyz, const double* d_motor, const int32_t* d_labels, int64_t n,
                  int32_t K, double* d_c3, double* d_c2, int64_t* d_counts, const WeightArgs* wa = nullptr) {
  hipStream_t st = ctx->stream;
  if (K == 0) return VCP_OK;
  // aux0: counts [K+2] | nch [K+2] ; aux1: keys in/out ; aux2: vals in/out ; aux3: rocprim temp ; aux4: partial
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, (size_t)(K + 2) * 4 * 2 + 64));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux1, (size_t)(n + 1) * 4 * 2));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux2, (size_t)(n + 1) * 4 * 2));
  uint32_t* counts = ctx->b_aux0.as<uint32_t>();
  uint32_t* nch = counts + (K + 2);
  uint32_t* bad = nch + (K + 2);
  uint32_t* keys_in = ctx->b_aux1.as<uint32_t>();
  uint32_t* keys_out = keys_in + (n + 1);
  uint32_t* vals_in = ctx->b_aux2.as<uint32_t>();
  uint32_t* vals_out = vals_in + (n + 1);
  vcp_phase(ctx, "centroid_sort");
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux5, (size_t)(K + 4) * 4));
  uint32_t* segstart = ctx->b_aux5.as<uint32_t>();  // [K+2]: first sorted slot of label k; [K+1] = n
  VCP_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)(K + 2) * 4 * 2 + 64, st));
  VCP_HIP(ctx, hipMemsetAsync(segstart, 0, (size_t)(K + 4) * 4, st));
  hipLaunchKernelGGL(k_lab_keys, dim3(vcp_blocks(n, CT)), dim3(CT), 0, st, d_labels, n, K, keys_in, vals_in, bad);
  int bits = 1;
  while (((int64_t)1 << bits) <= K) bits++;
  size_t temp_bytes = 0;
  VCP_HIP(ctx, rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, bits, st));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux3, temp_bytes + 64));
  VCP_HIP(ctx, rocprim::radix_sort_pairs(ctx->b_aux3.p, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0,
                                         bits, st));
  vcp_phase(ctx, "centroid_reduce");
  if (n > 0) hipLaunchKernelGGL(k_lab_mark, dim3(vcp_blocks(n, CT)), dim3(CT), 0, st, keys_out, n, segstart);
  VCP_TRY(vcp_exclusive_max_scan_u32(ctx, segstart, segstart, K + 2, nullptr));
  hipLaunchKernelGGL(k_nchunks, dim3(vcp_blocks(K + 1, CT)), dim3(CT), 0, st, segstart, K, counts, nch);
  uint32_t* d_tot = bad + 4;
  VCP_TRY(vcp_exclusive_scan_u32(ctx, nch, nch, K + 2, d_tot));  // nch -> chunkstart, [K+1] = total chunks
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(hp, bad, 8 * 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (hp[0] != 0) return vcp_fail(ctx, VCP_ERR_INDEX, "%u labels outside 0..K (clusList[clusterId-1], Tools.cs:185)", hp[0]);
  const uint32_t nchunks = hp[4];
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux4, ((size_t)nchunks + 1) * 5 * sizeof(double)));
  double* partial = ctx->b_aux4.as<double>();
  if (wa) {
    uint32_t* empty = bad + 2;
    if (nchunks > 0)
      hipLaunchKernelGGL(k_chunk_sums_w, dim3(nchunks), dim3(CT), 0, st, vals_out, segstart, counts, nch, K, d_xyz,
                         wa->cluster_id, wa->pts_count, wa->ignore_dup, partial);
    hipLaunchKernelGGL(k_centroid_final_w, dim3(vcp_blocks(K, CT)), dim3(CT), 0, st, partial, counts, nch, K, d_c3,
                       d_counts, empty);
    VCP_HIP(ctx, hipGetLastError());
    VCP_HIP(ctx, hipMemcpyAsync(hp, empty, 4, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    if (hp[0] != 0)
      return vcp_fail(ctx, VCP_ERR_INDEX, "%u empty groups: clusList[i].li[0] throws (Tools.cs:106)", hp[0]);
    return VCP_OK;
  }
  if (nchunks > 0)
    hipLaunchKernelGGL(k_chunk_sums, dim3(nchunks), dim3(CT), 0, st, vals_out, segstart, counts, nch, K, d_xyz, d_motor,
                       partial);
  hipLaunchKernelGGL(k_centroid_final, dim3(vcp_blocks(K, CT)), dim3(CT), 0, st, partial, counts, nch, K,
                     d_xyz != nullptr, d_motor != nullptr, d_c3, d_c2, d_counts);
  VCP_HIP(ctx, hipGetLastError());
  return VCP_OK;
}

// ---- merge helpers -------------------------------------------------------------------------------
__global__ __launch_bounds__(CT) void k_first_of(const int32_t* __restrict__ lab, int32_t K, int32_t nclus,
                                                uint32_t* __restrict__ first) {
  int k = blockIdx.x * CT + threadIdx.x;
  if (k >= K) return;
  int32_t l = lab[k];
  if (l > 0 && l <= nclus) atomicMin(&first[l], (uint32_t)k);
}
__global__ __launch_bounds__(CT) void k_map_to(const int32_t* __restrict__ lab, const int32_t* __restrict__ ids,
                                              const uint32_t* __restrict__ first, int32_t K,
                                              int32_t* __restrict__ map_to, uint32_t* __restrict__ merged) {
  int k = blockIdx.x * CT + threadIdx.x;
  if (k >= K) return;
  int32_t l = lab[k];
  int32_t m = 0;
  if (l > 0) {
    uint32_t f = first[l];
    if (f != (uint32_t)k && ids[f] != ids[k]) {  // q.IDBeforeMerge != p.IDBeforeMerge, Tools.cs:603
      m = ids[f];
      atomicAdd(merged, 1u);
    }
  }
  map_to[k] = m;
}

// relabel through the dictionary and renumber survivors 1..K' ascending (Tools.cs:534-565)
__global__ __launch_bounds__(CT) void k_survivor_flag(const int32_t* __restrict__ map_by_id, int32_t K,
                                                     uint32_t* __restrict__ flag, uint32_t* __restrict__ bad) {
  int k = blockIdx.x * CT + threadIdx.x;
  if (k >= K) return;
  int32_t t = map_by_id[k];
  flag[k] = t == 0 ? 1u : 0u;
  if (t != 0 && (t < 1 || t > K || t - 1 == k || map_by_id[t - 1] != 0)) atomicAdd(bad, 1u);
}
__global__ __launch_bounds__(CT) void k_relabel(int32_t* __restrict__ labels, int64_t n, int32_t K,
                                               const int32_t* __restrict__ map_by_id,
                                               const uint32_t* __restrict__ rank, uint32_t* __restrict__ bad) {
  int64_t i = (int64_t)blockIdx.x * CT + threadIdx.x;
  if (i >= n) return;
  int32_t l = labels[i];
  if (l == 0) return;
  if (l < 1 || l > K) {
    atomicAdd(bad, 1u);
    return;
  }
  int32_t t = map_by_id[l - 1];
  if (t != 0) l = t;
  labels[i] = (int32_t)rank[l - 1] + 1;
}
__global__ void k_count_zero(const int64_t* __restrict__ counts, int32_t K, uint32_t* __restrict__ nzero) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K && counts[k] == 0) atomicAdd(nzero, 1u);
}

}  // namespace

extern "C" {

int vcp_centroids_dev(vcp_ctx* ctx, const double* d_xyz, const double* d_motor, const int32_t* d_labels, int64_t n,
                      int32_t K, double* d_c3, double* d_c2, int64_t* d_counts) {
  if (!ctx) return VCP_ERR_ARG;
  if (n < 0 || K < 0 || (n > 0 && !d_labels)) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (n >= 0x7FFFFFF0LL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "n beyond 32-bit indexing");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  VCP_TRY(centroids_dev(ctx, d_xyz, d_motor, d_labels, n, K, d_c3, d_c2, d_counts));
  VCP_TRY(vcp_phase_finish(ctx));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VCP_OK;
}

int vcp_centroids(vcp_ctx* ctx, const double* xyz, const double* motor, const int32_t* labels, int64_t n, int32_t K,
                  double* c3, double* c2, int64_t* counts) {
  if (!ctx) return VCP_ERR_ARG;
  if (n < 0 || K < 0 || (n > 0 && !labels)) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  VCP_TRY(vcp_bind(ctx));
  hipStream_t st = ctx->stream;
  size_t nn = (size_t)(n > 0 ? n : 1), kk = (size_t)(K > 0 ? K : 1);
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, nn * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, nn * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in3, nn * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, kk * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out1, kk * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out2, kk * 8));
  if (n > 0) {
    if (xyz) VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, xyz, (size_t)n * 24, hipMemcpyHostToDevice, st));
    if (motor) VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, motor, (size_t)n * 16, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in3.p, labels, (size_t)n * 4, hipMemcpyHostToDevice, st));
  }
  VCP_TRY(vcp_centroids_dev(ctx, xyz ? ctx->b_in0.as<double>() : nullptr, motor ? ctx->b_in2.as<double>() : nullptr,
                            ctx->b_in3.as<int32_t>(), n, K, ctx->b_out0.as<double>(), ctx->b_out1.as<double>(),
                            ctx->b_out2.as<int64_t>()));
  if (K > 0) {
    if (c3 && xyz) VCP_HIP(ctx, hipMemcpyAsync(c3, ctx->b_out0.p, (size_t)K * 24, hipMemcpyDeviceToHost, st));
    if (c2 && motor) VCP_HIP(ctx, hipMemcpyAsync(c2, ctx->b_out1.p, (size_t)K * 16, hipMemcpyDeviceToHost, st));
    if (counts) VCP_HIP(ctx, hipMemcpyAsync(counts, ctx->b_out2.p, (size_t)K * 8, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
  }
  return VCP_OK;
}

int vcp_centroids_weighted(vcp_ctx* ctx, const double* xyz, const int32_t* group, const int32_t* cluster_id,
                           const int32_t* pts_count, int64_t n, int32_t K, int ignore_duplication, double* c3,
                           int64_t* inside_num) {
  if (!ctx) return VCP_ERR_ARG;
  if (n < 0 || K < 0 || (n > 0 && (!xyz || !group || !pts_count)) || (K > 0 && !c3))
    return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (n >= 0x7FFFFFF0LL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "n beyond 32-bit indexing");
  if (K == 0) return VCP_OK;
  if (n == 0) return vcp_fail(ctx, VCP_ERR_INDEX, "empty groups: clusList[i].li[0] throws (Tools.cs:106)");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  hipStream_t st = ctx->stream;
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)n * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in1, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in3, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, (size_t)K * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out2, (size_t)K * 8));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, xyz, (size_t)n * 24, hipMemcpyHostToDevice, st));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in3.p, group, (size_t)n * 4, hipMemcpyHostToDevice, st));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in1.p, cluster_id ? cluster_id : group, (size_t)n * 4, hipMemcpyHostToDevice, st));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, pts_count, (size_t)n * 4, hipMemcpyHostToDevice, st));
  WeightArgs wa{ctx->b_in1.as<int32_t>(), ctx->b_in2.as<int32_t>(), ignore_duplication ? 1 : 0};
  VCP_TRY(centroids_dev(ctx, ctx->b_in0.as<double>(), nullptr, ctx->b_in3.as<int32_t>(), n, K, ctx->b_out0.as<double>(),
                        nullptr, ctx->b_out2.as<int64_t>(), &wa));
  VCP_HIP(ctx, hipMemcpyAsync(c3, ctx->b_out0.p, (size_t)K * 24, hipMemcpyDeviceToHost, st));
  if (inside_num) VCP_HIP(ctx, hipMemcpyAsync(inside_num, ctx->b_out2.p, (size_t)K * 8, hipMemcpyDeviceToHost, st));
  VCP_TRY(vcp_phase_finish(ctx));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  return VCP_OK;
}

int vcp_merge_centroids(vcp_ctx* ctx, const double* cxy, const int32_t* ids, int32_t K, double thr, int32_t* map_to,
                        int32_t* merge_count) {
  if (!ctx) return VCP_ERR_ARG;
  if (K < 0 || (K > 0 && (!cxy || !ids || !map_to))) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (merge_count) *merge_count = 0;
  if (K == 0) return VCP_OK;
  {  // Dictionary.Add / HashSet logic of Tools.cs:597-606 presumes distinct ids
    std::vector<int32_t> s(ids, ids + K);
    std::sort(s.begin(), s.end());
    if (std::adjacent_find(s.begin(), s.end()) != s.end())
      return vcp_fail(ctx, VCP_ERR_ARG, "duplicate centroid ids (Dictionary.Add would throw, Tools.cs:606)");
  }
  VCP_TRY(vcp_bind(ctx));
  hipStream_t st = ctx->stream;
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)K * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in3, (size_t)K * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, (size_t)K * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out3, (size_t)K * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, (size_t)(K + 2) * 4 + 64));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, cxy, (size_t)K * 16, hipMemcpyHostToDevice, st));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in3.p, ids, (size_t)K * 4, hipMemcpyHostToDevice, st));
  // Tools.cs:591-592: new DBImproved().dbscan(centers, thre, 2) on (X,Y) copied into motor_x/motor_y
  int32_t nclus = 0;
  VCP_TRY(vcp_dbscan_dev(ctx, ctx->b_in0.as<double>(), K, 2, VCP_L1_2D, thr, 2, 0, nullptr, ctx->b_out0.as<int32_t>(),
                         nullptr, nullptr, &nclus, nullptr));
  uint32_t* first = ctx->b_aux0.as<uint32_t>();
  uint32_t* merged = first + (K + 2);
  VCP_HIP(ctx, hipMemsetAsync(first, 0xFF, (size_t)(K + 2) * 4, st));
  VCP_HIP(ctx, hipMemsetAsync(merged, 0, 16, st));
  hipLaunchKernelGGL(k_first_of, dim3(vcp_blocks(K, CT)), dim3(CT), 0, st, ctx->b_out0.as<int32_t>(), K, K, first);
  hipLaunchKernelGGL(k_map_to, dim3(vcp_blocks(K, CT)), dim3(CT), 0, st, ctx->b_out0.as<int32_t>(),
                     ctx->b_in3.as<int32_t>(), first, K, ctx->b_out3.as<int32_t>(), merged);
  VCP_HIP(ctx, hipGetLastError());
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(map_to, ctx->b_out3.p, (size_t)K * 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(hp, merged, 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (merge_count) *merge_count = (int32_t)hp[0];
  return VCP_OK;
}

int vcp_refresh_by_dictionary(vcp_ctx* ctx, const double* xyz, const double* motor, int32_t* labels, int64_t n,
                              int32_t K, const int32_t* map_by_id, int32_t* new_k, double* c3, double* c2,
                              int64_t* counts) {
  if (!ctx) return VCP_ERR_ARG;
  if (n < 0 || K < 0 || (n > 0 && !labels) || (K > 0 && !map_by_id)) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (new_k) *new_k = 0;
  if (K == 0) return VCP_OK;
  VCP_TRY(vcp_bind(ctx));
  hipStream_t st = ctx->stream;
  size_t nn = (size_t)(n > 0 ? n : 1);
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, nn * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, nn * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in3, nn * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in1, (size_t)K * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out3, (size_t)(K + 2) * 4 + 64));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, (size_t)K * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out1, (size_t)K * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out2, (size_t)K * 8));
  if (n > 0) {
    if (xyz) VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, xyz, (size_t)n * 24, hipMemcpyHostToDevice, st));
    if (motor) VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, motor, (size_t)n * 16, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in3.p, labels, (size_t)n * 4, hipMemcpyHostToDevice, st));
  }
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in1.p, map_by_id, (size_t)K * 4, hipMemcpyHostToDevice, st));
  uint32_t* flag = ctx->b_out3.as<uint32_t>();
  uint32_t* bad = flag + (K + 2);
  VCP_HIP(ctx, hipMemsetAsync(bad, 0, 32, st));
  hipLaunchKernelGGL(k_survivor_flag, dim3(vcp_blocks(K, CT)), dim3(CT), 0, st, ctx->b_in1.as<int32_t>(), K, flag, bad);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, flag, flag, K, bad + 1));
  if (n > 0)
    hipLaunchKernelGGL(k_relabel, dim3(vcp_blocks(n, CT)), dim3(CT), 0, st, ctx->b_in3.as<int32_t>(), n, K,
                       ctx->b_in1.as<int32_t>(), flag, bad);
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(hp, bad, 8, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (hp[0] != 0) return vcp_fail(ctx, VCP_ERR_INDEX, "dictionary or labels out of range (clusList[dic[id]-1], Tools.cs:530)");
  const int32_t nk = (int32_t)hp[1];
  vcp_phase_reset(ctx);
  VCP_TRY(centroids_dev(ctx, xyz ? ctx->b_in0.as<double>() : nullptr, motor ? ctx->b_in2.as<double>() : nullptr,
                        ctx->b_in3.as<int32_t>(), n, nk, ctx->b_out0.as<double>(), ctx->b_out1.as<double>(),
                        ctx->b_out2.as<int64_t>()));
  if (nk > 0) {
    VCP_HIP(ctx, hipMemsetAsync(bad, 0, 8, st));
    hipLaunchKernelGGL(k_count_zero, dim3(vcp_blocks(nk, CT)), dim3(CT), 0, st, ctx->b_out2.as<int64_t>(), nk, bad);
    VCP_HIP(ctx, hipMemcpyAsync(hp, bad, 4, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    if (hp[0] != 0)
      return vcp_fail(ctx, VCP_ERR_EMPTY, "%u surviving clusters are empty (Average() of an empty list, Tools.cs:568)", hp[0]);
  }
  if (n > 0) VCP_HIP(ctx, hipMemcpyAsync(labels, ctx->b_in3.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (nk > 0) {
    if (c3 && xyz) VCP_HIP(ctx, hipMemcpyAsync(c3, ctx->b_out0.p, (size_t)nk * 24, hipMemcpyDeviceToHost, st));
    if (c2 && motor) VCP_HIP(ctx, hipMemcpyAsync(c2, ctx->b_out1.p, (size_t)nk * 16, hipMemcpyDeviceToHost, st));
    if (counts) VCP_HIP(ctx, hipMemcpyAsync(counts, ctx->b_out2.p, (size_t)nk * 8, hipMemcpyDeviceToHost, st));
  }
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (new_k) *new_k = nk;
  return VCP_OK;
}

}  // extern "C"
