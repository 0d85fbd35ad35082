// dbdead.hip -- the v1.0 class DB (BaseClass/DB.cs:14-115) on MI355X.  Dead in the reference (its only use is commented
// out, FrmMain.cs:38) but part of the class surface, so it gets the same C-ABI entry (metric VCP_SIGNED_SUM_2D).
//
// DB's "distance" is the SIGNED sum dx + dy (DB.cs:21): j is a neighbour of p iff (p.X - j.X) + (p.Y - j.Y) <= e, i.e.
// iff s_j >= s_p - e with s = X + Y -- a half line in ONE dimension, not a ball.  Everything then collapses (derived
// from DB.cs:57-115; the gpu tests check it against a literal transcription of the class on random inputs):
//   * sort the shown points by s, descending; R(a) = number of positions within reach of position a (binary search);
//     core(a) <=> R(a) >= minPts; only points that are core and not classed on entry PROPAGATE (they are the ones
//     expandCluster queries, :66-69);
//   * a cluster that reaches position E also reaches everything the propagators below E reach: the reached set is the
//     prefix [0, E*) with E* the least fixed point >= E of G(E) = max(E, R(last propagator below E)) -- a max-scan
//     and a suffix-min over fixed-point flags;
//   * the main loop (:95-112) walks the LIST order: point i seeds a cluster iff it is shown, core, not classed on
//     entry and not yet inside the reached prefix, i.e. pos(i) >= max over earlier eligible j of E*(j) -- an
//     exclusive max-scan in list order; every cluster relabels everything it reaches (:87 unconditional), so in the
//     end every reached point carries the LAST cluster id;
//   * iritatorNum (:19): (#shown) evaluations per isKeyPoint call; a point not classed on entry is queried by the main
//     loop if it is not yet reached when its turn comes, and once more when a cluster first reaches it.
// The C# evaluates fl(fl(dx) + fl(dy)).  That equals the 1-D relation on s exactly when every coordinate is a
// multiple of one power of two with fewer than 52 bits of span (checked on the device); otherwise it is accepted when
// no pair of shown points sits within the rounding band of the threshold (checked too).  A cloud that fails both, an e < 0
// or NaN (a point is then not its own neighbour: the class leaves seeds unclassed) and non-finite coordinates take the
// pair-by-pair form of dbpairs.hip: the C#'s expression on every pair, O(n^2), up to 2^21 points.
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <cmath>

#include "dbscan_engine.hpp"

namespace {
constexpr int DT = 256;

// [0] shown, [1] non-finite shown coords, [2] ambiguous pairs, [3] seeds K, [4] queries Q, [5] -min lowbit + 4096,
// [6] max exponent + 4096, [7] E_total
__device__ __forceinline__ uint64_t sortable_desc(double v) {
  uint64_t u = (uint64_t)__double_as_longlong(v);
  u ^= (u >> 63) ? ~0ull : 0x8000000000000000ull;  // ascending order of the doubles
  return ~u;                                        // descending
}

__global__ __launch_bounds__(DT) void k_db_key(const double* __restrict__ c, int64_t n, int stride,
                                              const uint8_t* __restrict__ mask, uint64_t* __restrict__ key,
                                              uint32_t* __restrict__ idx, unsigned long long* __restrict__ ctr) {
  const int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x;
  if (i >= n) return;
  idx[i] = (uint32_t)i;
  if (mask && !mask[i]) {
    key[i] = ~0ull;  // not shown: behind every shown point
    return;
  }
  const double x = c[i * stride], y = c[i * stride + 1];
  atomicAdd(&ctr[0], 1ull);
  if (!isfinite(x) || !isfinite(y)) {
    atomicAdd(&ctr[1], 1ull);
    key[i] = ~0ull - 1;
    return;
  }
  key[i] = sortable_desc(x + y);
  for (double v : {x, y}) {
    if (v == 0.0) continue;
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    int e = (int)((b >> 52) & 0x7FF);
    uint64_t m = b & 0xFFFFFFFFFFFFFull;
    if (e == 0) e = 1; else m |= 1ull << 52;  // subnormal / normal
    const int low = e - 1075 + __ffsll((long long)m) - 1;  // exponent of the lowest set bit
    const int top = e - 1023;
    atomicMax(&ctr[5], (unsigned long long)(4096 - low));
    atomicMax(&ctr[6], (unsigned long long)(4096 + top));
  }
}

__global__ __launch_bounds__(DT) void k_db_unkey(const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sidx,
                                                uint32_t m, double* __restrict__ sk, uint32_t* __restrict__ pos) {
  const uint32_t a = blockIdx.x * DT + threadIdx.x;
  if (a >= m) return;
  uint64_t u = ~skey[a];
  u ^= (u >> 63) ? 0x8000000000000000ull : ~0ull;
  sk[a] = __longlong_as_double((long long)u);
  pos[sidx[a]] = a;
}

// first position b in [0, m) with sk[a] - sk[b] > lim (the differences do not decrease with b: sk is descending and a
// rounded subtraction is monotone)
__device__ __forceinline__ uint32_t reach(const double* __restrict__ sk, uint32_t m, double ka, double lim) {
  uint32_t lo = 0, hi = m;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (ka - sk[mid] <= lim) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(DT) void k_db_reach(const double* __restrict__ sk, const uint32_t* __restrict__ sidx, uint32_t m,
                                                double eps, int min_pts, double band, const uint8_t* __restrict__ in_classed,
                                                uint32_t* __restrict__ R, uint32_t* __restrict__ propmark,
                                                unsigned long long* __restrict__ ctr) {
  const uint32_t a = blockIdx.x * DT + threadIdx.x;
  if (a > m) return;
  if (a == m) {
    propmark[m] = 0;
    return;
  }
  const double ka = sk[a];
  const uint32_t r = reach(sk, m, ka, eps);
  R[a] = r;
  if (band > 0.0 && reach(sk, m, ka, eps - band) != reach(sk, m, ka, eps + band)) atomicAdd(&ctr[2], 1ull);
  const bool prop = (int64_t)r >= (int64_t)min_pts && !(in_classed && in_classed[sidx[a]]);
  propmark[a] = prop ? a + 1u : 0u;  // exclusive max-scan -> 1 + last propagator below E
}

// w[t] for t = m - E: t + 1 where E is a fixed point of G, else 0 (exclusive max-scan -> least fixed point >= E0)
__global__ __launch_bounds__(DT) void k_db_fixed(const uint32_t* __restrict__ lp1, const uint32_t* __restrict__ R, uint32_t m,
                                                uint32_t* __restrict__ w) {
  const uint32_t E = blockIdx.x * DT + threadIdx.x;
  if (E > m + 1) return;
  if (E == m + 1) {
    w[m + 1] = 0;
    return;
  }
  const uint32_t l = lp1[E];
  const uint32_t g = l ? max(E, R[l - 1]) : E;
  w[m - E] = g == E ? (m - E) + 1u : 0u;
}

// list order: E*(i) of every eligible point (shown, core, not classed on entry), 0 otherwise
__global__ __launch_bounds__(DT) void k_db_efin(int64_t n, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ in_classed,
                                               const uint32_t* __restrict__ pos, const uint32_t* __restrict__ R,
                                               const uint32_t* __restrict__ xs, uint32_t m, int min_pts,
                                               uint32_t* __restrict__ efin) {
  const int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x;
  if (i > n) return;
  uint32_t v = 0;
  if (i < n && !(mask && !mask[i]) && !(in_classed && in_classed[i])) {
    const uint32_t r = R[pos[i]];
    if ((int64_t)r >= (int64_t)min_pts) v = m - (xs[m - r + 1] - 1u);  // least fixed point >= r (m always is one)
  }
  efin[i] = v;
}

__global__ __launch_bounds__(DT) void k_db_out(int64_t n, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ in_classed,
                                              const uint32_t* __restrict__ pos, const uint32_t* __restrict__ R,
                                              const uint32_t* __restrict__ ebefore, int min_pts, int32_t cf_in,
                                              int32_t* __restrict__ labels, uint8_t* __restrict__ is_core,
                                              uint8_t* __restrict__ is_classed, unsigned long long* __restrict__ ctr,
                                              int phase) {
  const int64_t i = (int64_t)blockIdx.x * DT + threadIdx.x;
  if (i >= n) return;
  const bool shown = !(mask && !mask[i]);
  const bool cls = in_classed && in_classed[i];
  const uint32_t etot = ebefore[n];  // exclusive max-scan over n + 1 entries: the last one is the overall maximum
  if (phase == 0) {
    // seeds and queries (needs every ebefore): one pass of counting
    unsigned long long q = 0, k = 0;
    if (shown && !cls) {
      const uint32_t p = pos[i];
      const bool unreached = p >= ebefore[i];
      q = (unreached ? 1ull : 0ull) + (p < etot ? 1ull : 0ull);
      if (unreached && (int64_t)R[p] >= (int64_t)min_pts) k = 1;
    }
    if (q) atomicAdd(&ctr[4], q);
    if (k) atomicAdd(&ctr[3], k);
    if (i == 0) ctr[7] = etot;
    return;
  }
  const int32_t K = (int32_t)ctr[3];
  const bool reached = shown && pos[i] < etot;
  if (reached) labels[i] = cf_in + K;  // every cluster relabels all it reaches (:87): the last one stays
  else if (!in_classed) labels[i] = 0;
  if (is_core) is_core[i] = (reached && !cls && (int64_t)R[pos[i]] >= (int64_t)min_pts) ? 1 : 0;
  if (is_classed) is_classed[i] = (cls || reached) ? 1 : 0;
}

}  // namespace

// d_* device pointers; cf_out / dist_evals host pointers (may be null)
int vcp_db_engine(vcp_ctx* ctx, const double* d_coords, int64_t n, int stride, double eps, int min_pts, int32_t cf_in,
                  const uint8_t* d_mask, const uint8_t* d_in_classed, int32_t* d_labels, uint8_t* d_is_core,
                  uint8_t* d_is_classed, int32_t* cf_out, int64_t* dist_evals) {
  // what the 1-D formulation below cannot take goes to the pair-by-pair form (dbpairs.hip): exact for every input, O(n^2)
#define VCP_DB_PAIRS()                                                                                                 \
  vcp_db_pairs_engine(ctx, d_coords, n, stride, eps, min_pts, cf_in, d_mask, d_in_classed, d_labels, d_is_core, d_is_classed, \
                      cf_out, dist_evals)
  if (!(eps >= 0.0)) return VCP_DB_PAIRS();  // e < 0 or NaN: a point is not its own neighbour
  if (n >= 0x7FFFFFF0LL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "n beyond 32-bit indexing");
  hipStream_t st = ctx->stream;
  const size_t N1 = (size_t)n + 2;
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, N1 * 8 * 2));  // keys in / out
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux1, N1 * 4 * 2));  // idx in / out
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux2, N1 * 8));      // sk
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux4, N1 * 4 * 6));  // pos, R, propmark/lp1, w/xs, efin/ebefore
  VCP_TRY(vcp_ensure(ctx, ctx->b_misc, 64 * 8));
  uint64_t* key_in = ctx->b_aux0.as<uint64_t>();
  uint64_t* key_out = key_in + N1;
  uint32_t* idx_in = ctx->b_aux1.as<uint32_t>();
  uint32_t* idx_out = idx_in + N1;
  double* sk = ctx->b_aux2.as<double>();
  uint32_t* pos = ctx->b_aux4.as<uint32_t>();
  uint32_t* R = pos + N1;
  uint32_t* lp1 = R + N1;
  uint32_t* xs = lp1 + N1;
  uint32_t* efin = xs + N1 + 4;
  unsigned long long* ctr = reinterpret_cast<unsigned long long*>(ctx->b_misc.p);
  unsigned long long* hc = reinterpret_cast<unsigned long long*>(ctx->pinned) + 128;
  vcp_phase(ctx, "db_sort");
  VCP_HIP(ctx, hipMemsetAsync(ctr, 0, 8 * 8, st));
  hipLaunchKernelGGL(k_db_key, dim3(vcp_blocks(n, DT)), dim3(DT), 0, st, d_coords, n, stride, d_mask, key_in, idx_in, ctr);
  size_t tb = 0;
  VCP_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, key_in, key_out, idx_in, idx_out, (size_t)n, 0, 64, st));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux3, tb + 64));
  VCP_HIP(ctx, rocprim::radix_sort_pairs(ctx->b_aux3.p, tb, key_in, key_out, idx_in, idx_out, (size_t)n, 0, 64, st));
  VCP_HIP(ctx, hipMemcpyAsync(hc, ctr, 8 * 8, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  const uint32_t m = (uint32_t)hc[0];
  if (hc[1] != 0) return VCP_DB_PAIRS();  // non-finite coordinates among the shown points
  // is the C#'s fl(fl(dx) + fl(dy)) exactly s_p - s_j?  yes when all coordinates share a binary grid of < 52 bits span
  const bool have = hc[5] != 0;
  const int low = 4096 - (int)hc[5], top = (int)hc[6] - 4096;
  const bool exact = !have || (top + 2 - low <= 52);
  const double maxabs = have ? std::ldexp(1.0, top + 1) : 0.0;
  const double band = exact ? 0.0 : 16.0 * 4.0 * maxabs * 2.220446049250313e-16;
  vcp_phase(ctx, "db_reach");
  if (m > 0) hipLaunchKernelGGL(k_db_unkey, dim3(vcp_blocks(m, DT)), dim3(DT), 0, st, key_out, idx_out, m, sk, pos);
  hipLaunchKernelGGL(k_db_reach, dim3(vcp_blocks((int64_t)m + 1, DT)), dim3(DT), 0, st, sk, idx_out, m, eps, min_pts, band,
                     d_in_classed, R, lp1, ctr);
  if (!exact) {
    // a neighbour candidate within rounding of the threshold while the coordinates share no binary grid: the signed-sum
    // relation is not provably 1-D -- decided before anything is written to the caller's arrays
    VCP_HIP(ctx, hipMemcpyAsync(hc, ctr, 8 * 8, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    if (hc[2] != 0) return VCP_DB_PAIRS();
  }
  VCP_TRY(vcp_exclusive_max_scan_u32(ctx, lp1, lp1, (int64_t)m + 1, nullptr));
  hipLaunchKernelGGL(k_db_fixed, dim3(vcp_blocks((int64_t)m + 2, DT)), dim3(DT), 0, st, lp1, R, m, xs);
  VCP_TRY(vcp_exclusive_max_scan_u32(ctx, xs, xs, (int64_t)m + 2, nullptr));
  vcp_phase(ctx, "db_seeds");
  hipLaunchKernelGGL(k_db_efin, dim3(vcp_blocks(n + 1, DT)), dim3(DT), 0, st, n, d_mask, d_in_classed, pos, R, xs, m, min_pts,
                     efin);
  VCP_TRY(vcp_exclusive_max_scan_u32(ctx, efin, efin, n + 1, nullptr));
  for (int phase = 0; phase < 2; phase++)
    hipLaunchKernelGGL(k_db_out, dim3(vcp_blocks(n, DT)), dim3(DT), 0, st, n, d_mask, d_in_classed, pos, R, efin, min_pts, cf_in,
                       d_labels, d_is_core, d_is_classed, ctr, phase);
  VCP_HIP(ctx, hipGetLastError());
  VCP_HIP(ctx, hipMemcpyAsync(hc, ctr, 8 * 8, hipMemcpyDeviceToHost, st));
  VCP_TRY(vcp_phase_finish(ctx));
  VCP_HIP(ctx, hipStreamSynchronize(st));
#undef VCP_DB_PAIRS
  if (cf_out) *cf_out = cf_in + (int32_t)hc[3];
  if (dist_evals) *dist_evals = (int64_t)hc[4] * (int64_t)m;
  return VCP_OK;
}
