// dbpairs.hip -- the v1.0 class DB (BaseClass/DB.cs:14-115) for the inputs its 1-D formulation (dbdead.hip) cannot take:
// coordinates that share no binary grid with a pair inside the rounding band of the threshold, e < 0 or NaN, non-finite
// coordinates.  The predicate is evaluated pair by pair exactly as the C# writes it,
//     (p1.X - p2.X) + (p1.Y - p2.Y) <= e      (DB.cs:14-25, p1 = the point whose neighbourhood is asked for),
// two rounded differences and their rounded sum, whatever it then means geometrically.  O(n^2) evaluations on the GPU:
//   count   every shown point against every shown point: who has >= minPts points in its (one-sided) neighbourhood
//   seeds   the main loop (:92-115) walks the list: the next seed is the first shown point from there on that is not classed
//           and has such a neighbourhood (a point without one costs a query and changes nothing)
//   expand  expandCluster (:57-91) as breadth-first levels: everything in the neighbourhood of a frontier point takes the
//           cluster's id (:87 relabels unconditionally); what was not classed before becomes classed, is queried, and joins
//           the next frontier if its own neighbourhood is large enough.  The C#'s list `nei` holds duplicates (its dedupe
//           scan compares boxed references and never matches); a duplicate finds its point classed and only repeats the
//           same label, so the set formulation is the C#'s result, and every point is expanded at most once.
//   iritatorNum (:19) = #shown x (points not classed when the main loop reaches them + points a cluster reaches unclassed).
// One host read-back per seed and per level: a fallback for the rare cloud, bounded to 2^21 points (4 x 10^12 evaluations).
#include <cmath>

#include "dbscan_engine.hpp"

namespace {
constexpr int PT = 256;
constexpr int TILE = 1024;
constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ __forceinline__ bool db_near(double xp, double yp, double xi, double yi, double eps) {
  const double dx = xp - xi, dy = yp - yi;  // (-ffp-contract=off: three roundings, like the C#)
  return dx + dy <= eps;
}

__global__ __launch_bounds__(PT) void k_dbp_init(int64_t n, const uint8_t* __restrict__ in_classed, uint8_t* __restrict__ classed,
                                                uint32_t* __restrict__ firstseed, int32_t* __restrict__ labels,
                                                unsigned long long* __restrict__ ctr) {
  const int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x;
  if (i < 8) ctr[i] = 0ull;
  if (i >= n) return;
  classed[i] = in_classed ? (in_classed[i] ? 1 : 0) : 0;
  firstseed[i] = NONE;
  if (!in_classed) labels[i] = 0;  // (with isClassed the labels are in/out: a point no cluster reaches keeps its id)
}

// core[p] = shown and |{shown i : near(p, i)}| >= minPts; ctr[0] += shown points
__global__ __launch_bounds__(PT) void k_dbp_count(const double* __restrict__ c, int64_t n, int stride,
                                                 const uint8_t* __restrict__ mask, double eps, int min_pts,
                                                 uint8_t* __restrict__ core, unsigned long long* __restrict__ ctr) {
  __shared__ double tx[TILE], ty[TILE];
  const int64_t p = (int64_t)blockIdx.x * PT + threadIdx.x;
  const bool live = p < n && !(mask && !mask[p]);
  const double xp = live ? c[p * stride] : 0.0, yp = live ? c[p * stride + 1] : 0.0;
  unsigned long long cnt = 0;
  for (int64_t t0 = 0; t0 < n; t0 += TILE) {
    __syncthreads();
    for (int k = threadIdx.x; k < TILE; k += PT) {
      const int64_t i = t0 + k;
      const bool s = i < n && !(mask && !mask[i]);
      tx[k] = s ? c[i * stride] : NAN;  // (a NaN is in nobody's neighbourhood: a point that is not shown)
      ty[k] = s ? c[i * stride + 1] : NAN;
    }
    __syncthreads();
    if (live) {
      const int lim = (int)min((int64_t)TILE, n - t0);
#pragma unroll 4
      for (int k = 0; k < lim; k++) cnt += db_near(xp, yp, tx[k], ty[k], eps) ? 1ull : 0ull;
    }
  }
  if (p < n) core[p] = (live && (long long)cnt >= (long long)min_pts) ? 1 : 0;
  const unsigned long long m = __ballot(live);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&ctr[0], (unsigned long long)__popcll(m));
}

// next[0] <- the first i >= from that is shown, not classed and has a large enough neighbourhood (NONE: none)
__global__ __launch_bounds__(PT) void k_dbp_next(int64_t n, uint32_t from, const uint8_t* __restrict__ mask,
                                                const uint8_t* __restrict__ classed, const uint8_t* __restrict__ core,
                                                uint32_t* __restrict__ next) {
  const int64_t i = (int64_t)from + (int64_t)blockIdx.x * PT + threadIdx.x;
  const bool ok = i < n && !(mask && !mask[i]) && !classed[i] && core[i];
  const unsigned long long m = __ballot(ok);
  if (m && (threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicMin(next, (uint32_t)i);
}

// one level: every shown point against the frontier.  first: the frontier is the seed alone, which takes the id itself
__global__ __launch_bounds__(PT) void k_dbp_level(const double* __restrict__ c, int64_t n, int stride,
                                                 const uint8_t* __restrict__ mask, double eps, const uint32_t* __restrict__ front,
                                                 uint32_t nf, uint32_t seed, int32_t cid, int first,
                                                 const uint8_t* __restrict__ core, uint8_t* __restrict__ classed,
                                                 uint32_t* __restrict__ firstseed, int32_t* __restrict__ labels,
                                                 uint32_t* __restrict__ nextfront, uint32_t* __restrict__ nnext,
                                                 unsigned long long* __restrict__ ctr) {
  __shared__ double fx[TILE], fy[TILE];
  const int64_t j = (int64_t)blockIdx.x * PT + threadIdx.x;
  const bool live = j < n && !(mask && !mask[j]);
  const double xj = live ? c[j * stride] : 0.0, yj = live ? c[j * stride + 1] : 0.0;
  bool hit = false;
  for (uint32_t t0 = 0; t0 < nf; t0 += TILE) {
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < TILE && t0 + k < nf; k += PT) {
      const uint32_t f = front[t0 + k];
      fx[k] = c[(int64_t)f * stride];
      fy[k] = c[(int64_t)f * stride + 1];
    }
    __syncthreads();
    if (live && !hit) {
      const uint32_t lim = min((uint32_t)TILE, nf - t0);
      for (uint32_t k = 0; k < lim && !hit; k++) hit = db_near(fx[k], fy[k], xj, yj, eps);  // the frontier point asks
    }
  }
  if (first && j == (int64_t)seed) labels[j] = cid;  // DB.cs:59 (before the walk over nei; also when it is not its own neighbour)
  if (!hit) return;
  labels[j] = cid;  // :87
  if (!classed[j]) {
    classed[j] = 1;  // :64-65
    firstseed[j] = seed;
    atomicAdd(&ctr[1], 1ull);  // queried (:66)
    if (core[j]) nextfront[atomicAdd(nnext, 1u)] = (uint32_t)j;
  }
}

// queries of the main loop: shown points not classed on entry that no cluster seeded before them had reached; outputs
__global__ __launch_bounds__(PT) void k_dbp_final(int64_t n, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ in_classed,
                                                 const uint8_t* __restrict__ core, const uint8_t* __restrict__ classed,
                                                 const uint32_t* __restrict__ firstseed, uint8_t* __restrict__ is_core,
                                                 uint8_t* __restrict__ is_classed, unsigned long long* __restrict__ ctr) {
  const int64_t i = (int64_t)blockIdx.x * PT + threadIdx.x;
  bool turn = false;
  if (i < n) {
    const bool shown = !(mask && !mask[i]);
    const bool cls0 = in_classed && in_classed[i];
    turn = shown && !cls0 && firstseed[i] >= (uint32_t)i;
    if (is_core) is_core[i] = (shown && !cls0 && core[i]) ? 1 : 0;  // isKeyPoint ran on it (at its turn or when reached)
    if (is_classed) is_classed[i] = classed[i];
  }
  const unsigned long long m = __ballot(turn);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&ctr[2], (unsigned long long)__popcll(m));
}
}  // namespace

int vcp_db_pairs_engine(vcp_ctx* ctx, const double* d_coords, int64_t n, int stride, double eps, int min_pts, int32_t cf_in,
                        const uint8_t* d_mask, const uint8_t* d_in_classed, int32_t* d_labels, uint8_t* d_is_core,
                        uint8_t* d_is_classed, int32_t* cf_out, int64_t* dist_evals) {
  if (n > ((int64_t)1 << 21))
    return vcp_fail(ctx, VCP_ERR_UNSUPPORTED,
                    "DB (BaseClass/DB.cs): this cloud needs the pair-by-pair form (the signed-sum relation is not provably 1-D "
                    "here, or e < 0 / NaN / non-finite coordinates), which is limited to 2^21 points");
  hipStream_t st = ctx->stream;
  const size_t N1 = (size_t)n + 8;
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, N1 * 4 * 3));  // firstseed, two frontiers
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux1, N1 * 2));      // core, classed
  VCP_TRY(vcp_ensure(ctx, ctx->b_misc, 64 * 8));
  uint32_t* firstseed = ctx->b_aux0.as<uint32_t>();
  uint32_t* front[2] = {firstseed + N1, firstseed + 2 * N1};
  uint8_t* core = ctx->b_aux1.as<uint8_t>();
  uint8_t* classed = core + N1;
  unsigned long long* ctr = reinterpret_cast<unsigned long long*>(ctx->b_misc.p);  // [0] shown, [1] reached unclassed, [2] turns
  uint32_t* d_word = reinterpret_cast<uint32_t*>(ctr + 8);                          // next seed / size of the next frontier
  uint32_t* h_word = reinterpret_cast<uint32_t*>(ctx->pinned) + 512;
  const unsigned nb = vcp_blocks(n, PT);
  vcp_phase(ctx, "db_pairs_count");
  hipLaunchKernelGGL(k_dbp_init, dim3(nb), dim3(PT), 0, st, n, d_in_classed, classed, firstseed, d_labels, ctr);
  hipLaunchKernelGGL(k_dbp_count, dim3(nb), dim3(PT), 0, st, d_coords, n, stride, d_mask, eps, min_pts, core, ctr);
  vcp_phase(ctx, "db_pairs_clusters");
  int32_t K = 0;
  uint32_t from = 0;
  while ((int64_t)from < n) {
    VCP_HIP(ctx, hipMemsetAsync(d_word, 0xFF, 4, st));
    hipLaunchKernelGGL(k_dbp_next, dim3(vcp_blocks(n - from, PT)), dim3(PT), 0, st, n, from, d_mask, classed, core, d_word);
    VCP_HIP(ctx, hipMemcpyAsync(h_word, d_word, 4, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t seed = h_word[0];
    if (seed == NONE) break;
    K++;
    // the frontier of the first level is the seed itself
    VCP_HIP(ctx, hipMemcpyAsync(front[0], h_word, 4, hipMemcpyHostToDevice, st));
    uint32_t nf = 1;
    int cur = 0, first = 1;
    while (nf > 0) {
      VCP_HIP(ctx, hipMemsetAsync(d_word, 0, 4, st));
      hipLaunchKernelGGL(k_dbp_level, dim3(nb), dim3(PT), 0, st, d_coords, n, stride, d_mask, eps, front[cur], nf, seed,
                         cf_in + K, first, core, classed, firstseed, d_labels, front[cur ^ 1], d_word, ctr);
      VCP_HIP(ctx, hipMemcpyAsync(h_word + 1, d_word, 4, hipMemcpyDeviceToHost, st));
      VCP_HIP(ctx, hipStreamSynchronize(st));
      nf = h_word[1];
      cur ^= 1;
      first = 0;
    }
    from = seed + 1u;
  }
  hipLaunchKernelGGL(k_dbp_final, dim3(nb), dim3(PT), 0, st, n, d_mask, d_in_classed, core, classed, firstseed, d_is_core,
                     d_is_classed, ctr);
  unsigned long long* hc = reinterpret_cast<unsigned long long*>(ctx->pinned) + 128;
  VCP_HIP(ctx, hipGetLastError());
  VCP_HIP(ctx, hipMemcpyAsync(hc, ctr, 8 * 8, hipMemcpyDeviceToHost, st));
  VCP_TRY(vcp_phase_finish(ctx));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (cf_out) *cf_out = cf_in + K;
  if (dist_evals) *dist_evals = (int64_t)((hc[1] + hc[2]) * hc[0]);
  return VCP_OK;
}
