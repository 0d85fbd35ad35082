// nngrid.hip -- binning of a static 3-D point set for nngrid.hpp's exact nearest-neighbour query.
// Bounding box -> cell edge h with about one point per cell -> per-point cell id, histogram, exclusive scan, scatter
// of (x, y, z, original index) records into cell order.  The sets are small next to the clouds (truth lists, cluster
// centroids: 10^2..10^6 points), so the histogram and the scatter use plain global atomics.
#include <algorithm>
#include <cmath>

#include "nngrid.hpp"

namespace {
constexpr int NT = 256;

// out[0..2] = min, out[3..5] = max over finite values, out[6] = number of non-finite coordinates.  Up to NBB
// workgroups; each leaves its seven partials in part[], the last one to finish (ticket in *done, which it leaves at
// zero again) folds them.
constexpr int NBB = 64;
__global__ __launch_bounds__(NT) void k_nn_bounds(const double* __restrict__ p, int64_t n, double* __restrict__ part,
                                                 uint32_t* __restrict__ done, double* __restrict__ out) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, bad = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double v = p[3 * i + a];
      if (fabs(v) <= 1.7976931348623157e308) {
        lo[a] = fmin(lo[a], v);
        hi[a] = fmax(hi[a], v);
      } else {
        bad += 1.0;
      }
    }
  }
  __shared__ double sm[NT / 64][7];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double l = lo[a], h = hi[a];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      l = fmin(l, __shfl_down(l, d, 64));
      h = fmax(h, __shfl_down(h, d, 64));
    }
    if (lane == 0) {
      sm[w][a] = l;
      sm[w][3 + a] = h;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) bad += __shfl_down(bad, d, 64);
  if (lane == 0) sm[w][6] = bad;
  __syncthreads();
  if (threadIdx.x < 7) {
    double v = sm[0][threadIdx.x];
    for (int k = 1; k < NT / 64; k++)
      v = threadIdx.x < 3 ? fmin(v, sm[k][threadIdx.x]) : threadIdx.x < 6 ? fmax(v, sm[k][threadIdx.x]) : v + sm[k][6];
    __hip_atomic_store(&part[blockIdx.x * 8 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(done, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < 7) {
    double v = __hip_atomic_load(&part[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t k = 1; k < gridDim.x; k++) {
      const double o = __hip_atomic_load(&part[k * 8 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = threadIdx.x < 3 ? fmin(v, o) : threadIdx.x < 6 ? fmax(v, o) : v + o;
    }
    out[threadIdx.x] = v;
  }
  if (threadIdx.x == 0) *done = 0;
}

__device__ __forceinline__ uint32_t cell_of(const NNGrid& g, const double* q) {
  const int cx = nng::cell1(q[0], g.mn[0], g.inv_h, g.D[0]);
  const int cy = nng::cell1(q[1], g.mn[1], g.inv_h, g.D[1]);
  const int cz = nng::cell1(q[2], g.mn[2], g.inv_h, g.D[2]);
  return ((uint32_t)cz * (uint32_t)g.D[1] + (uint32_t)cy) * (uint32_t)g.D[0] + (uint32_t)cx;
}

__global__ __launch_bounds__(NT) void k_nn_count(const double* __restrict__ p, int64_t n, NNGrid g, uint32_t* __restrict__ cid,
                                                uint32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  const double q[3] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
  const uint32_t c = cell_of(g, q);
  cid[i] = c;
  atomicAdd(&cnt[c], 1u);
}

__global__ __launch_bounds__(NT) void k_nn_scatter(const double* __restrict__ p, int64_t n, const uint32_t* __restrict__ cid,
                                                  uint32_t* __restrict__ cursor, double4* __restrict__ rec) {
  const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  const uint32_t slot = atomicAdd(&cursor[cid[i]], 1u);
  rec[slot] = make_double4(p[3 * i], p[3 * i + 1], p[3 * i + 2], __hiloint2double(0, (int)i));
}

}  // namespace

int vcp_nngrid_build(vcp_ctx* ctx, const double* d_pts, int64_t n, NNGrid* out) {
  if (n <= 0 || n >= 0x7FFFFFF0LL / 3) return vcp_fail(ctx, VCP_ERR_ARG, "nngrid: bad size");
  hipStream_t st = ctx->stream;
  // b_nn_misc: [0..8) result, [8..16) the ticket word, [16 ..) NBB x 8 partials
  const bool fresh = ctx->b_nn_misc.p == nullptr;
  VCP_TRY(vcp_ensure(ctx, ctx->b_nn_misc, (size_t)(16 + NBB * 8) * sizeof(double)));
  double* d_b = ctx->b_nn_misc.as<double>();
  if (fresh) VCP_HIP(ctx, hipMemsetAsync(d_b + 8, 0, 64, st));
  const int nbb = (int)std::min<int64_t>(NBB, (n + 4 * NT - 1) / (4 * NT));
  hipLaunchKernelGGL(k_nn_bounds, dim3(nbb), dim3(NT), 0, st, d_pts, n, d_b + 16, reinterpret_cast<uint32_t*>(d_b + 8), d_b);
  double* h = reinterpret_cast<double*>(ctx->pinned) + 64;
  VCP_HIP(ctx, hipMemcpyAsync(h, d_b, 7 * sizeof(double), hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (h[6] != 0.0) return VCP_ERR_UNSUPPORTED;  // no message: the caller keeps its full scan
  NNGrid g;
  double ext[3], vol = 1.0;
  int npos = 0;
  for (int a = 0; a < 3; a++) {
    g.mn[a] = h[a];
    ext[a] = h[3 + a] - h[a];
    if (ext[a] > 0.0 && std::isfinite(ext[a])) {
      vol *= ext[a];
      npos++;
    }
  }
  // about four cells per point over the axes that have extent (never more than 8 n + 64 cells): the sets this serves are
  // clustered -- centroids of the fragments of a few hundred blobs -- so at one point per cell on AVERAGE the occupied
  // cells hold hundreds; the doubling rings step over the empty ones cheaply
  double hh = npos ? std::pow(vol / (4.0 * (double)n), 1.0 / npos) : 1.0;
  if (!(hh > 0.0) || !std::isfinite(hh)) hh = 1.0;
  int64_t ncells = 0;
  for (int it = 0; it < 200; it++) {
    ncells = 1;
    for (int a = 0; a < 3; a++) {
      double da = ext[a] > 0.0 && std::isfinite(ext[a]) ? std::floor(ext[a] / hh) + 1.0 : 1.0;
      if (!(da < 2048.0)) da = 2048.0;
      g.D[a] = (int)da;
      ncells *= g.D[a];
    }
    bool capped = false;  // an axis at the cap with cells narrower than h would break the r*h bound: widen h instead
    for (int a = 0; a < 3; a++) capped = capped || (g.D[a] == 2048 && ext[a] / hh >= 2048.0);
    if (ncells <= 8 * n + 64 && !capped) break;
    hh *= 1.26;
  }
  g.h = hh;
  g.inv_h = 1.0 / hh;
  g.ncells = (uint32_t)ncells;
  g.n = (int)n;
  VCP_TRY(vcp_ensure(ctx, ctx->b_nn_cells, (size_t)(ncells + 2) * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_nn_cid, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_nn_rec, (size_t)n * sizeof(double4)));
  VCP_TRY(vcp_ensure(ctx, ctx->b_nn_cur, (size_t)(ncells + 2) * 4));
  uint32_t* cells = ctx->b_nn_cells.as<uint32_t>();
  uint32_t* cur = ctx->b_nn_cur.as<uint32_t>();
  uint32_t* cid = ctx->b_nn_cid.as<uint32_t>();
  double4* rec = ctx->b_nn_rec.as<double4>();
  VCP_HIP(ctx, hipMemsetAsync(cells, 0, (size_t)(ncells + 2) * 4, st));
  hipLaunchKernelGGL(k_nn_count, dim3(vcp_blocks(n, NT)), dim3(NT), 0, st, d_pts, n, g, cid, cells);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, cells, cells, ncells + 1, nullptr));
  VCP_HIP(ctx, hipMemcpyAsync(cur, cells, (size_t)(ncells + 1) * 4, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_nn_scatter, dim3(vcp_blocks(n, NT)), dim3(NT), 0, st, d_pts, n, cid, cur, rec);
  VCP_HIP(ctx, hipGetLastError());
  g.cellstart = cells;
  g.rec = rec;
  *out = g;
  return VCP_OK;
}
