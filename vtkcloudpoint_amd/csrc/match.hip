// match.hip -- MainForm.calMatchedCoords + RecorrectMatchingPtsByDistance on MI355X.
//
// FrmMain.cs:3572-3587: matched = M * (tmp_X, tmp_Y, tmp_Z, 1) (row by row, left to right, no FMA);
// FrmMain.cs:3588-3618 with getDisP :829-835: nearest truth point by sqrt(dx^2+dy^2+dz^2) (binary64,
// correctly rounded sqrt), strict `<` so the lowest index wins ties, matched iff distance < max_dist.
// One thread per centroid; the truth index is wave-uniform, so truths are read through the scalar cache.
#include <algorithm>
#include <cmath>
#include <vector>

#include "nngrid.hpp"
#include <cstring>
#include "vcp_ctx.hpp"

namespace {
constexpr int MT = 128;

struct M16 {
  double m[16];
};

// GRID: the truths have been binned (nngrid.hpp); the search compares (sqrt distance, truth index) pairs, i.e. returns
// what the sequential strict-`<` scan over the correctly rounded distances returns.
template <bool GRID>
__global__ __launch_bounds__(MT) void k_match(const double* __restrict__ centers, int K, const double* __restrict__ truths,
                                             int T, M16 M, double max_dist, double* __restrict__ mxyz,
                                             uint8_t* __restrict__ is_matched, int32_t* __restrict__ nearest,
                                             double* __restrict__ ndist, uint32_t* __restrict__ count, NNGrid ng) {
  constexpr int LPQ = GRID ? nng::NNG : 1;  // lanes per centroid (they split the rows of its search block)
  const int j = (int)(((int64_t)blockIdx.x * MT + threadIdx.x) / LPQ);
  const int sub = (int)(threadIdx.x & (LPQ - 1));
  bool hit = false;
  if (j < K) {
    const double c0 = centers[3 * j], c1 = centers[3 * j + 1], c2 = centers[3 * j + 2];
    double m[3];
#pragma unroll
    for (int r = 0; r < 3; r++) m[r] = c0 * M.m[4 * r] + c1 * M.m[4 * r + 1] + c2 * M.m[4 * r + 2] + M.m[4 * r + 3];
    if (mxyz && sub == 0) {
      mxyz[3 * j] = m[0];
      mxyz[3 * j + 1] = m[1];
      mxyz[3 * j + 2] = m[2];
    }
    int best = 0;
    double bd;
    if (GRID) {
      nng::query<true>(ng, m, sub, best, bd);
      // the distance the C# holds for the winner (NaN / infinity included: the query only orders finite values)
      double dx = truths[3 * best] - m[0], dy = truths[3 * best + 1] - m[1], dz = truths[3 * best + 2] - m[2];
      bd = sqrt(dx * dx + dy * dy + dz * dz);
    } else {
      {
        double dx = truths[0] - m[0], dy = truths[1] - m[1], dz = truths[2] - m[2];
        bd = sqrt(dx * dx + dy * dy + dz * dz);
      }
      for (int i = 1; i < T; i++) {
        double dx = truths[3 * i] - m[0], dy = truths[3 * i + 1] - m[1], dz = truths[3 * i + 2] - m[2];
        double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (d < bd) {
          bd = d;
          best = i;
        }
      }
    }
    hit = bd < max_dist && sub == 0;
    if (sub == 0) {
      nearest[j] = best;
      if (ndist) ndist[j] = bd;
      is_matched[j] = hit ? 1 : 0;
    }
  }
  unsigned long long b = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (uint32_t)__popcll(b));
}
// MainForm.refreshClusList (FrmMain.cs:3437-3467): nearest truth within `radius` per raw point; among equal
// distances the LAST truth in list order wins (OrderByDescending + Reverse), id 0 = none.
__global__ __launch_bounds__(256) void k_assign_truths(const double* __restrict__ motor, int64_t n,
                                                      const double* __restrict__ txy, const int32_t* __restrict__ tid,
                                                      int T, double radius, int32_t* __restrict__ ids,
                                                      unsigned long long* __restrict__ outliers) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool none = false;
  if (i < n) {
    const double2 p = *reinterpret_cast<const double2*>(motor + 2 * i);
    int32_t id = 0;
    double best = 0;
    bool have = false;
    for (int s = 0; s < T; s++) {
      const double ax = txy[2 * s] - p.x, ay = txy[2 * s + 1] - p.y;
      const double d = sqrt(ax * ax + ay * ay);
      if (d < radius && (!have || d <= best)) {
        best = d;
        id = tid[s];
        have = true;
      }
    }
    ids[i] = id;
    none = id == 0;
  }
  __shared__ unsigned wc[4];
  unsigned long long b = __ballot(none);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = (unsigned)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = wc[0] + wc[1] + wc[2] + wc[3];
    if (t) atomicAdd(&outliers[blockIdx.x & 31], (unsigned long long)t);
  }
}
// The same query through a grid over the TRUTHS (cell edge >= radius): a raw point only meets the truths of its
// 3 x 3 cells, so the cost is O(n) instead of O(n T).  Candidates are evaluated with the identical binary64
// expression; "last truth in list order among equal distances" is carried by the original truth index.
struct TruthGrid {
  double x0, y0, inv_h;
  int Dx, Dy;
};
__global__ __launch_bounds__(256) void k_assign_truths_grid(const double* __restrict__ motor, int64_t n, TruthGrid g,
                                                           const uint32_t* __restrict__ cellstart,
                                                           const double* __restrict__ txy, const int32_t* __restrict__ tid,
                                                           const int32_t* __restrict__ torig, double radius,
                                                           int32_t* __restrict__ ids,
                                                           unsigned long long* __restrict__ outliers) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool none = false;
  if (i < n) {
    const double2 p = *reinterpret_cast<const double2*>(motor + 2 * i);
    int32_t id = 0, borig = -1;
    double best = 0;
    const double ux = (p.x - g.x0) * g.inv_h, uy = (p.y - g.y0) * g.inv_h;
    // a point more than one cell outside the truths' bounding box (or NaN) has no truth within the radius
    if (ux >= -1.0 && ux < (double)g.Dx + 1.0 && uy >= -1.0 && uy < (double)g.Dy + 1.0) {
      const int cx = (int)floor(ux), cy = (int)floor(uy);
      const int xa = max(cx - 1, 0), xb = min(cx + 1, g.Dx - 1);
      for (int y = max(cy - 1, 0); y <= min(cy + 1, g.Dy - 1); y++) {
        if (xa > xb) break;
        const uint32_t s0 = cellstart[(size_t)y * g.Dx + xa], s1 = cellstart[(size_t)y * g.Dx + xb + 1];
        for (uint32_t s = s0; s < s1; s++) {
          const double ax = txy[2 * s] - p.x, ay = txy[2 * s + 1] - p.y;
          const double d = sqrt(ax * ax + ay * ay);
          if (d < radius && (borig < 0 || d < best || (d == best && torig[s] > borig))) {
            best = d;
            id = tid[s];
            borig = torig[s];
          }
        }
      }
    }
    ids[i] = id;
    none = id == 0;
  }
  __shared__ unsigned wc[4];
  unsigned long long b = __ballot(none);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = (unsigned)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = wc[0] + wc[1] + wc[2] + wc[3];
    if (t) atomicAdd(&outliers[blockIdx.x & 31], (unsigned long long)t);
  }
}

// host side of the grid: counting sort of the finite truths by cell.  Returns false when the grid form does not
// apply (radius not a positive finite number, no finite truth, or degenerate extents) -> brute force.
bool build_truth_grid(const double* txy, const int32_t* tid, int32_t T, double radius, TruthGrid* g,
                      std::vector<uint32_t>* cellstart, std::vector<double>* sxy, std::vector<int32_t>* sid,
                      std::vector<int32_t>* sorig) {
  if (!(radius > 0.0) || !std::isfinite(radius) || T <= 0) return false;
  double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
  int32_t nf = 0;
  for (int32_t s = 0; s < T; s++) {
    const double x = txy[2 * s], y = txy[2 * s + 1];
    if (!std::isfinite(x) || !std::isfinite(y)) continue;  // never within a finite radius
    x0 = std::min(x0, x);
    x1 = std::max(x1, x);
    y0 = std::min(y0, y);
    y1 = std::max(y1, y);
    nf++;
  }
  if (nf == 0) return false;
  double h = radius * (1.0 + 1.0 / 1048576.0);
  for (int it = 0; it < 200; it++) {
    const double dx = (x1 - x0) / h, dy = (y1 - y0) / h;
    if (std::isfinite(dx) && std::isfinite(dy) && (dx + 1.0) * (dy + 1.0) <= 4194304.0) break;
    h *= 2.0;
  }
  const double dx = (x1 - x0) / h, dy = (y1 - y0) / h;
  if (!std::isfinite(h) || !std::isfinite(dx) || !std::isfinite(dy) || (dx + 1.0) * (dy + 1.0) > 4194304.0) return false;
  g->x0 = x0;
  g->y0 = y0;
  g->inv_h = 1.0 / h;
  g->Dx = (int)dx + 1;
  g->Dy = (int)dy + 1;
  const size_t nc = (size_t)g->Dx * g->Dy;
  cellstart->assign(nc + 1, 0u);
  std::vector<uint32_t> cell((size_t)T, 0xFFFFFFFFu);
  for (int32_t s = 0; s < T; s++) {
    const double x = txy[2 * s], y = txy[2 * s + 1];
    if (!std::isfinite(x) || !std::isfinite(y)) continue;
    int cx = (int)std::floor((x - x0) * g->inv_h), cy = (int)std::floor((y - y0) * g->inv_h);
    cx = std::min(std::max(cx, 0), g->Dx - 1);
    cy = std::min(std::max(cy, 0), g->Dy - 1);
    cell[s] = (uint32_t)((size_t)cy * g->Dx + cx);
    (*cellstart)[cell[s] + 1]++;
  }
  for (size_t c = 0; c < nc; c++) (*cellstart)[c + 1] += (*cellstart)[c];
  sxy->assign((size_t)nf * 2, 0.0);
  sid->assign((size_t)nf, 0);
  sorig->assign((size_t)nf, 0);
  std::vector<uint32_t> cur(cellstart->begin(), cellstart->end() - 1);
  for (int32_t s = 0; s < T; s++) {
    if (cell[s] == 0xFFFFFFFFu) continue;
    const uint32_t k = cur[cell[s]]++;
    (*sxy)[2 * (size_t)k] = txy[2 * s];
    (*sxy)[2 * (size_t)k + 1] = txy[2 * s + 1];
    (*sid)[k] = tid[s];
    (*sorig)[k] = s;
  }
  return true;
}

}  // namespace

extern "C" int vcp_assign_truths(vcp_ctx* ctx, const double* motor, int64_t n, const double* truths_xy,
                                 const int32_t* truth_ids, int32_t T, double radius, int32_t* ids, int64_t* outliers) {
  if (!ctx) return VCP_ERR_ARG;
  if (n < 0 || T < 0 || (n > 0 && (!motor || !ids)) || (T > 0 && (!truths_xy || !truth_ids)))
    return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (outliers) *outliers = 0;
  if (n == 0) return VCP_OK;
  VCP_TRY(vcp_bind(ctx));
  hipStream_t st = ctx->stream;
  const size_t tt = (size_t)(T > 0 ? T : 1);
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)n * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, tt * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in3, tt * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out2, 32 * 8));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, motor, (size_t)n * 16, hipMemcpyHostToDevice, st));
  if (T > 0) {
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, truths_xy, (size_t)T * 16, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in3.p, truth_ids, (size_t)T * 4, hipMemcpyHostToDevice, st));
  }
  VCP_HIP(ctx, hipMemsetAsync(ctx->b_out2.p, 0, 32 * 8, st));
  TruthGrid tg;
  std::vector<uint32_t> h_cellstart;
  std::vector<double> h_sxy;
  std::vector<int32_t> h_sid, h_sorig;
  if (build_truth_grid(truths_xy, truth_ids, T, radius, &tg, &h_cellstart, &h_sxy, &h_sid, &h_sorig)) {
    VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, h_cellstart.size() * 4));
    VCP_TRY(vcp_ensure(ctx, ctx->b_aux1, h_sid.size() * 4 + 16));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_aux0.p, h_cellstart.data(), h_cellstart.size() * 4, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, h_sxy.data(), h_sxy.size() * 8, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in3.p, h_sid.data(), h_sid.size() * 4, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_aux1.p, h_sorig.data(), h_sorig.size() * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_assign_truths_grid, dim3(vcp_blocks(n, 256)), dim3(256), 0, st, ctx->b_in0.as<double>(), n, tg,
                       ctx->b_aux0.as<uint32_t>(), ctx->b_in2.as<double>(), ctx->b_in3.as<int32_t>(),
                       ctx->b_aux1.as<int32_t>(), radius, ctx->b_out0.as<int32_t>(), ctx->b_out2.as<unsigned long long>());
    VCP_HIP(ctx, hipStreamSynchronize(st));  // the host vectors above are the source of the async copies
  } else {
    hipLaunchKernelGGL(k_assign_truths, dim3(vcp_blocks(n, 256)), dim3(256), 0, st, ctx->b_in0.as<double>(), n,
                       ctx->b_in2.as<double>(), ctx->b_in3.as<int32_t>(), T, radius, ctx->b_out0.as<int32_t>(),
                       ctx->b_out2.as<unsigned long long>());
  }
  VCP_HIP(ctx, hipGetLastError());
  unsigned long long* hp = reinterpret_cast<unsigned long long*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(ids, ctx->b_out0.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(hp, ctx->b_out2.p, 32 * 8, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (outliers) {
    unsigned long long t = 0;
    for (int k = 0; k < 32; k++) t += hp[k];
    *outliers = (int64_t)t;
  }
  return VCP_OK;
}

extern "C" int vcp_match(vcp_ctx* ctx, const double* centers, int32_t K, const double* truths, int32_t T,
                         const double M[16], double max_dist, double* matched_xyz, uint8_t* is_matched,
                         int32_t* nearest, double* nearest_dist, int32_t* count_matched) {
  if (!ctx) return VCP_ERR_ARG;
  if (K < 0 || T < 0 || !M) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (count_matched) *count_matched = 0;
  if (K == 0) return VCP_OK;
  if (T == 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "no truth points (truePointCloud.GetPoint(0), FrmMain.cs:3598)");
  if (!centers || !truths || !is_matched || !nearest) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(vcp_bind(ctx));
  hipStream_t st = ctx->stream;
  // device layout: [centers K*24 | truths T*24] in b_in0, [xyz K*24 | dist K*8 | nearest K*4 | count 16 | flags K] in
  // b_out0 -- one copy each way through the pinned stage (the caller's arrays are pageable: five or six separate
  // copies from/to them cost more than the search itself)
  const size_t in_c = 0, in_t = (size_t)K * 24, in_bytes = in_t + (size_t)T * 24;
  const size_t o_xyz = 0, o_dist = (size_t)K * 24, o_near = o_dist + (size_t)K * 8;
  const size_t o_cnt = (o_near + (size_t)K * 4 + 15) & ~(size_t)15, o_flag = o_cnt + 16, out_bytes = o_flag + (size_t)K;
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, in_bytes));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, out_bytes));
  char* din = ctx->b_in0.as<char>();
  char* dout = ctx->b_out0.as<char>();
  char* stage = static_cast<char*>(vcp_stage(ctx, std::max(in_bytes, out_bytes)));
  if (stage) {
    std::memcpy(stage + in_c, centers, (size_t)K * 24);
    std::memcpy(stage + in_t, truths, (size_t)T * 24);
    VCP_HIP(ctx, hipMemcpyAsync(din, stage, in_bytes, hipMemcpyHostToDevice, st));
  } else {
    VCP_HIP(ctx, hipMemcpyAsync(din + in_c, centers, (size_t)K * 24, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(din + in_t, truths, (size_t)T * 24, hipMemcpyHostToDevice, st));
  }
  const double* d_cen = reinterpret_cast<const double*>(din + in_c);
  const double* d_tru = reinterpret_cast<const double*>(din + in_t);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(dout + o_cnt);
  VCP_HIP(ctx, hipMemsetAsync(cnt, 0, 16, st));
  M16 m;
  for (int i = 0; i < 16; i++) m.m[i] = M[i];
  // long truth lists are binned once (FrmMain.cs:3588-3618 scans all of them per centroid); non-finite truths keep
  // the full scan
  NNGrid ng{};
  bool grid = false;
  if (T > 512) {
    const int grc = vcp_nngrid_build(ctx, d_tru, T, &ng);
    if (grc == VCP_OK) grid = true;
    else if (grc != VCP_ERR_UNSUPPORTED) return grc;
  }
  double* o_x = reinterpret_cast<double*>(dout + o_xyz);
  uint8_t* o_f = reinterpret_cast<uint8_t*>(dout + o_flag);
  int32_t* o_n = reinterpret_cast<int32_t*>(dout + o_near);
  double* o_d = reinterpret_cast<double*>(dout + o_dist);
  if (grid)
    hipLaunchKernelGGL(k_match<true>, dim3(vcp_blocks((int64_t)K * nng::NNG, MT)), dim3(MT), 0, st, d_cen, K, d_tru, T, m,
                       max_dist, o_x, o_f, o_n, o_d, cnt, ng);
  else
    hipLaunchKernelGGL(k_match<false>, dim3(vcp_blocks(K, MT)), dim3(MT), 0, st, d_cen, K, d_tru, T, m, max_dist, o_x,
                       o_f, o_n, o_d, cnt, ng);
  VCP_HIP(ctx, hipGetLastError());
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  if (stage) {
    VCP_HIP(ctx, hipMemcpyAsync(stage, dout, out_bytes, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    if (matched_xyz) std::memcpy(matched_xyz, stage + o_xyz, (size_t)K * 24);
    std::memcpy(is_matched, stage + o_flag, (size_t)K);
    std::memcpy(nearest, stage + o_near, (size_t)K * 4);
    if (nearest_dist) std::memcpy(nearest_dist, stage + o_dist, (size_t)K * 8);
    hp[0] = *reinterpret_cast<const uint32_t*>(stage + o_cnt);
  } else {
    if (matched_xyz) VCP_HIP(ctx, hipMemcpyAsync(matched_xyz, o_x, (size_t)K * 24, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipMemcpyAsync(is_matched, o_f, (size_t)K, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipMemcpyAsync(nearest, o_n, (size_t)K * 4, hipMemcpyDeviceToHost, st));
    if (nearest_dist) VCP_HIP(ctx, hipMemcpyAsync(nearest_dist, o_d, (size_t)K * 8, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipMemcpyAsync(hp, cnt, 4, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
  }
  if (count_matched) *count_matched = (int32_t)hp[0];
  return VCP_OK;
}
