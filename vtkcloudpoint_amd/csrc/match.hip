// match.hip -- MainForm.calMatchedCoords + RecorrectMatchingPtsByDistance on MI355X.
//
// FrmMain.cs:3572-3587: matched = M * (tmp_X, tmp_Y, tmp_Z, 1) (row by row, left to right, no FMA);
// FrmMain.cs:3588-3618 with getDisP :829-835: nearest truth point by sqrt(dx^2+dy^2+dz^2) (binary64,
// correctly rounded sqrt), strict `<` so the lowest index wins ties, matched iff distance < max_dist.
// One thread per centroid; the truth index is wave-uniform, so truths are read through the scalar cache.
#include "vcp_ctx.hpp"

namespace {
constexpr int MT = 128;

struct M16 {
  double m[16];
};

__global__ __launch_bounds__(MT) void k_match(const double* __restrict__ centers, int K, const double* __restrict__ truths,
                                             int T, M16 M, double max_dist, double* __restrict__ mxyz,
                                             uint8_t* __restrict__ is_matched, int32_t* __restrict__ nearest,
                                             double* __restrict__ ndist, uint32_t* __restrict__ count) {
  int j = blockIdx.x * MT + threadIdx.x;
  bool hit = false;
  if (j < K) {
    const double c0 = centers[3 * j], c1 = centers[3 * j + 1], c2 = centers[3 * j + 2];
    double m[3];
#pragma unroll
    for (int r = 0; r < 3; r++) m[r] = c0 * M.m[4 * r] + c1 * M.m[4 * r + 1] + c2 * M.m[4 * r + 2] + M.m[4 * r + 3];
    if (mxyz) {
      mxyz[3 * j] = m[0];
      mxyz[3 * j + 1] = m[1];
      mxyz[3 * j + 2] = m[2];
    }
    int best = 0;
    double bd;
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
    nearest[j] = best;
    if (ndist) ndist[j] = bd;
    hit = bd < max_dist;
    is_matched[j] = hit ? 1 : 0;
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
  hipLaunchKernelGGL(k_assign_truths, dim3(vcp_blocks(n, 256)), dim3(256), 0, st, ctx->b_in0.as<double>(), n,
                     ctx->b_in2.as<double>(), ctx->b_in3.as<int32_t>(), T, radius, ctx->b_out0.as<int32_t>(),
                     ctx->b_out2.as<unsigned long long>());
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
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)K * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, (size_t)T * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, (size_t)K * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out1, (size_t)K));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out2, (size_t)K * 8));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out3, (size_t)K * 4 + 64));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, centers, (size_t)K * 24, hipMemcpyHostToDevice, st));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, truths, (size_t)T * 24, hipMemcpyHostToDevice, st));
  uint32_t* cnt = reinterpret_cast<uint32_t*>(ctx->b_out3.as<char>() + (size_t)K * 4);
  cnt = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(cnt) + 15) & ~(uintptr_t)15);
  VCP_HIP(ctx, hipMemsetAsync(cnt, 0, 16, st));
  M16 m;
  for (int i = 0; i < 16; i++) m.m[i] = M[i];
  hipLaunchKernelGGL(k_match, dim3(vcp_blocks(K, MT)), dim3(MT), 0, st, ctx->b_in0.as<double>(), K,
                     ctx->b_in2.as<double>(), T, m, max_dist, ctx->b_out0.as<double>(), ctx->b_out1.as<uint8_t>(),
                     ctx->b_out3.as<int32_t>(), ctx->b_out2.as<double>(), cnt);
  VCP_HIP(ctx, hipGetLastError());
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  if (matched_xyz) VCP_HIP(ctx, hipMemcpyAsync(matched_xyz, ctx->b_out0.p, (size_t)K * 24, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(is_matched, ctx->b_out1.p, (size_t)K, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(nearest, ctx->b_out3.p, (size_t)K * 4, hipMemcpyDeviceToHost, st));
  if (nearest_dist) VCP_HIP(ctx, hipMemcpyAsync(nearest_dist, ctx->b_out2.p, (size_t)K * 8, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(hp, cnt, 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (count_matched) *count_matched = (int32_t)hp[0];
  return VCP_OK;
}
