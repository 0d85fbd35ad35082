// icp.hip -- ICP.go_hell_ICP on MI355X (gfx950).
//
// Per round ONE fused pass over the data (BaseClass/ICP.cs:195-219 TransPoint, :224-250
// FindClosestPointSet, :255-273 means, :38-52 sum p y^T, :126-133 SSE): every thread transforms its
// points with the current R,T, scans the model brute force (wave-uniform index => the model is read
// through the scalar cache, no LDS needed for a few thousand points), and keeps 16 binary64 partial sums.
// Sums are reduced wave -> block -> a fixed-order pass over the block partials (no float atomics: the
// result is run-to-run deterministic).  The 16 sums go to the host, which solves Horn's closed form
// (the INTENDED arithmetic of :53-124) and composes R,T (:149-177).
//
// Algorithmic bytes: 24 B per data point per round (binary64 xyz read once; the model stays in cache).
#include <cmath>
#include <cstring>
#include <vector>

#include "vcp_ctx.hpp"

namespace {

constexpr int ITPB = 256;
constexpr int ICP_MAX_BLOCKS = 1024;

struct Xf {
  double R[9];
  double T[3];
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}

__global__ __launch_bounds__(ITPB) void k_icp_pass(const double* __restrict__ model, int nm,
                                                  const double* __restrict__ data, int64_t nd, Xf xf,
                                                  double* __restrict__ partial, int32_t* __restrict__ nn) {
  double s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * ITPB + threadIdx.x; i < nd; i += (int64_t)gridDim.x * ITPB) {
    const double d0 = data[3 * i], d1 = data[3 * i + 1], d2 = data[3 * i + 2];
    // TransPoint: r = R*p accumulated k ascending from 0 (Matrix.StupidMultiply), then + T
    double p[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double acc = 0.0;
      acc += xf.R[3 * r] * d0;
      acc += xf.R[3 * r + 1] * d1;
      acc += xf.R[3 * r + 2] * d2;
      p[r] = acc + xf.T[r];
    }
    // FindClosestPointSet: strict <, lowest model index wins ties
    double best = (p[0] - model[0]) * (p[0] - model[0]) + (p[1] - model[1]) * (p[1] - model[1]) +
                  (p[2] - model[2]) * (p[2] - model[2]);
    int order = 0;
    for (int j = 1; j < nm; j++) {
      const double m0 = model[3 * j], m1 = model[3 * j + 1], m2 = model[3 * j + 2];
      double dd = (p[0] - m0) * (p[0] - m0) + (p[1] - m1) * (p[1] - m1) + (p[2] - m2) * (p[2] - m2);
      if (dd < best) {
        best = dd;
        order = j;
      }
    }
    if (nn) nn[i] = order;
    const double y0 = model[3 * order], y1 = model[3 * order + 1], y2 = model[3 * order + 2];
    const double y[3] = {y0, y1, y2};
#pragma unroll
    for (int r = 0; r < 3; r++) {
      s[r] += p[r];
      s[3 + r] += y[r];
#pragma unroll
      for (int c = 0; c < 3; c++) s[6 + 3 * r + c] += p[r] * y[c];
    }
    const double e0 = p[0] - y0, e1 = p[1] - y1, e2 = p[2] - y2;
    s[15] += e0 * e0 + e1 * e1 + e2 * e2;
  }
  __shared__ double sm[ITPB / 64][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    double v = wave_sum(s[k]);
    if (lane == 0) sm[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double v = sm[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < ITPB / 64; k++) v += sm[k][threadIdx.x];
    partial[(size_t)blockIdx.x * 16 + threadIdx.x] = v;
  }
}

// fixed-order reduction of the block partials: thread t adds blocks t, t+256, ... in order, then a fixed
// shuffle/LDS tree over the 256 threads (same order every run -> bitwise reproducible sums)
__global__ __launch_bounds__(ITPB) void k_icp_final(const double* __restrict__ partial, int nb, double* __restrict__ out) {
  double s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = 0.0;
  for (int b = threadIdx.x; b < nb; b += ITPB) {
    const double2* row = reinterpret_cast<const double2*>(partial + (size_t)b * 16);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      double2 v = row[k];
      s[2 * k] += v.x;
      s[2 * k + 1] += v.y;
    }
  }
  __shared__ double sm[ITPB / 64][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    double v = wave_sum(s[k]);
    if (lane == 0) sm[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double v = sm[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < ITPB / 64; k++) v += sm[k][threadIdx.x];
    out[threadIdx.x] = v;
  }
}

// ---- host: Horn's unit-quaternion closed form ---------------------------------------------------
// cyclic Jacobi sweeps on a symmetric 4x4 (independent of the oracle's max-pivot variant)
void jacobi4(double A[4][4], double V[4][4]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < 4; i++) {
      diag += A[i][i] * A[i][i];
      for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
    }
    if (off <= 1e-34 * (diag + off) || off == 0.0) break;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; k++) {
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 4; k++) {
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 4; k++) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

bool horn(const double s[16], int64_t nd, double R1[9], double T1[3]) {
  const double N = (double)nd;
  double muP[3], muY[3], m[3][3];
  for (int a = 0; a < 3; a++) {
    muP[a] = s[a] / N;
    muY[a] = s[3 + a] / N;
  }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) m[r][c] = s[6 + 3 * r + c] / N - muP[r] * muY[c];
  const double tr = m[0][0] + m[1][1] + m[2][2];
  const double delta[3] = {m[1][2] - m[2][1], m[2][0] - m[0][2], m[0][1] - m[1][0]};
  double Q[4][4], V[4][4];
  Q[0][0] = tr;
  for (int i = 0; i < 3; i++) {
    Q[0][i + 1] = Q[i + 1][0] = delta[i];
    for (int j = 0; j < 3; j++) Q[i + 1][j + 1] = m[i][j] + m[j][i] - (i == j ? tr : 0.0);
  }
  jacobi4(Q, V);
  int best = 0;
  for (int i = 1; i < 4; i++)
    if (Q[i][i] > Q[best][best]) best = i;
  double q[4] = {V[0][best], V[1][best], V[2][best], V[3][best]};
  const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (!(nrm > 0.0) || !std::isfinite(nrm)) return false;
  for (int i = 0; i < 4; i++) q[i] /= nrm;
  // CalculateRotation, BaseClass/ICP.cs:274-285
  R1[0] = q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3];
  R1[1] = 2.0 * (q[1] * q[2] - q[0] * q[3]);
  R1[2] = 2.0 * (q[1] * q[3] + q[0] * q[2]);
  R1[3] = 2.0 * (q[1] * q[2] + q[0] * q[3]);
  R1[4] = q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3];
  R1[5] = 2.0 * (q[2] * q[3] - q[0] * q[1]);
  R1[6] = 2.0 * (q[1] * q[3] - q[0] * q[2]);
  R1[7] = 2.0 * (q[2] * q[3] + q[0] * q[1]);
  R1[8] = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
  for (int i = 0; i < 3; i++)
    T1[i] = muY[i] - (R1[3 * i] * muP[0] + R1[3 * i + 1] * muP[1] + R1[3 * i + 2] * muP[2]);
  return true;
}

int icp_pass(vcp_ctx* ctx, const double* d_model, int64_t nm, const double* d_data, int64_t nd, const Xf& xf,
             double sums[16], int32_t* d_nn) {
  hipStream_t st = ctx->stream;
  int nb = (int)vcp_blocks(nd, ITPB, ICP_MAX_BLOCKS);
  VCP_TRY(vcp_ensure(ctx, ctx->b_icp_part, (size_t)(ICP_MAX_BLOCKS + 1) * 16 * sizeof(double)));
  double* part = ctx->b_icp_part.as<double>();
  double* out = part + (size_t)ICP_MAX_BLOCKS * 16;
  hipLaunchKernelGGL(k_icp_pass, dim3(nb), dim3(ITPB), 0, st, d_model, (int)nm, d_data, nd, xf, part, d_nn);
  hipLaunchKernelGGL(k_icp_final, dim3(1), dim3(ITPB), 0, st, part, nb, out);
  double* h = reinterpret_cast<double*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(h, out, 16 * sizeof(double), hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  std::memcpy(sums, h, 16 * sizeof(double));
  return VCP_OK;
}

void identity(Xf& xf) {
  for (int i = 0; i < 9; i++) xf.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  xf.T[0] = xf.T[1] = xf.T[2] = 0.0;
}

}  // namespace

extern "C" {

int vcp_icp_dev(vcp_ctx* ctx, const double* d_model, int64_t nm, const double* d_data, int64_t nd, double tol,
                int max_iter, int stop_rule, double R[9], double T[3], double* sse_o, double* rmse_o,
                int32_t* iters_o) {
  if (!ctx) return VCP_ERR_ARG;
  if (nm <= 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "empty model (model[0], BaseClass/ICP.cs:233)");
  if (nd < 0 || max_iter < 1 || !R || !T) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (nm >= 0x7FFFFFFFLL || nd >= ((int64_t)1 << 40)) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "too many points");
  if (stop_rule != VCP_STOP_SSE_DELTA && stop_rule != VCP_STOP_RMSE) return vcp_fail(ctx, VCP_ERR_ARG, "stop_rule");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  vcp_phase(ctx, "icp_rounds");
  Xf xf;
  identity(xf);  // round 1 matches the raw data (P = copy of data, :22); 1*x + 0*y + 0*z is exact
  double pre_d = 0.0, d = 0.0;
  int round = 0;
  bool go;
  do {
    pre_d = d;
    double s[16] = {0};
    double R1[9], T1[3];
    bool ok = true;
    if (nd > 0) {
      VCP_TRY(icp_pass(ctx, d_model, nm, d_data, nd, xf, s, nullptr));
      ok = horn(s, nd, R1, T1);
    }
    d = s[15];
    round++;
    if (stop_rule == VCP_STOP_RMSE)
      go = nd > 0 && std::sqrt(d / (double)nd) >= tol;
    else
      go = std::fabs(d - pre_d) >= tol;  // BaseClass/ICP.cs:149,180
    if (go && nd > 0) {
      if (!ok) return vcp_fail(ctx, VCP_ERR_ARG, "Horn solve failed (non-finite sums)");
      if (round == 1) {
        std::memcpy(R, R1, sizeof(R1));
        std::memcpy(T, T1, sizeof(T1));
      } else {
        double tR[9], tT[3];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) {
            double acc = 0.0;
            for (int k = 0; k < 3; k++) acc += R1[3 * i + k] * R[3 * k + j];  // R1 * R, :167
            tR[3 * i + j] = acc;
          }
        for (int i = 0; i < 3; i++) {
          double acc = 0.0;
          for (int k = 0; k < 3; k++) acc += R1[3 * i + k] * T[k];  // R1 * T, :168
          tT[i] = acc + T1[i];
        }
        std::memcpy(R, tR, sizeof(tR));
        std::memcpy(T, tT, sizeof(tT));
      }
      std::memcpy(xf.R, R, sizeof(xf.R));
      std::memcpy(xf.T, T, sizeof(xf.T));
    }
  } while (go && round < max_iter);
  VCP_TRY(vcp_phase_finish(ctx));
  if (sse_o) *sse_o = d;
  if (rmse_o) *rmse_o = nd > 0 ? std::sqrt(d / (double)nd) : 0.0;
  if (iters_o) *iters_o = round;
  return VCP_OK;
}

int vcp_icp(vcp_ctx* ctx, const double* model, int64_t nm, const double* data, int64_t nd, double tol, int max_iter,
            int stop_rule, double R[9], double T[3], double* sse, double* rmse, int32_t* iters) {
  if (!ctx) return VCP_ERR_ARG;
  if (nm <= 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "empty model (model[0], BaseClass/ICP.cs:233)");
  if (nd < 0 || !model || (nd > 0 && !data)) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  VCP_TRY(vcp_bind(ctx));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)nm * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, (size_t)(nd > 0 ? nd : 1) * 24));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, model, (size_t)nm * 24, hipMemcpyHostToDevice, ctx->stream));
  if (nd > 0) VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, data, (size_t)nd * 24, hipMemcpyHostToDevice, ctx->stream));
  return vcp_icp_dev(ctx, ctx->b_in0.as<double>(), nm, ctx->b_in2.as<double>(), nd, tol, max_iter, stop_rule, R, T,
                     sse, rmse, iters);
}

// "VTK-like" configuration of the same loop: what MainForm.ICP() asks of
// vtkIterativeClosestPointTransform (FrmMain.cs:851-862: RigidBody, 100 iterations, StartByMatchingCentroidsOn,
// no mean-distance check), following the VTK 5.0 header (vtkIterativeClosestPointTransform.h:49-180): landmarks =
// every step-th source point (step = ns / max_landmarks when ns > max_landmarks), optional initial translation
// target centroid - source centroid, max_iter rounds, accumulated 4x4 matrix.  VTK's sources are not in the
// reference tree: behaviour per the header only, PARITY UNPINNED against VTK itself.
int vcp_icp_vtklike(vcp_ctx* ctx, const double* source, int64_t ns, const double* target, int64_t nt, int max_iter,
                    int max_landmarks, int start_by_matching_centroids, double M[16], double* mean_dist,
                    int32_t* iters_o) {
  if (!ctx) return VCP_ERR_ARG;
  if (ns <= 0 || nt <= 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "empty source or target");
  if (max_iter < 1 || max_landmarks < 1 || !source || !target || !M) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (nt >= 0x7FFFFFFFLL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "target too large");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  int64_t step = 1;
  if (ns > max_landmarks) step = ns / max_landmarks;
  const int64_t nb = ns / step;
  std::vector<double> a((size_t)3 * nb);
  for (int64_t i = 0, j = 0; i < nb; i++, j += step)
    for (int c = 0; c < 3; c++) a[3 * i + c] = source[3 * j + c];
  Xf xf;
  identity(xf);
  if (start_by_matching_centroids) {  // sequential binary64 means over ALL points of both sets
    double cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
    for (int64_t i = 0; i < ns; i++)
      for (int c = 0; c < 3; c++) cs[c] += source[3 * i + c];
    for (int64_t i = 0; i < nt; i++)
      for (int c = 0; c < 3; c++) ct[c] += target[3 * i + c];
    for (int c = 0; c < 3; c++) xf.T[c] = ct[c] / (double)nt - cs[c] / (double)ns;
  }
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)nt * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, (size_t)nb * 24));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, target, (size_t)nt * 24, hipMemcpyHostToDevice, ctx->stream));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, a.data(), (size_t)nb * 24, hipMemcpyHostToDevice, ctx->stream));
  int it = 0;
  double md = 0;
  for (;;) {
    double s[16], R1[9], T1[3];
    VCP_TRY(icp_pass(ctx, ctx->b_in0.as<double>(), nt, ctx->b_in2.as<double>(), nb, xf, s, nullptr));
    if (!horn(s, nb, R1, T1)) return vcp_fail(ctx, VCP_ERR_ARG, "Horn solve failed (non-finite sums)");
    double tR[9], tT[3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double acc = 0.0;
        for (int k = 0; k < 3; k++) acc += R1[3 * i + k] * xf.R[3 * k + j];
        tR[3 * i + j] = acc;
      }
    for (int i = 0; i < 3; i++) {
      double acc = 0.0;
      for (int k = 0; k < 3; k++) acc += R1[3 * i + k] * xf.T[k];
      tT[i] = acc + T1[i];
    }
    std::memcpy(xf.R, tR, sizeof(tR));
    std::memcpy(xf.T, tT, sizeof(tT));
    md = std::sqrt(s[15] / (double)nb);
    it++;
    if (it >= max_iter) break;
  }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) M[4 * r + c] = xf.R[3 * r + c];
    M[4 * r + 3] = xf.T[r];
  }
  M[12] = M[13] = M[14] = 0;
  M[15] = 1;
  if (mean_dist) *mean_dist = md;
  if (iters_o) *iters_o = it;
  return VCP_OK;
}

int vcp_icp_sums(vcp_ctx* ctx, const double* model, int64_t nm, const double* data, int64_t nd, const double R[9],
                 const double T[3], double sums[16], int32_t* nn) {
  if (!ctx) return VCP_ERR_ARG;
  if (nm <= 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "empty model");
  if (nd <= 0 || !model || !data || !sums) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (nm >= 0x7FFFFFFFLL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "model too large");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)nm * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, (size_t)nd * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, (size_t)nd * 4));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, model, (size_t)nm * 24, hipMemcpyHostToDevice, ctx->stream));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, data, (size_t)nd * 24, hipMemcpyHostToDevice, ctx->stream));
  Xf xf;
  identity(xf);
  if (R) std::memcpy(xf.R, R, sizeof(xf.R));
  if (T) std::memcpy(xf.T, T, sizeof(xf.T));
  VCP_TRY(icp_pass(ctx, ctx->b_in0.as<double>(), nm, ctx->b_in2.as<double>(), nd, xf, sums,
                   nn ? ctx->b_out0.as<int32_t>() : nullptr));
  if (nn) {
    VCP_HIP(ctx, hipMemcpyAsync(nn, ctx->b_out0.p, (size_t)nd * 4, hipMemcpyDeviceToHost, ctx->stream));
    VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return VCP_OK;
}

}  // extern "C"
