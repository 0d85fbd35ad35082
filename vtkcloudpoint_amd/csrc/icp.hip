// icp.hip -- ICP.go_hell_ICP on MI355X (gfx950).
//
// Per round two launches, no host round trip:
//   k_icp_pass  one fused pass over the data (BaseClass/ICP.cs:195-219 TransPoint, :224-250 FindClosestPointSet,
//               :255-273 means, :38-52 sum p y^T, :126-133 SSE): every thread transforms its points with the
//               current R,T (read from the device-resident state), finds the nearest model point and keeps 16
//               binary64 partial sums; wave shuffle -> workgroup -> one partial row per workgroup.
//   k_icp_step  fixed-order reduction of the partial rows (bitwise reproducible, no float atomics), then ONE
//               thread solves Horn's closed form (the INTENDED arithmetic of :53-124), applies
//               the stop rule (:149,:180) and composes R <- R1 R, T <- R1 T + T1 (:149-177) in the state.
// The host enqueues rounds in batches of 8 and reads the 430-byte state back once per batch; kernels of rounds
// after the stop see state.done and return at once.
//
// Nearest neighbour: the model index is wave-uniform, so model points come through the scalar cache, four per
// trip.  Distances are first screened in binary32 (three FMAs per pair) with a rigorous rounding bound; only model points whose
// binary32 distance is within that bound of the smallest are re-evaluated in binary64, in index order with the
// C#'s strict `<` -- the chosen index is bit-identical to a full binary64 scan.
// Algorithmic bytes: 24 B per data point per round; at 1M x 100 the pass is VALU bound, not HBM bound.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "nngrid.hpp"
#include "vcp_ctx.hpp"

namespace {

constexpr int ITPB = 256;
constexpr int ICP_MAX_BLOCKS = 1024;
constexpr int ICP_BATCH = 8;  // rounds enqueued per host synchronisation

enum { MODE_REFERENCE = 0, MODE_VTK = 1, MODE_SUMS_ONLY = 2 };

struct IcpState {
  double R[9], T[3];
  double d, pre_d;
  double sums[16];
  double cen[3], mmax;  // centre of the model's bounding box and its half extent: frame and scale of the screening
  double V[16];         // eigenvector basis of the last Horn solve (all zero = none yet)
  int round, done, failed, pad;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}

// bounding box of the model -> centre and half extent in the state (single workgroup: models are small).  A model
// with non-finite coordinates gets an infinite scale: every data point is then resolved in binary64.
__global__ __launch_bounds__(ITPB) void k_model_frame(const double* __restrict__ m, int64_t nm, IcpState* __restrict__ st) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  bool bad = false;
  for (int64_t j = threadIdx.x; j < nm; j += ITPB) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double v = m[3 * j + a];
      if (!(fabs(v) <= 1.7976931348623157e308)) bad = true;
      lo[a] = fmin(lo[a], v);
      hi[a] = fmax(hi[a], v);
    }
  }
  __shared__ double sl[ITPB / 64][3], sh[ITPB / 64][3];
  __shared__ int sbad;
  if (threadIdx.x == 0) sbad = 0;
  __syncthreads();
  if (bad) sbad = 1;
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
      sl[w][a] = l;
      sh[w][a] = h;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double mm = 0.0;
    for (int a = 0; a < 3; a++) {
      double l = sl[0][a], h = sh[0][a];
      for (int k = 1; k < ITPB / 64; k++) {
        l = fmin(l, sl[k][a]);
        h = fmax(h, sh[k][a]);
      }
      const double c = 0.5 * l + 0.5 * h;
      st->cen[a] = sbad ? 0.0 : c;
      mm = fmax(mm, fmax(h - c, c - l));
    }
    st->mmax = sbad ? INFINITY : mm;
  }
}

// screening copy of the model in the centred frame: (m - c, |m - c|^2 / 2) in binary32, one 16-byte load per point
__global__ __launch_bounds__(ITPB) void k_model32(const double* __restrict__ m, int64_t nm, const IcpState* __restrict__ st,
                                                 float4* __restrict__ o) {
  int64_t j = (int64_t)blockIdx.x * ITPB + threadIdx.x;
  if (j >= nm) return;
  const double a = m[3 * j] - st->cen[0], b = m[3 * j + 1] - st->cen[1], c = m[3 * j + 2] - st->cen[2];
  o[j] = make_float4((float)a, (float)b, (float)c, (float)(0.5 * (a * a + b * b + c * c)));
}

// TB = threads per workgroup (64 for small data sets, so that they spread over more CUs).  TILED: the screening
// copy of the model is staged through LDS in tiles of MTILE points (every lane reads the same address: an LDS
// broadcast).  The scalar-cache path is the faster one while the model fits that cache (C3: 100 points); a model
// of thousands of points (ICP of cluster centroids against the truth list, MainForm.ICP's real use) makes every
// scalar load an L2 round trip, 30x slower than the tiled form.
// NNMODE 2: the model has been binned (nngrid.hpp): every lane searches the cells round its own transformed point --
// O(1) candidates per data point instead of the whole model, same exact binary64 decision and tie rule.
struct StepArgs {
  long long nd;
  double tol;
  int stop_rule, max_iter, mode;
};

constexpr int MTILE = 1024;
template <int TB, int NNMODE>
__global__ __launch_bounds__(TB) void k_icp_pass(const double* __restrict__ model, const float4* __restrict__ model32,
                                                int nm, const double* __restrict__ data, int64_t nd,
                                                const IcpState* __restrict__ st, double* __restrict__ partial,
                                                int32_t* __restrict__ nn, NNGrid ng) {
  if (st->done) return;
  constexpr bool TILED = NNMODE == 1, GRID = NNMODE == 2;
  __shared__ float4 tile[TILED ? MTILE : 1];
  double R[9], T[3];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = st->R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) T[k] = st->T[k];
  // binary32 screening on the score h_j - q.m_j (= (|q - m_j|^2 - |q|^2) / 2, h_j = |m_j|^2 / 2), q and m taken
  // relative to the centre of the model's bounding box (the score differences are translation invariant): three
  // FMAs per model point.  With S = a bound on every |coordinate| involved (this lane's |q - c| and the model's half
  // extent) and u = 2^-24 the computed score is within 27 u S^2 of the exact one (input conversions 9 u S^2, h_j
  // 4.5 u S^2, three fused roundings 13.5 u S^2), so a model point can be the binary64 winner only if
  // score <= best score + 54 u S^2; 2^-17 S^2 = 128 u S^2 is used.  The bound is per data point.
  const double cen0 = st->cen[0], cen1 = st->cen[1], cen2 = st->cen[2], mmax = st->mmax;
  double s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = 0.0;
  // GRID: nng::NNG consecutive lanes work one data point (they split the rows of its search block)
  constexpr int LPQ = GRID ? nng::NNG : 1;
  const int sub = GRID ? (int)(threadIdx.x & (LPQ - 1)) : 0;
  for (int64_t base = (int64_t)blockIdx.x * TB; base < nd * LPQ; base += (int64_t)gridDim.x * TB) {  // uniform trip count
    const int64_t i = (base + threadIdx.x) / LPQ;
    const bool live = i < nd;
    const int64_t il = live ? i : nd - 1;  // idle lanes of the last workgroup recompute the last point, unused
    const double d0 = data[3 * il], d1 = data[3 * il + 1], d2 = data[3 * il + 2];
    // TransPoint: r = R*p accumulated k ascending from 0 (Matrix.StupidMultiply), then + T
    double p[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double acc = 0.0;
      acc += R[3 * r] * d0;
      acc += R[3 * r + 1] * d1;
      acc += R[3 * r + 2] * d2;
      p[r] = acc + T[r];
    }
    int order = 0;
    if (GRID) {
      double bestv;
      nng::query<false>(ng, p, sub, order, bestv);
    } else {
    const double pc0 = p[0] - cen0, pc1 = p[1] - cen1, pc2 = p[2] - cen2;
    const float q0 = (float)pc0, q1 = (float)pc1, q2 = (float)pc2;
    const double S = fmax(mmax, fmax(fabs(pc0), fmax(fabs(pc1), fabs(pc2))));
    // 2^-17 S^2, rounded up (NaN -> all candidates; also outside binary32's normal range, where the relative rounding
    // model behind the bound does not hold)
    const float tol2r = (float)(S * S * 7.62939453125e-06) * 1.0001f;
    const float tol2 = ((float)(S * S) < 1.0e37f && tol2r >= 1.0e-30f) ? tol2r : NAN;
    // pass 1 (binary32): smallest and second smallest screened score, four model points per trip
    float b1 = INFINITY, b2 = INFINITY;
    int j1 = 0;
    int j = 0;
    if (TILED) {
      for (int t0 = 0; t0 < nm; t0 += MTILE) {
        const int cnt = min(MTILE, nm - t0);
        __syncthreads();  // the previous tile has been consumed
        for (int k = threadIdx.x; k < cnt; k += TB) tile[k] = model32[t0 + k];
        __syncthreads();
        int u = 0;
        for (; u + 3 < cnt; u += 4) {
          float4 mm[4];
#pragma unroll
          for (int v = 0; v < 4; v++) mm[v] = tile[u + v];
#pragma unroll
          for (int v = 0; v < 4; v++) {
            const float sc = __builtin_fmaf(-q0, mm[v].x, __builtin_fmaf(-q1, mm[v].y, __builtin_fmaf(-q2, mm[v].z, mm[v].w)));
            const bool lt = sc < b1;
            b2 = lt ? b1 : fminf(b2, sc);
            j1 = lt ? t0 + u + v : j1;
            b1 = lt ? sc : b1;
          }
        }
        for (; u < cnt; u++) {
          const float4 m4 = tile[u];
          const float sc = __builtin_fmaf(-q0, m4.x, __builtin_fmaf(-q1, m4.y, __builtin_fmaf(-q2, m4.z, m4.w)));
          const bool lt = sc < b1;
          b2 = lt ? b1 : fminf(b2, sc);
          j1 = lt ? t0 + u : j1;
          b1 = lt ? sc : b1;
        }
      }
      j = nm;
    }
    for (; j + 3 < nm; j += 4) {
      float4 mm[4];
#pragma unroll
      for (int u = 0; u < 4; u++) mm[u] = model32[j + u];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float sc = __builtin_fmaf(-q0, mm[u].x, __builtin_fmaf(-q1, mm[u].y, __builtin_fmaf(-q2, mm[u].z, mm[u].w)));
        const bool lt = sc < b1;
        b2 = lt ? b1 : fminf(b2, sc);
        j1 = lt ? j + u : j1;
        b1 = lt ? sc : b1;
      }
    }
    for (; j < nm; j++) {
      const float4 m4 = model32[j];
      const float sc = __builtin_fmaf(-q0, m4.x, __builtin_fmaf(-q1, m4.y, __builtin_fmaf(-q2, m4.z, m4.w)));
      const bool lt = sc < b1;
      b2 = lt ? b1 : fminf(b2, sc);
      j1 = lt ? j : j1;
      b1 = lt ? sc : b1;
    }
    order = j1;
    const bool amb = !(b2 > b1 + tol2);
    if (TILED) {
      // second sweep over the tiles for the lanes whose screening left more than one candidate; the whole
      // workgroup walks the tiles (uniform barriers), only ambiguous lanes look at them
      if (__syncthreads_or(amb ? 1 : 0)) {
        const float lim = b1 + tol2;
        const bool all = !(lim == lim);  // non-finite bound: every model point is a candidate
        double best = INFINITY;
        bool have = false;
        if (amb) order = 0;
        for (int t0 = 0; t0 < nm; t0 += MTILE) {
          const int cnt = min(MTILE, nm - t0);
          __syncthreads();
          for (int k = threadIdx.x; k < cnt; k += TB) tile[k] = model32[t0 + k];
          __syncthreads();
          if (!amb) continue;
          for (int u = 0; u < cnt; u += 4) {
            float sc[4];
#pragma unroll
            for (int v = 0; v < 4; v++) {
              const float4 m4 = tile[min(u + v, cnt - 1)];
              sc[v] = __builtin_fmaf(-q0, m4.x, __builtin_fmaf(-q1, m4.y, __builtin_fmaf(-q2, m4.z, m4.w)));
            }
#pragma unroll
            for (int v = 0; v < 4; v++) {
              if (u + v >= cnt || !(sc[v] <= lim || !(sc[v] == sc[v]) || all)) continue;
              const int jj = t0 + u + v;
              const double e0 = p[0] - model[3 * jj], e1 = p[1] - model[3 * jj + 1], e2 = p[2] - model[3 * jj + 2];
              const double dd = e0 * e0 + e1 * e1 + e2 * e2;
              if (!have || dd < best) {
                best = dd;
                order = jj;
                have = true;
              }
            }
          }
        }
      }
    } else if (amb) {
      // more than one candidate within the bound (or non-finite values): exact binary64 among the candidates, in
      // index order, strict `<` -- FindClosestPointSet's rule (the C# seeds with model[0] and replaces on `<`, so
      // the lowest index among the exact minima wins; every exact minimum is a candidate by the bound)
      const float lim = b1 + tol2;
      double best = INFINITY;
      bool have = false;
      order = 0;
      for (int jj = 0; jj < nm; jj++) {
        const float4 m4 = model32[jj];
        const float sc = __builtin_fmaf(-q0, m4.x, __builtin_fmaf(-q1, m4.y, __builtin_fmaf(-q2, m4.z, m4.w)));
        if (sc <= lim || !(sc == sc) || !(lim == lim)) {
          const double e0 = p[0] - model[3 * jj], e1 = p[1] - model[3 * jj + 1], e2 = p[2] - model[3 * jj + 2];
          const double dd = e0 * e0 + e1 * e1 + e2 * e2;
          if (!have || dd < best) {
            best = dd;
            order = jj;
            have = true;
          }
        }
      }
    }
    }  // !GRID
    if (!live || sub != 0) continue;  // one lane of the group carries the point into the sums
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
  __shared__ double sm[TB / 64][16];
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
    for (int k = 1; k < TB / 64; k++) v += sm[k][threadIdx.x];
    partial[(size_t)blockIdx.x * 16 + threadIdx.x] = v;
  }
}

// ---- small models (nm <= 512: MainForm's 100 truths, C3) ---------------------------------------------------
// The same pass with the screening loop reshaped for the VALU, which bounds it at 1 M x 100:
//   * every lane works TWO data points at once, so the three FMAs of a score are packed binary32 operations
//     (v_pk_fma_f32: two points per instruction against the one wave-uniform model point from the scalar cache);
//   * the model index rides in the low `ib` mantissa bits of the score, so "smallest and second smallest score and
//     the index of the smallest" is one v_and_or, one v_med3 and one v_min per pair instead of a compare, a min and
//     three selects.  Replacing the low bits moves a score by at most 2^(ib-24) of its magnitude (<= 4.5 S^2); the
//     ambiguity bound grows by twice that: (54 + 9 * 2^ib) u S^2, rounded up to a power of two on the host (tolk).
//     A point whose two best scores are closer than the bound is decided in binary64 exactly as before.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// min of two scores as ONE instruction: fminf() first quiets both operands (a v_max x, x each) for IEEE signalling-NaN
// semantics, doubling the cost of the hottest statement; NaN scores never reach the result (see tol2 below)
__device__ __forceinline__ float min_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int TB>
__global__ __launch_bounds__(TB) void k_icp_pass_small(const double* __restrict__ model, const float4* __restrict__ model32,
                                                      int nm, const double* __restrict__ data, int64_t nd,
                                                      const IcpState* __restrict__ st, double* __restrict__ partial,
                                                      int32_t* __restrict__ nn, uint32_t imask, double tolk) {
  if (st->done) return;
  double R[9], T[3];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = st->R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) T[k] = st->T[k];
  const double cen0 = st->cen[0], cen1 = st->cen[1], cen2 = st->cen[2], mmax = st->mmax;
  double s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = 0.0;
  const int64_t half = (nd + 1) >> 1;  // lane v works points v and v + half
  for (int64_t base = (int64_t)blockIdx.x * TB; base < half; base += (int64_t)gridDim.x * TB) {  // uniform trip count
    const int64_t v = base + threadIdx.x;
    int64_t idx[2] = {v, v + half};
    bool live[2] = {v < half, v < half && v + half < nd};
    double p[2][3];
    float q[2][3], tol2[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int64_t il = live[h] ? idx[h] : nd - 1;  // idle slots recompute the last point, unused
      const double d0 = data[3 * il], d1 = data[3 * il + 1], d2 = data[3 * il + 2];
      // TransPoint: r = R*p accumulated k ascending from 0 (Matrix.StupidMultiply), then + T
#pragma unroll
      for (int r = 0; r < 3; r++) {
        double acc = 0.0;
        acc += R[3 * r] * d0;
        acc += R[3 * r + 1] * d1;
        acc += R[3 * r + 2] * d2;
        p[h][r] = acc + T[r];
      }
      const double pc0 = p[h][0] - cen0, pc1 = p[h][1] - cen1, pc2 = p[h][2] - cen2;
      q[h][0] = (float)pc0;
      q[h][1] = (float)pc1;
      q[h][2] = (float)pc2;
      const double S = fmax(mmax, fmax(fabs(pc0), fmax(fabs(pc1), fabs(pc2))));
      const float t2 = (float)(S * S * tolk) * 1.0001f;
      // scores are bounded by 4.5 S^2: beyond binary32 range (or NaN) the point goes to the exact scan
      // outside binary32's normal range the rounding model behind tolk does not hold (underflow: absolute errors of
      // 2^-149; overflow: inf scores): NaN = every candidate goes through the binary64 comparison
      tol2[h] = ((float)(S * S) < 1.0e37f && t2 >= 1.0e-30f) ? t2 : NAN;
    }
    const f32x2 nq0 = {-q[0][0], -q[1][0]}, nq1 = {-q[0][1], -q[1][1]}, nq2 = {-q[0][2], -q[1][2]};
    float b1[2] = {INFINITY, INFINITY}, b2[2] = {INFINITY, INFINITY};
    // the mask lives in a VGPR so that (score & keep) | j is ONE v_and_or_b32 (a VOP3 reads one scalar operand: j)
    uint32_t keep;
    asm volatile("v_mov_b32 %0, %1" : "=v"(keep) : "s"(~imask));
    int j = 0;
    for (; j + 1 < nm; j += 2) {
      const float4 ma = model32[j], mb = model32[j + 1];
      const f32x2 sa = __builtin_elementwise_fma(nq0, (f32x2)(ma.x), __builtin_elementwise_fma(nq1, (f32x2)(ma.y),
                       __builtin_elementwise_fma(nq2, (f32x2)(ma.z), (f32x2)(ma.w))));
      const f32x2 sb = __builtin_elementwise_fma(nq0, (f32x2)(mb.x), __builtin_elementwise_fma(nq1, (f32x2)(mb.y),
                       __builtin_elementwise_fma(nq2, (f32x2)(mb.z), (f32x2)(mb.w))));
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const float ta = __uint_as_float((__float_as_uint(sa[h]) & keep) | (uint32_t)j);
        const float tb = __uint_as_float((__float_as_uint(sb[h]) & keep) | (uint32_t)(j + 1));
        b2[h] = __builtin_amdgcn_fmed3f(b1[h], b2[h], ta);
        b1[h] = min_raw(b1[h], ta);
        b2[h] = __builtin_amdgcn_fmed3f(b1[h], b2[h], tb);
        b1[h] = min_raw(b1[h], tb);
      }
    }
    if (j < nm) {
      const float4 ma = model32[j];
      const f32x2 sa = __builtin_elementwise_fma(nq0, (f32x2)(ma.x), __builtin_elementwise_fma(nq1, (f32x2)(ma.y),
                       __builtin_elementwise_fma(nq2, (f32x2)(ma.z), (f32x2)(ma.w))));
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const float ta = __uint_as_float((__float_as_uint(sa[h]) & keep) | (uint32_t)j);
        b2[h] = __builtin_amdgcn_fmed3f(b1[h], b2[h], ta);
        b1[h] = min_raw(b1[h], ta);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      int order = (int)(__float_as_uint(b1[h]) & imask);
      const bool amb = !(b2[h] > b1[h] + tol2[h]);
      if (amb) {
        // more than one candidate within the bound (or non-finite values): exact binary64 among the candidates, in
        // index order, strict `<` -- FindClosestPointSet's rule (lowest index among the exact minima)
        const float lim = b1[h] + tol2[h];
        double best = INFINITY;
        bool have = false;
        order = 0;
        for (int jj = 0; jj < nm; jj++) {
          const float4 m4 = model32[jj];
          const float sc = __builtin_fmaf(-q[h][0], m4.x, __builtin_fmaf(-q[h][1], m4.y, __builtin_fmaf(-q[h][2], m4.z, m4.w)));
          if (sc <= lim || !(sc == sc) || !(lim == lim)) {
            const double e0 = p[h][0] - model[3 * jj], e1 = p[h][1] - model[3 * jj + 1], e2 = p[h][2] - model[3 * jj + 2];
            const double dd = e0 * e0 + e1 * e1 + e2 * e2;
            if (!have || dd < best) {
              best = dd;
              order = jj;
              have = true;
            }
          }
        }
      }
      if (!live[h]) continue;
      if (nn) nn[idx[h]] = order;
      const double y0 = model[3 * order], y1 = model[3 * order + 1], y2 = model[3 * order + 2];
      const double y[3] = {y0, y1, y2};
#pragma unroll
      for (int r = 0; r < 3; r++) {
        s[r] += p[h][r];
        s[3 + r] += y[r];
#pragma unroll
        for (int c = 0; c < 3; c++) s[6 + 3 * r + c] += p[h][r] * y[c];
      }
      const double e0 = p[h][0] - y0, e1 = p[h][1] - y1, e2 = p[h][2] - y2;
      s[15] += e0 * e0 + e1 * e1 + e2 * e2;
    }
  }
  __shared__ double sm[TB / 64][16];
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
    for (int k = 1; k < TB / 64; k++) v += sm[k][threadIdx.x];
    partial[(size_t)blockIdx.x * 16 + threadIdx.x] = v;
  }
}

// ---- Horn's unit-quaternion closed form (host and device: same code, same rounding) -----------------
// cyclic Jacobi sweeps on a symmetric 4x4 (independent of the oracle's max-pivot variant).  Every index is a
// compile-time constant after unrolling, so on the device A and V live in registers (dynamic indexing would put
// them in scratch memory and make the single solving thread several times slower).
template <int P, int Q>
__host__ __device__ inline void jrot(double (&A)[4][4], double (&V)[4][4]) {
  if (A[P][Q] == 0.0) return;
  const double theta = (A[Q][Q] - A[P][P]) / (2.0 * A[P][Q]);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double akp = A[k][P], akq = A[k][Q];
    A[k][P] = c * akp - s * akq;
    A[k][Q] = s * akp + c * akq;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double apk = A[P][k], aqk = A[Q][k];
    A[P][k] = c * apk - s * aqk;
    A[Q][k] = s * apk + c * aqk;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double vkp = V[k][P], vkq = V[k][Q];
    V[k][P] = c * vkp - s * vkq;
    V[k][Q] = s * vkp + c * vkq;
  }
}

// A = V0^T Q V0 and V = V0 on entry (V0 = identity for a cold start): on return the columns of V are eigenvectors of Q
__host__ __device__ inline void jacobi4(double (&A)[4][4], double (&V)[4][4]) {
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, diag = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      diag += A[i][i] * A[i][i];
#pragma unroll
      for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
    }
    if (off <= 1e-34 * (diag + off) || off == 0.0) break;
    // three rounds of two rotations in DISJOINT planes: the angle of the second reads nothing the first one writes, so
    // the two chains of divisions and square roots overlap in the single lane that runs this
    jrot<0, 1>(A, V);
    jrot<2, 3>(A, V);
    jrot<0, 2>(A, V);
    jrot<1, 3>(A, V);
    jrot<0, 3>(A, V);
    jrot<1, 2>(A, V);
  }
}

// Vst [16]: the eigenvector basis of the previous round (row major), or NULL.  Consecutive rounds of an ICP solve
// nearly the same 4x4 problem, so the sweeps start from the previous basis (A = V^T Q V is then almost diagonal: one
// or two sweeps instead of six or seven -- the Jacobi chain is what bounds k_icp_step); the basis found is stored back.
// A basis that is not finite or has drifted from orthonormal (first round, failed round) is replaced by the identity.
__host__ __device__ inline bool horn(const double s[16], long long nd, double R1[9], double T1[3], double* Vst) {
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
  bool warm = Vst != nullptr;
  if (warm) {
    bool ortho = true;  // | V^T V - I |_max <= 1e-9, tested per entry: fmax would drop a NaN
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        V[i][j] = Vst[4 * i + j];
      }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = i; j < 4; j++) {
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) d += V[k][i] * V[k][j];
        ortho = ortho && (fabs(d - (i == j ? 1.0 : 0.0)) <= 1e-9);  // false for NaN
      }
    warm = ortho;
  }
  if (warm) {
    double QV[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) d += Q[i][k] * V[k][j];
        QV[i][j] = d;
      }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = i; j < 4; j++) {
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) d += V[k][i] * QV[k][j];
        Q[i][j] = d;
        Q[j][i] = d;
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) V[i][j] = i == j ? 1.0 : 0.0;
  }
  jacobi4(Q, V);
  if (Vst) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) Vst[4 * i + j] = V[i][j];
  }
  // eigenvector of the largest eigenvalue, picked without dynamic indexing (first maximum wins)
  double ev = Q[0][0];
  double q[4] = {V[0][0], V[1][0], V[2][0], V[3][0]};
#pragma unroll
  for (int i = 1; i < 4; i++)
    if (Q[i][i] > ev) {
      ev = Q[i][i];
#pragma unroll
      for (int k = 0; k < 4; k++) q[k] = V[k][i];
    }
  const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (!(nrm > 0.0) || !(nrm <= 1.7976931348623157e308)) return false;
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

// What follows the pass: fixed-order reduction of the partial rows (thread t adds rows t, t+TB, ... in order, then a
// fixed shuffle/LDS tree: bitwise reproducible, no float atomics), then thread 0 advances the ICP state by one round.
// (Running this in the pass's last workgroup -- ticket + release fence per workgroup -- was built and measured at
// 1 M x 100: 74 us per round against 51 us for the two launches; the Horn solve's registers also halve the pass's
// occupancy.  It stays a launch of its own.)
template <int TB>
__device__ __forceinline__ void icp_step_body(const double* __restrict__ partial, int nb, IcpState* __restrict__ st,
                                              const StepArgs& a) {
  double s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = 0.0;
  for (int b = threadIdx.x; b < nb; b += TB) {
#pragma unroll
    for (int k = 0; k < 16; k++)
      s[k] += partial[(size_t)b * 16 + k];
  }
  __shared__ double sm2[TB / 64][16];
  __shared__ double tot[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    double v = wave_sum(s[k]);
    if (lane == 0) sm2[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double v = sm2[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < TB / 64; k++) v += sm2[k][threadIdx.x];
    tot[threadIdx.x] = v;
    st->sums[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (a.mode == MODE_SUMS_ONLY) {
    st->done = 1;
    return;
  }
  double S[16];
  for (int k = 0; k < 16; k++) S[k] = tot[k];
  double R1[9], T1[3];
  const bool ok = horn(S, a.nd, R1, T1, st->V);
  const double pre_d = st->d;
  const double d = S[15];
  st->pre_d = pre_d;
  st->d = d;
  const int round = st->round + 1;
  st->round = round;
  bool go;
  if (a.mode == MODE_VTK) go = true;  // fixed number of rounds, mean-distance check off (FrmMain.cs:855-858)
  else if (a.stop_rule == VCP_STOP_RMSE) go = sqrt(d / (double)a.nd) >= a.tol;
  else go = fabs(d - pre_d) >= a.tol;  // BaseClass/ICP.cs:149,180
  if (go) {
    if (!ok) {
      st->failed = 1;
      st->done = 1;
      return;
    }
    if (a.mode == MODE_REFERENCE && round == 1) {  // :151-162 the first result overwrites R, T
      for (int k = 0; k < 9; k++) st->R[k] = R1[k];
      for (int k = 0; k < 3; k++) st->T[k] = T1[k];
    } else {  // :163-177  R <- R1 R, T <- R1 T + T1
      double tR[9], tT[3];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          double acc = 0.0;
          for (int k = 0; k < 3; k++) acc += R1[3 * i + k] * st->R[3 * k + j];
          tR[3 * i + j] = acc;
        }
      for (int i = 0; i < 3; i++) {
        double acc = 0.0;
        for (int k = 0; k < 3; k++) acc += R1[3 * i + k] * st->T[k];
        tT[i] = acc + T1[i];
      }
      for (int k = 0; k < 9; k++) st->R[k] = tR[k];
      for (int k = 0; k < 3; k++) st->T[k] = tT[k];
    }
  }
  if (!go || round >= a.max_iter) st->done = 1;
}

__global__ __launch_bounds__(ITPB) void k_icp_step(const double* __restrict__ partial, int nb, IcpState* __restrict__ st,
                                                  StepArgs a) {
  if (st->done) return;
  icp_step_body<ITPB>(partial, nb, st, a);
}

void identity(IcpState& s) {
  std::memset(&s, 0, sizeof(s));
  s.R[0] = s.R[4] = s.R[8] = 1.0;
}

// Runs rounds on device-resident model/data until the state says done.  `init` carries the starting R,T.
int icp_run(vcp_ctx* ctx, const double* d_model, int64_t nm, const double* d_data, int64_t nd, const IcpState& init,
            double tol, int stop_rule, int max_iter, int mode, IcpState* out, int32_t* d_nn) {
  hipStream_t st = ctx->stream;
  // small data sets: one wave per workgroup so that they reach more CUs; large models: LDS tiles
  const bool small0 = nd <= (int64_t)64 * ICP_MAX_BLOCKS;
  // models beyond the scalar cache: binned once per call (they do not move), grid search per data point; a model with
  // non-finite coordinates keeps the LDS-tiled full scan
  NNGrid ng{};
  bool grid = false;
  if (nm > 512) {
    const int grc = vcp_nngrid_build(ctx, d_model, nm, &ng);
    if (grc == VCP_OK) grid = true;
    else if (grc != VCP_ERR_UNSUPPORTED) return grc;
  }
  const bool tiled = nm > 512 && !grid;
  const bool small = grid ? nd * nng::NNG <= (int64_t)64 * ICP_MAX_BLOCKS : small0;
  const int tb = small ? 64 : ITPB;
  const bool pairs = !grid && !tiled;  // scalar-cache models: two data points per lane (k_icp_pass_small)
  const int nb = (int)vcp_blocks(grid ? nd * nng::NNG : pairs ? (nd + 1) / 2 : nd, tb, ICP_MAX_BLOCKS);
  int ib = 1;
  while ((1 << ib) < (int)nm) ib++;
  const uint32_t imask = (1u << ib) - 1u;
  double tolk = 1.0;  // smallest power of two >= (54 + 9 * 2^ib) * 2^-24
  while (tolk * 0.5 >= (54.0 + 9.0 * (double)(1u << ib)) / 16777216.0) tolk *= 0.5;
  VCP_TRY(vcp_ensure(ctx, ctx->b_icp_part, (size_t)ICP_MAX_BLOCKS * 16 * sizeof(double) + sizeof(IcpState) + 256));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, (size_t)nm * sizeof(float4) + 64));
  double* part = ctx->b_icp_part.as<double>();
  IcpState* d_st = reinterpret_cast<IcpState*>(part + (size_t)ICP_MAX_BLOCKS * 16);
  const StepArgs sa{(long long)nd, tol, stop_rule, max_iter, mode};
  float4* model32 = ctx->b_aux0.as<float4>();
  IcpState* h_st = reinterpret_cast<IcpState*>(ctx->pinned);
  *h_st = init;
  VCP_HIP(ctx, hipMemcpyAsync(d_st, h_st, sizeof(IcpState), hipMemcpyHostToDevice, st));
  if (!grid) {  // the binary32 screening frame and copy serve the full scans only
    hipLaunchKernelGGL(k_model_frame, dim3(1), dim3(ITPB), 0, st, d_model, nm, d_st);
    hipLaunchKernelGGL(k_model32, dim3(vcp_blocks(nm, ITPB)), dim3(ITPB), 0, st, d_model, nm, d_st, model32);
  }
  int launched = 0;
  for (;;) {
    const int batch = mode == MODE_SUMS_ONLY ? 1 : std::min(ICP_BATCH, max_iter - launched);
    for (int b = 0; b < batch; b++) {
#define VCP_PASS(TBV, TL) \
  hipLaunchKernelGGL((k_icp_pass<TBV, TL>), dim3(nb), dim3(TBV), 0, st, d_model, model32, (int)nm, d_data, nd, d_st, part, d_nn, ng)
      if (small && grid) VCP_PASS(64, 2);
      else if (grid) VCP_PASS(ITPB, 2);
      else if (small && tiled) VCP_PASS(64, 1);
      else if (tiled) VCP_PASS(ITPB, 1);
      else if (small)
        hipLaunchKernelGGL(k_icp_pass_small<64>, dim3(nb), dim3(64), 0, st, d_model, model32, (int)nm, d_data, nd, d_st, part,
                           d_nn, imask, tolk);
      else
        hipLaunchKernelGGL(k_icp_pass_small<ITPB>, dim3(nb), dim3(ITPB), 0, st, d_model, model32, (int)nm, d_data, nd, d_st,
                           part, d_nn, imask, tolk);
#undef VCP_PASS
      hipLaunchKernelGGL(k_icp_step, dim3(1), dim3(ITPB), 0, st, part, nb, d_st, sa);
    }
    launched += batch;
    VCP_HIP(ctx, hipGetLastError());
    VCP_HIP(ctx, hipMemcpyAsync(h_st, d_st, sizeof(IcpState), hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    if (h_st->done || launched >= max_iter || mode == MODE_SUMS_ONLY) break;
  }
  *out = *h_st;
  if (out->failed) return vcp_fail(ctx, VCP_ERR_ARG, "Horn solve failed (non-finite sums)");
  return VCP_OK;
}

}  // namespace

extern "C" {

int vcp_icp_dev(vcp_ctx* ctx, const double* d_model, int64_t nm, const double* d_data, int64_t nd, double tol,
                int max_iter, int stop_rule, double R[9], double T[3], double* sse_o, double* rmse_o,
                int32_t* iters_o) {
  if (!ctx) return VCP_ERR_ARG;
  if (nm <= 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "empty model (model[0], BaseClass/ICP.cs:233)");
  if (nd < 0 || max_iter < 1 || !R || !T) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (nm >= 0x7FFFFFFFLL / 3 || nd >= ((int64_t)1 << 40)) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "too many points");
  if (stop_rule != VCP_STOP_SSE_DELTA && stop_rule != VCP_STOP_RMSE) return vcp_fail(ctx, VCP_ERR_ARG, "stop_rule");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  if (nd == 0) {  // the C# divides by zero: NaN sums, |NaN - 0| >= e is false, one round, R and T untouched
    if (sse_o) *sse_o = 0.0;
    if (rmse_o) *rmse_o = 0.0;
    if (iters_o) *iters_o = 1;
    ctx->last_timing.clear();
    return VCP_OK;
  }
  vcp_phase(ctx, "icp_rounds");
  IcpState init, fin;
  identity(init);  // round 1 matches the raw data (P = copy of data, :22); 1*x + 0*y + 0*z is exact
  VCP_TRY(icp_run(ctx, d_model, nm, d_data, nd, init, tol, stop_rule, max_iter, MODE_REFERENCE, &fin, nullptr));
  VCP_TRY(vcp_phase_finish(ctx));
  // R, T are written once some round has asked to continue (:149-162); if round 1 already stops they stay
  // whatever the caller passed in
  const bool wrote = fin.round > 1 || (stop_rule == VCP_STOP_RMSE ? std::sqrt(fin.d / (double)nd) >= tol
                                                                    : std::fabs(fin.d) >= tol);
  if (wrote) {
    std::memcpy(R, fin.R, sizeof(fin.R));
    std::memcpy(T, fin.T, sizeof(fin.T));
  }
  if (sse_o) *sse_o = fin.d;
  if (rmse_o) *rmse_o = std::sqrt(fin.d / (double)nd);
  if (iters_o) *iters_o = fin.round;
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
  if (nt >= 0x7FFFFFFFLL / 3) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "target too large");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  int64_t step = 1;
  if (ns > max_landmarks) step = ns / max_landmarks;
  const int64_t nb = ns / step;
  std::vector<double> a((size_t)3 * nb);
  for (int64_t i = 0, j = 0; i < nb; i++, j += step)
    for (int c = 0; c < 3; c++) a[3 * i + c] = source[3 * j + c];
  IcpState init, fin;
  identity(init);
  if (start_by_matching_centroids) {  // sequential binary64 means over ALL points of both sets
    double cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
    for (int64_t i = 0; i < ns; i++)
      for (int c = 0; c < 3; c++) cs[c] += source[3 * i + c];
    for (int64_t i = 0; i < nt; i++)
      for (int c = 0; c < 3; c++) ct[c] += target[3 * i + c];
    for (int c = 0; c < 3; c++) init.T[c] = ct[c] / (double)nt - cs[c] / (double)ns;
  }
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)nt * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, (size_t)nb * 24));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, target, (size_t)nt * 24, hipMemcpyHostToDevice, ctx->stream));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, a.data(), (size_t)nb * 24, hipMemcpyHostToDevice, ctx->stream));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `a` is a local buffer
  VCP_TRY(icp_run(ctx, ctx->b_in0.as<double>(), nt, ctx->b_in2.as<double>(), nb, init, 0.0, VCP_STOP_SSE_DELTA, max_iter,
                  MODE_VTK, &fin, nullptr));
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) M[4 * r + c] = fin.R[3 * r + c];
    M[4 * r + 3] = fin.T[r];
  }
  M[12] = M[13] = M[14] = 0;
  M[15] = 1;
  if (mean_dist) *mean_dist = std::sqrt(fin.d / (double)nb);
  if (iters_o) *iters_o = fin.round;
  return VCP_OK;
}

// Host-side run of the Horn step the device executes per round (same source: horn() is __host__ __device__).
int vcp_selftest_horn(const double sums[16], int64_t nd, double V[16], int use_v, double R1[9], double T1[3]) {
  if (!sums || !R1 || !T1 || nd <= 0 || (use_v && !V)) return VCP_ERR_ARG;
  return horn(sums, (long long)nd, R1, T1, use_v ? V : nullptr) ? 1 : 0;
}

int vcp_icp_sums(vcp_ctx* ctx, const double* model, int64_t nm, const double* data, int64_t nd, const double R[9],
                 const double T[3], double sums[16], int32_t* nn) {
  if (!ctx) return VCP_ERR_ARG;
  if (nm <= 0) return vcp_fail(ctx, VCP_ERR_EMPTY, "empty model");
  if (nd <= 0 || !model || !data || !sums) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (nm >= 0x7FFFFFFFLL / 3) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "model too large");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)nm * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, (size_t)nd * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, (size_t)nd * 4));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, model, (size_t)nm * 24, hipMemcpyHostToDevice, ctx->stream));
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, data, (size_t)nd * 24, hipMemcpyHostToDevice, ctx->stream));
  IcpState init, fin;
  identity(init);
  if (R) std::memcpy(init.R, R, sizeof(init.R));
  if (T) std::memcpy(init.T, T, sizeof(init.T));
  VCP_TRY(icp_run(ctx, ctx->b_in0.as<double>(), nm, ctx->b_in2.as<double>(), nd, init, 0.0, VCP_STOP_SSE_DELTA, 1,
                  MODE_SUMS_ONLY, &fin, nn ? ctx->b_out0.as<int32_t>() : nullptr));
  std::memcpy(sums, fin.sums, sizeof(fin.sums));
  if (nn) {
    VCP_HIP(ctx, hipMemcpyAsync(nn, ctx->b_out0.p, (size_t)nd * 4, hipMemcpyDeviceToHost, ctx->stream));
    VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return VCP_OK;
}

}  // extern "C"
