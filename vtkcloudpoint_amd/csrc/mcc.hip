// mcc.hip -- Tools.getCircles / Geometry.FindMinimalBoundingCircle on MI355X (the step
// right after centroid extraction, FrmMain.cs:1539-1540).
//
// Reference: BaseClass/Tools.cs:394-409 (one circle per cluster with more than 3 points) and
// BaseClass/Geometry.cs:247-319 (MakeConvexHull :122-208 = gift wrapping on the pseudo-angle of AngleValue
// :220-246; then the smallest circle through 2 or 3 hull points that encloses the hull, first found on ties;
// FindCircle :340-372 via FindIntersection :373-432).  HullCull (:83-120) culls nothing but NaN points: the
// Rectangle2D it compares against never gets Left/Right/Top/Bottom assigned (DataModel.cs:191-208).
//
// One 256-thread workgroup per cluster.  The members of a cluster are first brought together in list order
// (stable rocPRIM radix sort by label).  Every choice the C# makes sequentially ("first in the list wins") is
// an argmin over (value, list position), so the parallel reductions reproduce it exactly; the arithmetic is
// binary64 without FMA contraction, sqrt and division correctly rounded: results are bit-identical to the oracle.
#include <string.h>  // rocprim's texture_cache_iterator.hpp calls ::memset without including it

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <cstring>

#include "vcp_ctx.hpp"

namespace {
constexpr int MT = 256;
constexpr int HMAX = 2048;  // hull points kept in LDS (32 KB); a larger hull is reported as VCP_ERR_TOO_LARGE
constexpr double DMAX = 1.7976931348623157e308;

__global__ __launch_bounds__(MT) void k_mcc_keys(const int32_t* __restrict__ labels, const int64_t* __restrict__ order,
                                                int64_t m, int32_t K, uint32_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals, uint32_t* __restrict__ bad) {
  int64_t t = (int64_t)blockIdx.x * MT + threadIdx.x;
  if (t >= m) return;
  int64_t i = order ? order[t] : t;
  int32_t l = labels[i];
  if (l < 0 || l > K) {
    atomicAdd(bad, 1u);
    l = 0;
  }
  keys[t] = (uint32_t)l;
  vals[t] = (uint32_t)i;
}

// segment bounds from the sorted keys (no per-point atomics): mark[l] = (last position of label l) + 1, an
// exclusive max-scan of the marks is the first position of every label, counts are the differences
__global__ __launch_bounds__(MT) void k_mcc_mark(const uint32_t* __restrict__ skey, int64_t m, uint32_t* __restrict__ mark) {
  int64_t t = (int64_t)blockIdx.x * MT + threadIdx.x;
  if (t >= m) return;
  const uint32_t k = skey[t];
  if (t == m - 1 || skey[t + 1] != k) mark[k] = (uint32_t)t + 1u;
}
__global__ __launch_bounds__(MT) void k_mcc_counts(const uint32_t* __restrict__ segstart, int32_t K,
                                                  uint32_t* __restrict__ counts) {
  int k = blockIdx.x * MT + threadIdx.x;
  if (k <= K) counts[k] = segstart[k + 1] - segstart[k];
}

__global__ __launch_bounds__(MT) void k_mcc_gather(const double* __restrict__ xy, const uint32_t* __restrict__ idx,
                                                  int64_t m, double* __restrict__ cxy) {
  int64_t t = (int64_t)blockIdx.x * MT + threadIdx.x;
  if (t >= m) return;
  *reinterpret_cast<double2*>(cxy + 2 * t) = *reinterpret_cast<const double2*>(xy + 2 * (int64_t)idx[t]);
}

__device__ __forceinline__ double angle_value(double x1, double y1, double x2, double y2) {  // Geometry.cs:220-246
  double dx = x2 - x1, ax = fabs(dx), dy = y2 - y1, ay = fabs(dy), t;
  if (ax + ay == 0)
    t = 40.0;  // 360f / 9f
  else
    t = dy / (ax + ay);
  if (dx < 0)
    t = 2 - t;
  else if (dy < 0)
    t = 4 + t;
  return t * 90;
}

struct Key {  // (value, value2, position): lexicographic minimum = "first in the list among the smallest"
  double a, b;
  uint32_t p;
};
__device__ __forceinline__ bool key_less(const Key& x, const Key& y) {
  if (x.a < y.a) return true;
  if (x.a > y.a) return false;
  if (x.b < y.b) return true;
  if (x.b > y.b) return false;
  return x.p < y.p;
}
__device__ __forceinline__ Key key_shfl(const Key& k, int d) {
  Key r;
  r.a = __shfl_xor(k.a, d, 64);
  r.b = __shfl_xor(k.b, d, 64);
  r.p = (uint32_t)__shfl_xor((int)k.p, d, 64);
  return r;
}
// block-wide lexicographic minimum; every thread gets the result
__device__ __forceinline__ Key block_min(Key k, Key* sm /*[MT/64]*/) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    Key o = key_shfl(k, d);
    if (key_less(o, k)) k = o;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = k;
  __syncthreads();
  Key r = sm[0];
#pragma unroll
  for (int w = 1; w < MT / 64; w++)
    if (key_less(sm[w], r)) r = sm[w];
  return r;
}

struct Best {  // (radius^2, sequence number of the candidate in the C#'s loop order)
  double r2;
  unsigned long long seq;
};

__device__ __forceinline__ bool encloses(double cx, double cy, double r2, const double2* hull, int h, int s1, int s2,
                                         int s3) {  // Geometry.cs:322-337
  for (int i = 0; i < h; i++)
    if (i != s1 && i != s2 && i != s3) {
      double dx = cx - hull[i].x, dy = cy - hull[i].y;
      if (dx * dx + dy * dy > r2) return false;
    }
  return true;
}

__device__ __forceinline__ void find_circle(double2 a, double2 b, double2 c, double* cx, double* cy, double* r2) {
  // Geometry.cs:340-372 with FindIntersection :373-406
  double x1 = (b.x + a.x) / 2, y1 = (b.y + a.y) / 2, dy1 = b.x - a.x, dx1 = -(b.y - a.y);
  double x2 = (c.x + b.x) / 2, y2 = (c.y + b.y) / 2, dy2 = c.x - b.x, dx2 = -(c.y - b.y);
  double p2x = x1 + dx1, p2y = y1 + dy1, p4x = x2 + dx2, p4y = y2 + dy2;
  double dx12 = p2x - x1, dy12 = p2y - y1, dx34 = p4x - x2, dy34 = p4y - y2;
  double den = dy12 * dx34 - dx12 * dy34;
  double t1 = ((x1 - x2) * dy34 + (y2 - y1) * dx34) / den;
  *cx = x1 + dx12 * t1;
  *cy = y1 + dy12 * t1;
  double dx = *cx - a.x, dy = *cy - a.y;
  *r2 = dx * dx + dy * dy;
}

// segstart[k] = first member slot of cluster k (k = 1..K; slot range of label 0 precedes them)
__global__ __launch_bounds__(MT) void k_mcc(const double* __restrict__ cxy, const uint32_t* __restrict__ segstart,
                                           const uint32_t* __restrict__ counts, uint8_t* __restrict__ removed,
                                           double* __restrict__ centers, double* __restrict__ radius,
                                           uint8_t* __restrict__ valid, int32_t* __restrict__ hull_n) {
  const int k = blockIdx.x + 1;
  const uint32_t cnt = counts[k];
  const int tid = threadIdx.x;
  if (cnt <= 3) {  // Tools.cs:400
    if (tid == 0) {
      valid[k - 1] = 0;
      radius[k - 1] = 0;
      centers[2 * (k - 1)] = centers[2 * (k - 1) + 1] = 0;
      if (hull_n) hull_n[k - 1] = 0;
    }
    return;
  }
  const double2* pts = reinterpret_cast<const double2*>(cxy) + segstart[k];
  uint8_t* rem = removed + segstart[k];
  __shared__ double2 hull[HMAX];
  __shared__ Key smk[MT / 64];
  __shared__ unsigned s_alive[MT / 64];

  // HullCull: only NaN coordinates fail every comparison and get dropped
  unsigned alive_local = 0;
  for (uint32_t t = tid; t < cnt; t += MT) {
    double2 p = pts[t];
    bool keep = !(p.x != p.x && p.y != p.y);  // `x <= L || x >= R || y <= T || y >= B` with L=R=T=B=0: false only for NaN,NaN
    rem[t] = keep ? 0 : 1;
    alive_local += keep;
  }
  {
    unsigned v = alive_local;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor((int)v, d, 64);
    __syncthreads();
    if ((tid & 63) == 0) s_alive[tid >> 6] = v;
    __syncthreads();
  }
  unsigned alive = 0;
  for (int w = 0; w < MT / 64; w++) alive += s_alive[w];
  if (alive == 0) {  // points[0] of an empty list: the C# throws; not reachable with finite input
    if (tid == 0) valid[k - 1] = 3;
    return;
  }
  // Geometry.cs:129-150: smallest y, then smallest x, first in the list
  Key kk{DMAX, DMAX, 0xFFFFFFFFu};
  for (uint32_t t = tid; t < cnt; t += MT)
    if (!rem[t]) {
      Key c{pts[t].y, pts[t].x, t};
      if (key_less(c, kk)) kk = c;
    }
  kk = block_min(kk, smk);
  int h = 0;
  if (tid == 0) {
    hull[0] = pts[kk.p];
    rem[kk.p] = 1;
  }
  h = 1;
  alive--;
  __syncthreads();
  double sweep = 0;
  bool overflow = false;
  while (alive > 0) {
    const double X = hull[h - 1].x, Y = hull[h - 1].y;
    // smallest pseudo-angle >= sweep (strictly below 3600), first in the list; and the first live point
    Key best{3600.0, 0.0, 0xFFFFFFFFu};
    Key first{0.0, 0.0, 0xFFFFFFFFu};  // a = b = 0 so that only the position orders it
    for (uint32_t t = tid; t < cnt; t += MT)
      if (!rem[t]) {
        if (t < first.p) first.p = t;
        double ta = angle_value(X, Y, pts[t].x, pts[t].y);
        if (ta >= sweep) {
          Key c{ta, 0.0, t};
          if (key_less(c, best)) best = c;
        }
      }
    best = block_min(best, smk);
    first = block_min(first, smk);
    uint32_t bp = best.p;
    double best_angle = best.a;
    if (bp == 0xFFFFFFFFu) {  // nobody qualified: best_pt stays points[0], best_angle stays 3600 (:168-169)
      bp = first.p;
      best_angle = 3600;
    }
    const double first_angle = angle_value(X, Y, hull[0].x, hull[0].y);
    if (first_angle >= sweep && best_angle >= first_angle) break;  // :190-195
    if (h >= HMAX) {
      overflow = true;
      break;
    }
    __syncthreads();
    if (tid == 0) {
      hull[h] = pts[bp];
      rem[bp] = 1;
    }
    h++;
    alive--;
    sweep = best_angle;
    __syncthreads();
  }
  __syncthreads();
  if (overflow) {
    if (tid == 0) {
      valid[k - 1] = 2;
      if (hull_n) hull_n[k - 1] = h;
    }
    return;
  }
  // Geometry.cs:260-312: pairs, then triples; the winner is the smallest (radius^2, loop position)
  Best mine{DMAX, ~0ull};
  const unsigned long long H = (unsigned long long)h;
  for (int i = 0; i < h - 1; i++)
    for (int j = i + 1 + tid; j < h; j += MT) {
      const double tcx = (hull[i].x + hull[j].x) / 2.0, tcy = (hull[i].y + hull[j].y) / 2.0;
      const double dx = tcx - hull[i].x, dy = tcy - hull[i].y;
      const double tr2 = dx * dx + dy * dy;
      const unsigned long long seq = (unsigned long long)i * H + (unsigned long long)j;
      if ((tr2 < mine.r2 || (tr2 == mine.r2 && seq < mine.seq)) && tr2 < DMAX && encloses(tcx, tcy, tr2, hull, h, i, j, -1)) {
        mine.r2 = tr2;
        mine.seq = seq;
      }
    }
  for (int i = 0; i < h - 2; i++)
    for (int j = i + 1; j < h - 1; j++)
      for (int kq = j + 1 + tid; kq < h; kq += MT) {
        double tcx, tcy, tr2;
        find_circle(hull[i], hull[j], hull[kq], &tcx, &tcy, &tr2);
        const unsigned long long seq = H * H + ((unsigned long long)i * H + (unsigned long long)j) * H + (unsigned long long)kq;
        if ((tr2 < mine.r2 || (tr2 == mine.r2 && seq < mine.seq)) && tr2 < DMAX && encloses(tcx, tcy, tr2, hull, h, i, j, kq)) {
          mine.r2 = tr2;
          mine.seq = seq;
        }
      }
  // block reduction of (r2, seq)
  Key bk{mine.r2, 0.0, 0};
  // seq is 64-bit: reduce in two steps -- first the smallest r2, then the smallest seq among its holders
  Key r2min = block_min(Key{mine.r2, 0.0, 0u}, smk);
  __shared__ unsigned long long s_seq;
  if (tid == 0) s_seq = ~0ull;
  __syncthreads();
  if (mine.r2 == r2min.a && mine.seq != ~0ull) atomicMin(&s_seq, mine.seq);
  __syncthreads();
  (void)bk;
  if (tid == 0) {
    double cx = pts[0].x, cy = pts[0].y, rad = 0;  // best_center = points[0] of the ORIGINAL list (:254-257)
    const unsigned long long seq = s_seq;
    if (seq != ~0ull && r2min.a < DMAX) {
      if (seq < H * H) {
        const int i = (int)(seq / H), j = (int)(seq % H);
        cx = (hull[i].x + hull[j].x) / 2.0;
        cy = (hull[i].y + hull[j].y) / 2.0;
      } else {
        const unsigned long long q = seq - H * H;
        const int kq = (int)(q % H), j = (int)((q / H) % H), i = (int)(q / (H * H));
        double r2;
        find_circle(hull[i], hull[j], hull[kq], &cx, &cy, &r2);
      }
      rad = sqrt(r2min.a);
    }
    centers[2 * (k - 1)] = cx;
    centers[2 * (k - 1) + 1] = cy;
    radius[k - 1] = rad;
    valid[k - 1] = 1;
    if (hull_n) hull_n[k - 1] = h;
  }
}

int bits_for_u32(uint64_t maxval) {
  int b = 1;
  while (b < 32 && (maxval >> b)) b++;
  return b;
}
}  // namespace

extern "C" int vcp_mcc(vcp_ctx* ctx, const double* xy, const int32_t* labels, const int64_t* order, int64_t m, int64_t n,
                       int32_t K, double* centers, double* radius, uint8_t* valid, int32_t* hull_n) {
  if (!ctx) return VCP_ERR_ARG;
  if (m < 0 || n < 0 || K < 0 || (m > 0 && (!xy || !labels))) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (K == 0) return VCP_OK;
  if (!centers || !radius || !valid) return vcp_fail(ctx, VCP_ERR_ARG, "null output");
  if (n >= 0x7FFFFFF0LL || m >= 0x7FFFFFF0LL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "n beyond 32-bit indexing");
  VCP_TRY(vcp_bind(ctx));
  hipStream_t st = ctx->stream;
  const size_t nn = (size_t)(n > 0 ? n : 1), mm = (size_t)(m > 0 ? m : 1), kk = (size_t)K;
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, nn * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in3, nn * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in2, mm * 8));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, (kk + 4) * 4 * 2 + 64));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux1, (mm + 1) * 4 * 2));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux2, (mm + 1) * 4 * 2));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux4, mm * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux5, mm));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, kk * 16));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out1, kk));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out2, kk * 8));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out3, kk * 4));
  if (m > 0) {
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, xy, (size_t)n * 16, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in3.p, labels, (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (order) VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, order, (size_t)m * 8, hipMemcpyHostToDevice, st));
  }
  uint32_t* counts = ctx->b_aux0.as<uint32_t>();   // [K+2], label 0 included
  uint32_t* segstart = counts + (K + 4);
  uint32_t* bad = segstart + (K + 4);
  uint32_t* keys_in = ctx->b_aux1.as<uint32_t>();
  uint32_t* keys_out = keys_in + (mm + 1);
  uint32_t* vals_in = ctx->b_aux2.as<uint32_t>();
  uint32_t* vals_out = vals_in + (mm + 1);
  VCP_HIP(ctx, hipMemsetAsync(counts, 0, (kk + 4) * 4 * 2 + 64, st));
  if (m > 0) {
    hipLaunchKernelGGL(k_mcc_keys, dim3(vcp_blocks(m, MT)), dim3(MT), 0, st, ctx->b_in3.as<int32_t>(),
                       order ? ctx->b_in2.as<int64_t>() : nullptr, m, K, keys_in, vals_in, bad);
    size_t tb = 0;
    const int bits = bits_for_u32((uint64_t)K);
    VCP_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, (size_t)m, 0, bits, st));
    VCP_TRY(vcp_ensure(ctx, ctx->b_aux3, tb + 64));
    VCP_HIP(ctx, rocprim::radix_sort_pairs(ctx->b_aux3.p, tb, keys_in, keys_out, vals_in, vals_out, (size_t)m, 0, bits, st));
    hipLaunchKernelGGL(k_mcc_gather, dim3(vcp_blocks(m, MT)), dim3(MT), 0, st, ctx->b_in0.as<double>(), vals_out, m,
                       ctx->b_aux4.as<double>());
  }
  if (m > 0) hipLaunchKernelGGL(k_mcc_mark, dim3(vcp_blocks(m, MT)), dim3(MT), 0, st, keys_out, m, segstart);
  VCP_TRY(vcp_exclusive_max_scan_u32(ctx, segstart, segstart, K + 2, nullptr));  // segstart[K+1] = m
  hipLaunchKernelGGL(k_mcc_counts, dim3(vcp_blocks(K + 1, MT)), dim3(MT), 0, st, segstart, K, counts);
  hipLaunchKernelGGL(k_mcc, dim3(K), dim3(MT), 0, st, ctx->b_aux4.as<double>(), segstart, counts, ctx->b_aux5.as<uint8_t>(),
                     ctx->b_out0.as<double>(), ctx->b_out2.as<double>(), ctx->b_out1.as<uint8_t>(), ctx->b_out3.as<int32_t>());
  VCP_HIP(ctx, hipGetLastError());
  uint32_t* hp = reinterpret_cast<uint32_t*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(hp, bad, 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(centers, ctx->b_out0.p, kk * 16, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(radius, ctx->b_out2.p, kk * 8, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(valid, ctx->b_out1.p, kk, hipMemcpyDeviceToHost, st));
  if (hull_n) VCP_HIP(ctx, hipMemcpyAsync(hull_n, ctx->b_out3.p, kk * 4, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (hp[0] != 0) return vcp_fail(ctx, VCP_ERR_INDEX, "%u labels outside 0..K (clusList[clusterId-1])", hp[0]);
  for (int32_t k = 0; k < K; k++) {
    if (valid[k] == 2) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "cluster %d: convex hull beyond %d points", k + 1, HMAX);
    if (valid[k] == 3) return vcp_fail(ctx, VCP_ERR_EMPTY, "cluster %d: no finite point", k + 1);
  }
  return VCP_OK;
}
