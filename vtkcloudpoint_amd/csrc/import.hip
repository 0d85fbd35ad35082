// import.hip -- MainForm.AddFolder's per-row work for scan points on MI355X:
// the Distance filter (FrmMain.cs:1011), the spherical -> Cartesian conversion (:1025-1062) and the exact
// duplicate removal (:1063-1068), which in the C# is an O(n^2) FindAll over everything kept so far.
//
// Here: one conversion pass, then a device hash table on the exact (tmpx, tmpy, tmpz) triple -- insert with
// CAS + atomicMin (a slot ends up holding the smallest row index of its class), second pass keeps a row iff it
// is that smallest index, i.e. the first occurrence, exactly what the sequential FindAll rule keeps.
// Equality is the C#'s `==` on doubles (-0 == +0, NaN != NaN).  sin/cos are the device libm's: X,Y,Z agree with
// the host's to a few ulp (tests: 1e-12 relative), duplicate decisions are bit-exact (identical inputs give
// identical outputs on either side).
// The duplicate test of the C# compares the STORED, direction-mapped p.X, p.Y with the unmapped tmpx, tmpy
// (:1065); they coincide for the ImportPts defaults xdir = 2, ydir = 1 (ImportPts.cs:18-19).  For other
// directions the C# test can only fire on mirrored points; that quirk is not rebuilt: VCP_ERR_UNSUPPORTED.
#include <cmath>

#include "vcp_ctx.hpp"

namespace {
constexpr int IT = 256;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;

struct ImpP {
  double x_angle, y_angle;
  int xdir, ydir;
};

__global__ __launch_bounds__(IT) void k_import_convert(const double* __restrict__ rows, int64_t n, ImpP P,
                                                      double* __restrict__ xyz, double* __restrict__ q,
                                                      uint8_t* __restrict__ state) {
  int64_t i = (int64_t)blockIdx.x * IT + threadIdx.x;
  if (i >= n) return;
  const double mx = rows[3 * i], my = rows[3 * i + 1], D = rows[3 * i + 2];
  double X = 0, Y = 0, Z = 0, tx = 0, ty = 0, tz = 0;
  uint8_t st = 0;
  if (!(D == 0 || D > 1000)) {  // FrmMain.cs:1011
    const double PI = 3.14159265358979323846;
    const double yangjiao = (-2) * (mx - P.x_angle) / 180 * PI;
    const double fangweijiao = 2 * (my - P.y_angle) / 180 * PI;
    tx = D * cos(yangjiao) * sin(fangweijiao);
    ty = D * sin(yangjiao) * cos(fangweijiao);
    tz = D * cos(yangjiao);
    const double pick[5] = {0, ty, tx, -ty, -tx};
    X = pick[P.xdir];
    Y = pick[P.ydir];
    Z = tz;
    st = 1;
  }
  xyz[3 * i] = X;
  xyz[3 * i + 1] = Y;
  xyz[3 * i + 2] = Z;
  if (q) {
    q[3 * i] = tx;
    q[3 * i + 1] = ty;
    q[3 * i + 2] = tz;
  }
  state[i] = st;
}

__device__ __forceinline__ uint64_t hash3(double a, double b, double c) {
  uint64_t h = 1469598103934665603ull;
  const double v[3] = {a == 0 ? 0.0 : a, b == 0 ? 0.0 : b, c == 0 ? 0.0 : c};  // -0 and +0 are one class
#pragma unroll
  for (int k = 0; k < 3; k++) {
    h = (h ^ (uint64_t)__double_as_longlong(v[k])) * 1099511628211ull;
    h ^= h >> 29;
  }
  return h;
}
__device__ __forceinline__ bool same3(const double* __restrict__ q, uint32_t a, const double* t) {
  return q[3 * (size_t)a] == t[0] && q[3 * (size_t)a + 1] == t[1] && q[3 * (size_t)a + 2] == t[2];
}

__global__ __launch_bounds__(IT) void k_import_insert(const double* __restrict__ q, const uint8_t* __restrict__ state,
                                                     int64_t n, uint32_t* __restrict__ table, uint32_t mask) {
  int64_t i = (int64_t)blockIdx.x * IT + threadIdx.x;
  if (i >= n || state[i] == 0) return;
  const double t[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]};
  if (!(t[0] == t[0] && t[1] == t[1] && t[2] == t[2])) return;  // NaN: equal to nothing, never a duplicate
  uint32_t sl = (uint32_t)hash3(t[0], t[1], t[2]) & mask;
  for (;;) {
    uint32_t cur = __hip_atomic_load(&table[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == EMPTY) {
      cur = atomicCAS(&table[sl], EMPTY, (uint32_t)i);
      if (cur == EMPTY) return;  // claimed the slot for this class
    }
    if (same3(q, cur, t)) {  // the slot belongs to my class (any member has the same triple)
      atomicMin(&table[sl], (uint32_t)i);
      return;
    }
    sl = (sl + 1) & mask;
  }
}

__global__ __launch_bounds__(IT) void k_import_mark(const double* __restrict__ q, uint8_t* __restrict__ state, int64_t n,
                                                   const uint32_t* __restrict__ table, uint32_t mask,
                                                   unsigned long long* __restrict__ counters) {
  int64_t i = (int64_t)blockIdx.x * IT + threadIdx.x;
  unsigned kept = 0, dup = 0;
  if (i < n && state[i] != 0) {
    const double t[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]};
    bool isdup = false;
    if (t[0] == t[0] && t[1] == t[1] && t[2] == t[2]) {
      uint32_t sl = (uint32_t)hash3(t[0], t[1], t[2]) & mask;
      for (;;) {
        const uint32_t cur = table[sl];
        if (cur == EMPTY) break;
        if (same3(q, cur, t)) {
          isdup = cur != (uint32_t)i;  // the class keeps its first (smallest-index) row
          break;
        }
        sl = (sl + 1) & mask;
      }
    }
    state[i] = isdup ? 2 : 1;
    kept = !isdup;
    dup = isdup;
  }
  __shared__ unsigned wc[2][IT / 64];
  const unsigned long long mk = __ballot(kept), md = __ballot(dup);
  if ((threadIdx.x & 63) == 0) {
    wc[0][threadIdx.x >> 6] = (unsigned)__popcll(mk);
    wc[1][threadIdx.x >> 6] = (unsigned)__popcll(md);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = 0, b = 0;
    for (int k = 0; k < IT / 64; k++) {
      a += wc[0][k];
      b += wc[1][k];
    }
    if (a) atomicAdd(&counters[2 * (blockIdx.x & 15)], (unsigned long long)a);
    if (b) atomicAdd(&counters[2 * (blockIdx.x & 15) + 1], (unsigned long long)b);
  }
}
}  // namespace

extern "C" int vcp_import_convert(vcp_ctx* ctx, const double* rows, int64_t n, double x_angle, double y_angle, int xdir,
                                  int ydir, int dedupe, double* xyz, uint8_t* state, int64_t* kept, int64_t* duplicates) {
  if (!ctx) return VCP_ERR_ARG;
  if (n < 0 || xdir < 1 || xdir > 4 || ydir < 1 || ydir > 4 || (n > 0 && (!rows || !xyz || !state)))
    return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  if (dedupe && !(xdir == 2 && ydir == 1))
    return vcp_fail(ctx, VCP_ERR_UNSUPPORTED,
                    "duplicate removal is built for the default directions xdir=2, ydir=1 (FrmMain.cs:1065 compares the "
                    "mapped X,Y with the unmapped tmpx,tmpy)");
  if (kept) *kept = 0;
  if (duplicates) *duplicates = 0;
  if (n == 0) return VCP_OK;
  if (n >= 0x7FFFFFF0LL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "n beyond 32-bit indexing");
  VCP_TRY(vcp_bind(ctx));
  hipStream_t st = ctx->stream;
  uint64_t cap = 1;
  while (cap < (uint64_t)(2 * n + 16)) cap <<= 1;
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, (size_t)n * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, (size_t)n * 24));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out1, (size_t)n));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out2, 64 * 8));
  if (dedupe) {
    VCP_TRY(vcp_ensure(ctx, ctx->b_aux4, (size_t)n * 24));
    VCP_TRY(vcp_ensure(ctx, ctx->b_aux1, (size_t)cap * 4));
  }
  VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, rows, (size_t)n * 24, hipMemcpyHostToDevice, st));
  ImpP P{x_angle, y_angle, xdir, ydir};
  double* q = dedupe ? ctx->b_aux4.as<double>() : nullptr;
  const unsigned nb = vcp_blocks(n, IT);
  hipLaunchKernelGGL(k_import_convert, dim3(nb), dim3(IT), 0, st, ctx->b_in0.as<double>(), n, P, ctx->b_out0.as<double>(),
                     q, ctx->b_out1.as<uint8_t>());
  unsigned long long* counters = ctx->b_out2.as<unsigned long long>();
  VCP_HIP(ctx, hipMemsetAsync(counters, 0, 64 * 8, st));
  if (dedupe) {
    uint32_t* table = ctx->b_aux1.as<uint32_t>();
    VCP_HIP(ctx, hipMemsetAsync(table, 0xFF, (size_t)cap * 4, st));
    hipLaunchKernelGGL(k_import_insert, dim3(nb), dim3(IT), 0, st, q, ctx->b_out1.as<uint8_t>(), n, table,
                       (uint32_t)(cap - 1));
    hipLaunchKernelGGL(k_import_mark, dim3(nb), dim3(IT), 0, st, q, ctx->b_out1.as<uint8_t>(), n, table,
                       (uint32_t)(cap - 1), counters);
  }
  VCP_HIP(ctx, hipGetLastError());
  unsigned long long* hp = reinterpret_cast<unsigned long long*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(xyz, ctx->b_out0.p, (size_t)n * 24, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(state, ctx->b_out1.p, (size_t)n, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipMemcpyAsync(hp, counters, 32 * 8, hipMemcpyDeviceToHost, st));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  int64_t k = 0, d = 0;
  if (dedupe) {
    for (int b = 0; b < 16; b++) {
      k += (int64_t)hp[2 * b];
      d += (int64_t)hp[2 * b + 1];
    }
  } else {
    for (int64_t i = 0; i < n; i++) k += state[i] != 0;
  }
  if (kept) *kept = k;
  if (duplicates) *duplicates = d;
  return VCP_OK;
}
