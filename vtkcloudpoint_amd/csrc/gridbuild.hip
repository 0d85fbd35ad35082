// gridbuild.hip -- caller order <-> cell order by two-level partition (MI355X, gfx950).
//
// The round-1 build sorted (cell id, index) pairs with a library radix sort and then gathered the coordinates by
// index: every gathered point paid a 128-byte line for 16 bytes (3.7x the algorithmic traffic), and the output pass
// paid the same on the way back (9x).  Here the coordinates TRAVEL with the index:
//
//   k_part_hist     per chunk of >= 8192 points: bucket histogram in LDS (bucket = 2^csh consecutive cell ids, i.e. a band
//                   of grid rows) -> counts[bucket][chunk]
//   (scan)          exclusive scan of counts in bucket-major order = where each chunk's share of each bucket starts
//   k_part_scatter  per chunk: the scanned counts become per-bucket cursors in LDS; one returning LDS atomic per point
//                   gives the slot of its record (x, y[, z], index) in the bucket-major record arrays -- each chunk
//                   owns a contiguous run per bucket, so partial lines are completed in the writing XCD's L2
//   k_part_fine     one workgroup per bucket: count the bucket's records per cell in LDS, scan, store the bucket's
//                   slice of the cell table coalesced (written exactly once, never zero-filled), then place every
//                   record at cellstart + rank: coordinates, list position, and the optional per-point extras.  The
//                   scattered stores stay inside the bucket's own window of the cell-ordered arrays (L2-resident).
//
// and back (vcp_grid_output_partition): (list position, label word) pairs partitioned by windows of 2^OWSH list
// positions, every window then written by workgroups that share an XCD (b and b+8 share an L2).
//
// The order of points inside a cell follows the LDS atomics and may differ from run to run; every result of the engine
// is order-free by construction (numbering by smallest LIST position, border rule by max id), which
// tests/test_determinism_gpu.py re-checks on the GPU.
#include "grid_common.hpp"

using namespace vcpg;

namespace {

constexpr int PT = 512;            // threads per workgroup, histogram / scatter passes
constexpr int PCH_MIN = 8192;      // smallest chunk
constexpr int FT = 1024;           // threads per workgroup, fine pass
constexpr uint32_t MAXB = 8192;    // buckets (LDS histogram of the coarse passes: 32 KB)
// fine pass: a bucket of at most wcap records is staged in LDS in its final order (counters + staging <= 160 KB)
constexpr uint32_t wcap(int gd) { return gd == 2 ? 4096u : 2816u; }

// One record of the bucket-major intermediate: 32 bytes, so that a lane's store is one contiguous 32-byte piece of a
// line (measured on MI355X, tools/micro/part_bench.hip: 0.26 ms for 10 M records into 4096 buckets against 0.33 ms
// with the coordinates and the index in separate arrays -- a divergent store costs per instruction, not per byte).
struct __attribute__((aligned(32))) Rec {
  double x, y, z;
  uint32_t idx, pad;
};

template <int GD>
__device__ __forceinline__ void rec_store(Rec* __restrict__ r, uint32_t slot, const double* q, uint32_t idx) {
  double4 v;
  v.x = q[0];
  v.y = q[1];
  v.z = GD == 3 ? q[2] : 0.0;
  v.w = __hiloint2double(0, (int)idx);
  *reinterpret_cast<double4*>(r + slot) = v;
}
template <int GD>
__device__ __forceinline__ uint32_t rec_load(const Rec* __restrict__ r, uint32_t slot, double* q) {
  const double4 v = *reinterpret_cast<const double4*>(r + slot);
  q[0] = v.x;
  q[1] = v.y;
  if (GD == 3) q[2] = v.z;
  return (uint32_t)__double2loint(v.w);
}

struct PartGeom {
  uint32_t csh;     // log2(cells per bucket)
  uint32_t B;       // buckets
  uint32_t chunk;   // points per chunk (a multiple of PT)
  uint32_t nchunk;
};

inline int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  int v = e ? atoi(e) : dflt;
  return v < lo ? lo : v > hi ? hi : v;
}

inline PartGeom part_geom(int64_t n, uint32_t ncells) {
  static const int csh = env_int("VCP_CPB_LOG2", 14, 10, 14);
  static const int target = env_int("VCP_PART_CHUNKS", 256, 64, 8192);  // long runs per (bucket, chunk) matter more
  PartGeom p;                                                           // than workgroups per CU (part_bench)
  p.csh = (uint32_t)csh;
  p.B = (uint32_t)((((uint64_t)ncells + 1) + ((1ull << p.csh) - 1)) >> p.csh);  // the table has ncells + 1 entries
  int64_t chunk = (n + target - 1) / target;
  if (chunk < PCH_MIN) chunk = PCH_MIN;
  chunk = (chunk + PT - 1) / PT * PT;
  p.chunk = (uint32_t)chunk;
  p.nchunk = (uint32_t)((n + chunk - 1) / chunk);
  return p;
}

template <int GD, bool GROUPED>
__global__ __launch_bounds__(PT) void k_part_hist(const double* __restrict__ c, int64_t n, int stride, GridP g,
                                                 const int32_t* __restrict__ group, int glo, int ghi, uint32_t csh,
                                                 uint32_t B, uint32_t chunk, uint32_t nchunk,
                                                 uint32_t* __restrict__ counts) {
  extern __shared__ uint32_t h[];
  for (uint32_t k = threadIdx.x; k < B; k += PT) h[k] = 0;
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * chunk;
  const int64_t last = min(first + (int64_t)chunk, n);
#pragma unroll 4
  for (int64_t i = first + threadIdx.x; i < last; i += PT) {
    if (GROUPED) {
      const int gg = group[i];
      if (gg < glo || gg >= ghi) continue;
    }
    double q[3];
    int cc[3];
    load_in<GD>(c, i, stride, q);
    atomicAdd(&h[cell_of<GD>(q, g, cc) >> csh], 1u);
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < B; k += PT) counts[(size_t)k * nchunk + blockIdx.x] = h[k];
}

// The scanned counts say where this chunk's share of every bucket starts: loaded into LDS they are per-bucket
// cursors, and one returning LDS atomic per point yields the record's final slot.
template <int GD, bool GROUPED>
__global__ __launch_bounds__(PT) void k_part_scatter(const double* __restrict__ c, int64_t n, int stride, GridP g,
                                                    const int32_t* __restrict__ group, int glo, int ghi, uint32_t csh,
                                                    uint32_t B, uint32_t chunk, uint32_t nchunk,
                                                    const uint32_t* __restrict__ base, Rec* __restrict__ rec,
                                                    uint32_t* __restrict__ pos) {
  extern __shared__ uint32_t h[];
  for (uint32_t k = threadIdx.x; k < B; k += PT) h[k] = base[(size_t)k * nchunk + blockIdx.x];
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * chunk;
  const int64_t last = min(first + (int64_t)chunk, n);
#pragma unroll 4
  for (int64_t i = first + threadIdx.x; i < last; i += PT) {
    if (GROUPED) {
      const int gg = group[i];
      if (gg < glo || gg >= ghi) {
        if (pos) pos[i] = NONE;  // left out of this call
        continue;
      }
    }
    double q[3];
    int cc[3];
    load_in<GD>(c, i, stride, q);
    rec_store<GD>(rec, atomicAdd(&h[cell_of<GD>(q, g, cc) >> csh], 1u), q, (uint32_t)i);
  }
}

// one pad word per 32 counters: a thread's run of consecutive counters in the scan then walks all banks
__device__ __forceinline__ uint32_t padded(uint32_t i) { return i + (i >> 5); }

// One workgroup per bucket.  LDS: the bucket's cell counters, then (behind them) a staging area of WCAP records.
// Every bucket is assembled in LDS in its final order and stored with consecutive lanes on consecutive addresses (a
// scattered 16-byte store per record was measured 7x slower than the same bytes as full lines).  A bucket of up to
// WCAP records takes one window: the scanned counters serve as per-cell cursors.  A larger one (dense clouds, the heart
// of a cluster) keeps, from the counting pass, each record's RANK inside its cell (rk, 4 B per record), so that its
// final position start[cell] + rank is known without a cursor, and walks its records once per window of WCAP
// positions (the records of a bucket stay L2-resident).
template <int GD, bool GROUPED>
__device__ __forceinline__ void fine_flush(uint32_t s0, uint32_t m, const double* sxy, const uint32_t* sidx, const GridP& g,
                                           const uint32_t* __restrict__ ord, const uint8_t* __restrict__ in_classed,
                                           const int32_t* __restrict__ group, double* __restrict__ sorted,
                                           float* __restrict__ sorted32, uint32_t* __restrict__ sord,
                                           int32_t* __restrict__ sgroup, uint8_t* __restrict__ flags,
                                           uint32_t* __restrict__ pos) {
  if (GD == 2) {
    const double2* src = reinterpret_cast<const double2*>(sxy);
    double2* dst = reinterpret_cast<double2*>(sorted) + s0;
    float2* dst32 = reinterpret_cast<float2*>(sorted32) + s0;
    for (uint32_t k = threadIdx.x; k < m; k += FT) {
      const double2 v = src[k];
      dst[k] = v;
      dst32[k] = make_float2((float)(v.x - g.mn[0]), (float)(v.y - g.mn[1]));
    }
  } else {
    for (uint32_t k = threadIdx.x; k < 3 * m; k += FT) sorted[(size_t)3 * s0 + k] = sxy[k];
    for (uint32_t k = threadIdx.x; k < m; k += FT) store_pt32<GD>(sorted32, (int64_t)s0 + k, sxy + (size_t)3 * k, g);
  }
  for (uint32_t k = threadIdx.x; k < m; k += FT) {
    const uint32_t i = sidx[k], p = s0 + k;
    sord[p] = ord ? ord[i] : i;
    if (GROUPED) sgroup[p] = group[i];
    if (in_classed) flags[p] = in_classed[i] ? F_CLASSED : 0;
    if (pos) pos[i] = p;
  }
}

template <int GD, bool GROUPED>
__global__ __launch_bounds__(FT) void k_part_fine(const Rec* __restrict__ rec, uint32_t* __restrict__ rk,
                                                 const uint32_t* __restrict__ base, const uint32_t* __restrict__ total,
                                                 uint32_t nchunk, uint32_t B, uint32_t csh, GridP g,
                                                 const uint32_t* __restrict__ ord, const uint8_t* __restrict__ in_classed,
                                                 const int32_t* __restrict__ group, uint32_t* __restrict__ cellstart,
                                                 double* __restrict__ sorted, float* __restrict__ sorted32,
                                                 uint32_t* __restrict__ sord, int32_t* __restrict__ sgroup,
                                                 uint8_t* __restrict__ flags, uint32_t* __restrict__ pos) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  __shared__ uint32_t wsum[FT / 64];
  const uint32_t b = blockIdx.x;
  constexpr uint32_t WCAP = wcap(GD);
  const uint32_t CPB = 1u << csh;
  uint32_t* cnt = reinterpret_cast<uint32_t*>(lds);                                  // [padded(CPB)]
  double* sxy = reinterpret_cast<double*>(lds + (((size_t)padded(CPB) + 4) & ~3ull) * 4);  // [WCAP * GD]
  uint32_t* sidx = reinterpret_cast<uint32_t*>(sxy + (size_t)WCAP * GD);              // [WCAP]
  const uint32_t c0 = b << csh;
  const uint32_t s = base[(size_t)b * nchunk];
  const uint32_t e = (b + 1 < B) ? base[(size_t)(b + 1) * nchunk] : *total;
  const uint32_t m = e - s;
  const bool big = m > WCAP;
  for (uint32_t k = threadIdx.x; k < padded(CPB); k += FT) cnt[k] = 0;
  __syncthreads();
  for (uint32_t j = s + threadIdx.x; j < e; j += FT) {
    double q[3];
    int cc[3];
    rec_load<GD>(rec, j, q);
    uint32_t* slot = &cnt[padded(cell_of<GD>(q, g, cc) - c0)];
    if (big) rk[j] = atomicAdd(slot, 1u);  // rank inside the cell
    else atomicAdd(slot, 1u);
  }
  __syncthreads();
  // exclusive scan of the CPB counters, offset by the bucket's first position: thread t owns PER consecutive ones
  const uint32_t PER = CPB / FT;
  const uint32_t first = threadIdx.x * PER;
  uint32_t loc = 0;
  for (uint32_t k = 0; k < PER; k++) loc += cnt[padded(first + k)];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = loc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t pre = s + inc - loc;
  for (int k = 0; k < w; k++) pre += wsum[k];
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t v = cnt[padded(first + k)];
    cnt[padded(first + k)] = pre;
    pre += v;
  }
  __syncthreads();
  // the bucket's slice of the cell table, coalesced; entries up to and including index ncells exist
  for (uint32_t k = threadIdx.x; k < CPB; k += FT)
    if (c0 + k <= g.ncells) cellstart[c0 + k] = cnt[padded(k)];
  __syncthreads();
  if (!big) {
    // one window: the scanned counters serve as per-cell cursors
    for (uint32_t j = s + threadIdx.x; j < e; j += FT) {
      double q[3];
      int cc[3];
      const uint32_t i = rec_load<GD>(rec, j, q);
      const uint32_t p = atomicAdd(&cnt[padded(cell_of<GD>(q, g, cc) - c0)], 1u) - s;
      store_pt<GD>(sxy, p, q);
      sidx[p] = i;
    }
    __syncthreads();
    fine_flush<GD, GROUPED>(s, m, sxy, sidx, g, ord, in_classed, group, sorted, sorted32, sord, sgroup, flags, pos);
    return;
  }
  for (uint32_t w0 = 0; w0 < m; w0 += WCAP) {
    for (uint32_t j = s + threadIdx.x; j < e; j += FT) {
      double q[3];
      int cc[3];
      const uint32_t i = rec_load<GD>(rec, j, q);
      const uint32_t p = cnt[padded(cell_of<GD>(q, g, cc) - c0)] + rk[j] - s - w0;  // wraps below the window
      if (p < WCAP) {
        store_pt<GD>(sxy, p, q);
        sidx[p] = i;
      }
    }
    __syncthreads();
    fine_flush<GD, GROUPED>(s + w0, min(WCAP, m - w0), sxy, sidx, g, ord, in_classed, group, sorted, sorted32, sord, sgroup,
                            flags, pos);
    __syncthreads();
  }
}

// ---- cell order -> caller order --------------------------------------------------------------------------
// Windows of 2^OWSH list positions: 8192 x (4 + 1 + 1) bytes are assembled in LDS and stored as full lines (0.022 ms for
// 10 M points, against 0.15 ms for the same stores made straight from the lanes -- tools/micro/part_bench.hip).
constexpr int OWSH = 13;
constexpr int OT = 1024, OPT = 16;  // scatter pass: OT * OPT positions per workgroup (longer runs per window)
constexpr int OWT = 512;           // write pass: one workgroup per window

__global__ __launch_bounds__(OT) void k_out_scatter(const uint32_t* __restrict__ sord, const uint32_t* __restrict__ labk,
                                                   int64_t n, uint32_t OB, uint32_t* __restrict__ gcur,
                                                   uint2* __restrict__ rec) {
  extern __shared__ uint32_t h[];
  for (uint32_t k = threadIdx.x; k < OB; k += OT) h[k] = 0;
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * (OT * OPT);
  uint32_t o[OPT], w[OPT], r[OPT];
#pragma unroll
  for (int k = 0; k < OPT; k++) {
    const int64_t p = first + (int64_t)k * OT + threadIdx.x;
    o[k] = NONE;
    if (p < n) {
      o[k] = sord[p];
      w[k] = labk[p];
      r[k] = atomicAdd(&h[o[k] >> OWSH], 1u);
    }
  }
  __syncthreads();
  // claim this workgroup's run in every window it touches (one returning atomic per touched window)
  for (uint32_t k = threadIdx.x; k < OB; k += OT) {
    const uint32_t c = h[k];
    if (c) h[k] = atomicAdd(&gcur[k], c);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < OPT; k++) {
    if (o[k] == NONE) continue;
    const uint32_t bk = o[k] >> OWSH;
    rec[((size_t)bk << OWSH) + h[bk] + r[k]] = make_uint2(o[k], w[k]);
  }
}

__global__ __launch_bounds__(OWT) void k_out_write(const uint2* __restrict__ rec, int64_t n, bool have_in_classed,
                                                  int32_t cf_in, int32_t* __restrict__ labels,
                                                  uint8_t* __restrict__ is_core, uint8_t* __restrict__ is_classed,
                                                  unsigned long long* __restrict__ counters, uint32_t* __restrict__ gcur) {
  __shared__ int32_t sl[1 << OWSH];
  if (threadIdx.x == 0) gcur[blockIdx.x] = 0u;  // the window cursors are left at zero for the next call
  __shared__ __attribute__((aligned(16))) uint8_t sc[1 << OWSH], sk[1 << OWSH];
  const int64_t lo = (int64_t)blockIdx.x << OWSH;
  const uint32_t cnt = (uint32_t)min((int64_t)1 << OWSH, n - lo);  // every list position appears exactly once
  unsigned unclassed = 0;
  for (uint32_t j = threadIdx.x; j < cnt; j += OWT) {
    const uint2 v = rec[lo + j];
    const uint32_t o = v.x - (uint32_t)lo, k1 = v.y >> 2;
    const bool core = v.y & 1u, classed = v.y & 2u;
    const int32_t lab = k1 ? cf_in + (int32_t)k1 : 0;
    sl[o] = lab;
    sc[o] = (core && !classed) ? 1 : 0;
    sk[o] = (classed || lab != 0) ? 1 : 0;
    if (!classed) unclassed++;
  }
  __syncthreads();
  if (have_in_classed) {  // labels are in/out: a point this call does not label keeps its id
    for (uint32_t j = threadIdx.x; j < cnt; j += OWT)
      if (sl[j] != 0) labels[lo + j] = sl[j];
  } else {
    for (uint32_t j = threadIdx.x; j < cnt; j += OWT) labels[lo + j] = sl[j];
  }
  // the byte arrays: 4 entries per lane where the window is whole (lo is a multiple of 2^OWSH)
  const uint32_t c4 = cnt >> 2;
  if (is_core) {
    for (uint32_t j = threadIdx.x; j < c4; j += OWT)
      reinterpret_cast<uint32_t*>(is_core + lo)[j] = reinterpret_cast<const uint32_t*>(sc)[j];
    for (uint32_t j = (c4 << 2) + threadIdx.x; j < cnt; j += OWT) is_core[lo + j] = sc[j];
  }
  if (is_classed) {
    for (uint32_t j = threadIdx.x; j < c4; j += OWT)
      reinterpret_cast<uint32_t*>(is_classed + lo)[j] = reinterpret_cast<const uint32_t*>(sk)[j];
    for (uint32_t j = (c4 << 2) + threadIdx.x; j < cnt; j += OWT) is_classed[lo + j] = sk[j];
  }
  if (have_in_classed) {  // one atomic per workgroup, spread over 32 slots (a single hot word serialises)
    __shared__ unsigned wc[OWT / 64];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) unclassed += __shfl_down(unclassed, d, 64);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = unclassed;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int k = 0; k < OWT / 64; k++) t += wc[k];
      if (t) atomicAdd(&counters[4 + (blockIdx.x & 31)], (unsigned long long)t);
    }
  }
}

// dynamic LDS beyond 64 KB has to be allowed per kernel (once per process and size)
template <class K>
int allow_lds(vcp_ctx* ctx, K kernel, size_t bytes) {
  static size_t allowed = 64 * 1024;
  if (bytes > allowed) {
    VCP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes));
    allowed = bytes;
  }
  return VCP_OK;
}

template <int GD, bool GROUPED>
int build(vcp_ctx* ctx, const GridBuildArgs& a) {
  hipStream_t st = ctx->stream;
  const PartGeom pg = part_geom(a.n, a.g.ncells);
  const size_t nc = (size_t)pg.B * pg.nchunk;
  VCP_TRY(vcp_ensure(ctx, ctx->b_hist, (nc + 8) * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_rec, (size_t)a.n * sizeof(Rec)));
  uint32_t* counts = ctx->b_hist.as<uint32_t>();
  uint32_t* total = counts + nc;
  Rec* rec = ctx->b_rec.as<Rec>();
  const size_t lds_h = (size_t)pg.B * 4;
  const uint32_t CPB = 1u << pg.csh;
  const size_t lds_f = ((((size_t)CPB + CPB / 32) + 4) & ~3ull) * 4 + (size_t)wcap(GD) * (GD * 8 + 4);
  VCP_TRY(allow_lds(ctx, k_part_fine<GD, GROUPED>, lds_f));
  vcp_phase(ctx, "part_hist");
  hipLaunchKernelGGL((k_part_hist<GD, GROUPED>), dim3(pg.nchunk), dim3(PT), lds_h, st, a.d_coords, a.n, a.stride, a.g,
                     a.d_group, a.glo, a.ghi, pg.csh, pg.B, pg.chunk, pg.nchunk, counts);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, counts, counts, (int64_t)nc, total));
  vcp_phase(ctx, "part_scatter");
  hipLaunchKernelGGL((k_part_scatter<GD, GROUPED>), dim3(pg.nchunk), dim3(PT), lds_h, st, a.d_coords, a.n, a.stride, a.g,
                     a.d_group, a.glo, a.ghi, pg.csh, pg.B, pg.chunk, pg.nchunk, counts, rec, a.pos);
  vcp_phase(ctx, "part_fine");
  VCP_TRY(vcp_ensure(ctx, ctx->b_rank, (size_t)a.n * 4));  // ranks inside the cell, written for large buckets only
  hipLaunchKernelGGL((k_part_fine<GD, GROUPED>), dim3(pg.B), dim3(FT), lds_f, st, rec, ctx->b_rank.as<uint32_t>(), counts,
                     total, pg.nchunk, pg.B,
                     pg.csh, a.g, a.d_ord, a.d_in_classed, a.d_group, a.cellstart, a.sorted, a.sorted32, a.sord, a.sgroup,
                     a.flags, a.pos);
  VCP_HIP(ctx, hipGetLastError());
  return VCP_OK;
}

}  // namespace

bool vcp_grid_partition_fits(int64_t n, uint32_t ncells) {
  if (getenv("VCP_BUILD_SORT")) return false;  // A/B switch: keep the round-1 sort-based build
  if (n <= 0) return false;
  return part_geom(n, ncells).B <= MAXB;
}

int vcp_grid_build_partition(vcp_ctx* ctx, const GridBuildArgs& a) {
  const bool grouped = a.d_group != nullptr;
  if (a.gd == 3) {
    if (grouped) return vcp_fail(ctx, VCP_ERR_ARG, "grouped calls are 2-D");
    return build<3, false>(ctx, a);
  }
  return grouped ? build<2, true>(ctx, a) : build<2, false>(ctx, a);
}

int vcp_grid_output_partition(vcp_ctx* ctx, const GridOutputArgs& a) {
  hipStream_t st = ctx->stream;
  const uint32_t OB = (uint32_t)((a.n + (1 << OWSH) - 1) >> OWSH);
  if (OB > 16384) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "more than 2^27 points: use the gather output");
  // window cursors: zero between calls (k_out_write leaves them so), cleared here only when the array is new
  const void* before = ctx->b_outcur.p;
  const size_t before_cap = ctx->b_outcur.cap;
  VCP_TRY(vcp_ensure(ctx, ctx->b_outcur, ((size_t)OB + 8) * 4));
  if (ctx->b_outcur.p != before || ctx->b_outcur.cap != before_cap)
    VCP_HIP(ctx, hipMemsetAsync(ctx->b_outcur.p, 0, ctx->b_outcur.cap, st));
  VCP_TRY(vcp_ensure(ctx, ctx->b_rec, (size_t)OB << (OWSH + 3)));
  uint32_t* gcur = ctx->b_outcur.as<uint32_t>();
  uint2* rec = ctx->b_rec.as<uint2>();
  vcp_phase(ctx, "out_scatter");
  hipLaunchKernelGGL(k_out_scatter, dim3(vcp_blocks(a.n, OT * OPT)), dim3(OT), (size_t)OB * 4, st, a.sord, a.labk, a.n, OB,
                     gcur, rec);
  vcp_phase(ctx, "out_write");
  hipLaunchKernelGGL(k_out_write, dim3(OB), dim3(OWT), 0, st, rec, a.n, a.have_in_classed, a.cf_in, a.labels, a.is_core,
                     a.is_classed, a.counters, gcur);
  VCP_HIP(ctx, hipGetLastError());
  return VCP_OK;
}
