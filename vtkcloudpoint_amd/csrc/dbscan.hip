// dbscan.hip -- DBImproved.dbscan on MI355X (gfx950).
//
// Semantics: BaseClass/DBImproved.cs:14-114 in the order-free form of SURVEY.md 8a row A3 (see
// include/vcp.h).  The `d <= eps` predicate is decided on binary32 copies of the coordinates wherever that decision is
// provably the binary64 one (screen_bounds / within_scr) and re-evaluated in binary64 with FMA contraction OFF -- the
// C#'s (SSE2) evaluation -- everywhere else; the grid only proposes candidates and is conservative by construction
// (cell edge = eps + the rounding of the binary32 coordinates it bins on, candidates re-tested).
//
// Data layout in HBM (n points, original index i, cell-ordered position p, GD = dimension of the metric):
//   cellcnt [ncells+1] u32   first position of each cell (x-fastest linear cell id)
//   sorted32[nin] f32x2 / x4 coordinates relative to the grid origin in binary32, cell order: binning and screening
//   (binary64 coordinates)   read through ExactSrc: the caller's array by index after the partition build
//                            (gridbuild.hip), a cell-ordered copy `sorted` after the sort-based build (grids too large
//                            for the partition: cellof / skey = cell ids by index / in cell order, rocPRIM radix sort)
//   pos     [n] u32          position of original index i (NONE = excluded from this call); not built when the output
//                            goes through the partition's windows
//   sord    [nin] u32        "list position" of the point (original index, or the caller's ord)
//   sgroup  [nin] i32        group (block) of the point, grouped calls only
//   flags   [nin] u8         bit0 core, bit1 classed on entry, bit2 expanding, bit3 border candidate; high nibble:
//                            number of recorded neighbours (nbr / nboff: the lists, packed per block of 256 positions)
//   wlE, wlB                 position-ordered work lists: expanding points; non-core points with a neighbour
//   parent  [nin] u32        union-find over positions (pointers only decrease); NONE = not expanding
//   minord  [nin] u32        per root: smallest list position in the component (= the seed)
//   seedflag[n/32] u32       bitmap of seed list positions; seedpref = popcount prefix per word
//   rootk   [nin] u32        seed rank of the point's cluster (NONE = not expanding); clseed [K]: seed per rank
//   labk    [nin] u32        per point: (1 + seed rank of its final cluster, 0 = none) << 2 | core | classed<<1
// Passes: bounds -> grid build (partition, or cell keys / sort / cell starts / gather) -> core count + lists + work
// lists -> union -> flatten / number -> border -> output.
#include <string.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "dbscan_engine.hpp"
#include "grid_common.hpp"

using namespace vcpg;

namespace {

// candidates examined per loop trip in the neighbour-search kernels (independent loads in flight per lane)
#ifndef VCP_UNR2
#define VCP_UNR2 4
#endif
#ifndef VCP_UNR3
#define VCP_UNR3 2
#endif
// the same for loops whose first load per candidate is one 4-byte word (k_union after phases 1-2, k_border)
#ifndef VCP_UNRW
#define VCP_UNRW 4
#endif

}  // namespace

// what vcp_slab_comps / vcp_slab_finish need from vcp_slab_begin (everything else stays in the context's
// workspace buffers, which nothing else touches between the two calls)
struct SlabState {
  bool valid = false;
  int gd = 2, metric = 0;
  int64_t n = 0, n_comp = 0;
  GridP g;
  double thr = 0.0;
  Screen sc{-1.0f, INFINITY};
  ExactSrc xs{nullptr, nullptr, 0};  // where the binary64 coordinates are: the caller's d_coords stay valid until finish
  unsigned nb = 0;
};

namespace {

template <int METRIC>
__device__ __forceinline__ bool within(const double* a, const double* b, double thr) {
  double dx = a[0] - b[0];
  double dy = a[1] - b[1];
  if (METRIC == VCP_L1_2D) {
    return fabs(dx) + fabs(dy) <= thr;
  } else if (METRIC == VCP_L2_2D) {
    return dx * dx + dy * dy <= thr;
  } else {
    double dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz <= thr;
  }
}

// binary32 value of the distance form between two screening copies (relative coordinates): what Screen.lo / hi bound
template <int METRIC>
__device__ __forceinline__ float value32(const float* a, const float* b) {
  const float dx = a[0] - b[0], dy = a[1] - b[1];
  // (the sign bit is clear in every case, also for a NaN: the counting kernels compare bit patterns)
  if (METRIC == VCP_L1_2D) {
    return fabsf(dx) + fabsf(dy);
  } else if (METRIC == VCP_L2_2D) {
    return fabsf(dx * dx + dy * dy);
  } else {
    const float dz = a[2] - b[2];
    return fabsf(dx * dx + dy * dy + dz * dz);
  }
}

// the predicate through the screen: binary32 where it is provably the binary64 answer, else the exact expression on
// the binary64 coordinates of the two cell-ordered positions
template <int GD, int METRIC>
__device__ __forceinline__ bool within_scr(const float* qf, const float* cf, Screen sc, const ExactSrc& xs, uint32_t p,
                                           uint32_t j, double thr) {
  const float v = value32<METRIC>(qf, cf);
  if (v <= sc.lo) return true;
  if (v > sc.hi) return false;
  double q[3], r[3];
  load_exact<GD>(xs, p, q);
  load_exact<GD>(xs, j, r);
  return within<METRIC>(q, r, thr);
}

// ---- bounds ---------------------------------------------------------------------------------
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmin(v, __shfl_down(v, d, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_down(v, d, 64));
  return v;
}

// out[0..2] = min, out[3..5] = max, out[6] = number of non-finite coordinates seen
__device__ __forceinline__ void block_minmax(double* mn, double* mx, double bad, double* __restrict__ out8) {
  __shared__ double sm[TPB / 64][7];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double lo = wave_min(mn[a]), hi = wave_max(mx[a]);
    if (lane == 0) {
      sm[w][a] = lo;
      sm[w][3 + a] = hi;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) bad += __shfl_down(bad, d, 64);
  if (lane == 0) sm[w][6] = bad;
  __syncthreads();
  if (threadIdx.x < 7) {
    double v = sm[0][threadIdx.x];
    for (int k = 1; k < TPB / 64; k++)
      v = threadIdx.x < 3 ? fmin(v, sm[k][threadIdx.x]) : threadIdx.x < 6 ? fmax(v, sm[k][threadIdx.x]) : v + sm[k][6];
    out8[threadIdx.x] = v;
  }
}

// partial[b*8 + a] = min of axis a, partial[b*8 + 3 + a] = max over finite values, partial[b*8 + 6] = #non-finite
template <int GD, bool GROUPED>
__global__ __launch_bounds__(TPB) void k_bounds(const double* __restrict__ c, int64_t n, int stride,
                                               const int32_t* __restrict__ group, int glo, int ghi,
                                               double* __restrict__ partial) {
  double mn[3], mx[3], bad = 0;
  for (int a = 0; a < 3; a++) {
    mn[a] = INFINITY;
    mx[a] = -INFINITY;
  }
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    if (GROUPED) {
      int g = group[i];
      if (g < glo || g >= ghi) continue;
    }
    double q[3];
    load_in<GD>(c, i, stride, q);  // (one 16-byte load per point where the layout allows: 68 -> 40 us at 10 M points)
#pragma unroll
    for (int a = 0; a < GD; a++) {
      const double v = q[a];
      if (isfinite(v)) {
        mn[a] = fmin(mn[a], v);
        mx[a] = fmax(mx[a], v);
      } else {
        bad += 1.0;
      }
    }
  }
  block_minmax(mn, mx, bad, partial + (size_t)blockIdx.x * 8);
}

__global__ __launch_bounds__(TPB) void k_bounds_final(const double* __restrict__ partial, int nb,
                                                     double* __restrict__ out) {
  double mn[3], mx[3], bad = 0;
  for (int a = 0; a < 3; a++) {
    mn[a] = INFINITY;
    mx[a] = -INFINITY;
  }
  for (int b = threadIdx.x; b < nb; b += TPB) {
    for (int a = 0; a < 3; a++) {
      mn[a] = fmin(mn[a], partial[b * 8 + a]);
      mx[a] = fmax(mx[a], partial[b * 8 + 3 + a]);
    }
    bad += partial[b * 8 + 6];
  }
  block_minmax(mn, mx, bad, out);
}

// Trimmed moments for a robust grid range: per axis the count, sum and sum of squares (about mid[a]) of the
// finite values inside [lo[a], hi[a]].  partial[b*9 + 3*a + {0,1,2}].  Only used when the eps-grid over the full
// bounding box would not fit the cell budget (a few far outliers would otherwise coarsen the grid for everybody).
struct Range3 {
  double lo[3], hi[3], mid[3];
};
template <int GD, bool GROUPED>
__global__ __launch_bounds__(TPB) void k_moments(const double* __restrict__ c, int64_t n, int stride,
                                                const int32_t* __restrict__ group, int glo, int ghi, Range3 R,
                                                double* __restrict__ partial) {
  double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    if (GROUPED) {
      int g = group[i];
      if (g < glo || g >= ghi) continue;
    }
#pragma unroll
    for (int a = 0; a < GD; a++) {
      const double v = c[i * stride + a];
      if (v >= R.lo[a] && v <= R.hi[a]) {
        const double d = v - R.mid[a];
        m[3 * a] += 1.0;
        m[3 * a + 1] += d;
        m[3 * a + 2] += d * d;
      }
    }
  }
  __shared__ double sm[TPB / 64][9];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    double v = m[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if (lane == 0) sm[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    double v = sm[0][threadIdx.x];
    for (int k = 1; k < TPB / 64; k++) v += sm[k][threadIdx.x];
    partial[(size_t)blockIdx.x * 9 + threadIdx.x] = v;
  }
}

// Order in which the counting kernels walk the candidate rows: the row of the point's OWN cell first, then the rows that
// share a face with it, then the corner rows.  A candidate of the own row is within eps far more often than one of a
// neighbouring row (the L1 diamond covers about half of the own row's three cells and a twelfth of each neighbour row's), so
// a point in a dense region reaches min_pts -- and stops -- after a fraction of the candidates the bottom-up order tested.
template <int GD>
struct RowOrder;
template <>
struct RowOrder<2> {  // (2-D: own row first was measured neutral for the count and +5 % for the union phase, whose first tree
  static constexpr int at(int k) { return k; }  // links come from the recorded lists: the bottom-up order stays)
};
template <>
struct RowOrder<3> {  // rows are indexed (dz + 1) * 3 + (dy + 1)
  static constexpr int at(int k) {
    constexpr int o[9] = {1, 3, 4, 5, 7, 0, 2, 6, 8};
    return o[k];
  }
};

// iterate the candidate rows of a cell neighbourhood: f(s, e) for the position range of each of the 3 (9)
// x-rows, in increasing position order; f returns false to stop early.  All row bounds are fetched up front
// (6 or 18 independent cellstart loads in flight) instead of two dependent loads per row.
template <int GD, bool OWN_FIRST = false, class F>
__device__ __forceinline__ void for_rows(const int* cc, const GridP& g, const CellTab& ct, F&& f) {
  constexpr int NR = GD == 3 ? 9 : 3;
  const int x0 = max(cc[0] - 1, 0), x1 = min(cc[0] + 1, g.D[0] - 1);
  uint32_t rs[NR], re[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int y = cc[1] + (r % 3) - 1;
    const int z = GD == 3 ? cc[2] + (r / 3) - 1 : 0;
    const bool ok = y >= 0 && y < g.D[1] && (GD != 3 || (z >= 0 && z < g.D[2]));
    const uint32_t base = ok ? cell_id<GD>(g, 0, y, z) : 0u;
    rs[r] = re[r] = 0u;
    if (ok) ct_range(ct, base + x0, base + x1 + 1, rs[r], re[r]);
  }
  for (int k = 0; k < NR; k++) {
    const int r = OWN_FIRST ? RowOrder<GD>::at(k) : k;
    if (rs[r] < re[r])
      if (!f(rs[r], re[r])) return;
  }
}

// the same bounds as plain arrays, for kernels that keep per-lane state across rows (a loop body that is not
// a lambda keeps that state in registers)
template <int GD>
__device__ __forceinline__ void row_bounds(const int* cc, const GridP& g, const CellTab& ct, uint32_t* rs, uint32_t* re) {
  constexpr int NR = GD == 3 ? 9 : 3;
  const int x0 = max(cc[0] - 1, 0), x1 = min(cc[0] + 1, g.D[0] - 1);
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int y = cc[1] + (r % 3) - 1;
    const int z = GD == 3 ? cc[2] + (r / 3) - 1 : 0;
    const bool ok = y >= 0 && y < g.D[1] && (GD != 3 || (z >= 0 && z < g.D[2]));
    const uint32_t base = ok ? cell_id<GD>(g, 0, y, z) : 0u;
    rs[r] = re[r] = 0u;
    if (ok) ct_range(ct, base + x0, base + x1 + 1, rs[r], re[r]);
  }
}

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  The neighbour-search kernels
// remap the block index so that each XCD walks contiguous bands of the cell-ordered arrays: a point's
// neighbour rows are then (mostly) in its own XCD's 4 MiB L2.  The array is cut into 8*XCHUNK bands dealt
// round-robin to the XCDs, so that dense and sparse parts of the cloud are spread over all of them (with one
// band per XCD the slowest XCD set the kernel time).  Speed only, never correctness.
constexpr unsigned XCHUNK = 8;
// bands per XCD for the work-list kernels: 1.  More bands cost the union-find kernels 20-35 % (more trees
// straddle two XCDs' L2s); the border search is indifferent.
constexpr unsigned LCHUNK = 1;
__device__ __forceinline__ int64_t xcd_block(unsigned nblocks) {
  const unsigned b = blockIdx.x;
  const unsigned per = nblocks / (8u * XCHUNK);  // blocks per chunk in the remapped part
  if (b >= per * 8u * XCHUNK) return b;          // tail blocks keep their index
  const unsigned x = b & 7u, i = b >> 3;         // XCD, and this block's turn on it
  return (int64_t)((i / per) * 8u + x) * per + (i % per);
}

// Work lists.  Only ~1/4 of the points are expanding and only the non-core points that have a neighbour
// need the border search; running those kernels over all positions leaves most lanes idle.  k_core therefore
// counts, per block of TPB positions, the entries of two lists; an exclusive scan of the counts and k_wl_fill
// build both lists in position order without any append atomic.  List kernels split a list into bands per XCD
// (wl_fetch).
struct WorkList {
  uint32_t* list;        // positions, in position order
  const uint32_t* scan;  // [nblk+1] exclusive scan of the per-block entry counts (blocks of TPB positions)
  uint32_t perblk;       // workgroups per band of the list
  uint32_t nblk;
};

// position handled by this thread of a list kernel, or NONE.  The LIST is cut into 8*LCHUNK bands of equal
// entry counts, dealt round-robin to the XCDs like the position bands of xcd_block (eighths of the position
// range were up to 34 % apart in entries on the C4 cloud).  The list is in position order, so a band of the
// list is still one contiguous band of the cell-ordered arrays.  perblk = workgroups per band.
__device__ __forceinline__ uint32_t wl_fetch(const WorkList& w) {
  const uint32_t total = w.scan[w.nblk] - w.scan[0];  // the scan may start at an offset (both lists share one)
  const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
  const uint32_t c = (i / w.perblk) * 8u + x;  // band
  const uint32_t lo = (uint32_t)(((uint64_t)total * c) / (8u * LCHUNK));
  const uint32_t hi = (uint32_t)(((uint64_t)total * (c + 1u)) / (8u * LCHUNK));
  const uint32_t t = lo + (i % w.perblk) * TPB + threadIdx.x;
  if (t >= hi) return NONE;
  return w.list[t];
}

// per-block entry counts of both lists (no atomics: one slot per block of TPB positions)
__device__ __forceinline__ void wl_count(bool isE, bool isB, uint32_t blk, uint32_t* __restrict__ cntE,
                                         uint32_t* __restrict__ cntB) {
  __shared__ unsigned wc[2][TPB / 64];
  const unsigned long long mE = __ballot(isE), mB = __ballot(isB);
  if ((threadIdx.x & 63) == 0) {
    wc[0][threadIdx.x >> 6] = (unsigned)__popcll(mE);
    wc[1][threadIdx.x >> 6] = (unsigned)__popcll(mB);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = 0, b = 0;
    for (int k = 0; k < TPB / 64; k++) {
      a += wc[0][k];
      b += wc[1][k];
    }
    cntE[blk] = a;
    cntB[blk] = b;
  }
}

// fill both lists from the flags, in position order
__global__ __launch_bounds__(TPB) void k_wl_fill(const uint8_t* __restrict__ flags, CellTab ct, const uint32_t* __restrict__ scanE,
                                                const uint32_t* __restrict__ scanB, uint32_t* __restrict__ listE,
                                                uint32_t* __restrict__ listB, uint32_t* __restrict__ seedflag, uint32_t nw,
                                                unsigned long long* __restrict__ counters, uint32_t* __restrict__ gtw,
                                                uint32_t G) {
  // (grouped calls: the per-group counters of border points queried twice, read by k_border_list / k_group_stats)
  for (uint32_t g = blockIdx.x * TPB + threadIdx.x; g < G; g += gridDim.x * TPB) gtw[g] = 0u;
  // also: clears the seed bitmap (8 words per workgroup of 256 positions, the tail by workgroup 0) and the counters
  // for the phases that follow
  if (threadIdx.x < 8) {
    const uint32_t wd = blockIdx.x * 8u + threadIdx.x;
    if (wd < nw) seedflag[wd] = 0u;
  }
  if (blockIdx.x == 0) {
    for (uint32_t wd = gridDim.x * 8u + threadIdx.x; wd < nw; wd += TPB) seedflag[wd] = 0u;
    if (threadIdx.x < 68) counters[threadIdx.x] = 0ull;
  }
  const uint32_t nin = *ct.nin;
  const int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const uint8_t fl = p < nin ? flags[p] : 0;
  const bool isE = fl & F_EXPAND, isB = fl & F_BCAND;
  __shared__ unsigned wo[2][TPB / 64];
  const unsigned long long mE = __ballot(isE), mB = __ballot(isB);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    wo[0][w] = (unsigned)__popcll(mE);
    wo[1][w] = (unsigned)__popcll(mB);
  }
  __syncthreads();
  unsigned oE = scanE[blockIdx.x] - scanE[0], oB = scanB[blockIdx.x] - scanB[0];
  for (int k = 0; k < w; k++) {
    oE += wo[0][k];
    oB += wo[1][k];
  }
  const unsigned long long below = (1ull << lane) - 1ull;
  if (isE) listE[oE + (unsigned)__popcll(mE & below)] = (uint32_t)p;
  if (isB) listB[oB + (unsigned)__popcll(mB & below)] = (uint32_t)p;
}

// ---- core flags (region query with early exit at min_pts) ---------------------------------------
// Neighbour lists.  While counting, every point also RECORDS the positions of the first NB = min_pts - 1 neighbours it
// meets (itself excluded): a point that stays below min_pts has then recorded its whole neighbourhood, so the border
// rule needs no second search (k_border_list), and an expanding point can take its first tree link from the list
// (k_union_init_list) instead of searching again.  Lists are staged per lane in LDS (k-major: lane t's k-th entry at
// [k * TPB + t], conflict free), then COMPACTED per workgroup: the lists of the 256 positions of a block are packed
// back to back at the front of the block's fixed slot of NB * 256 words, so they are stored as full lines (sparse
// per-point slots were measured 0.2 ms slower: partial-line stores).  off[p] = start of p's list inside its block's
// slot (16 bits), the count sits in the upper four bits of the point's flag byte.  NB == 0 switches the recording off
// (min_pts outside 2..16).
// A core point's list is a hint for the union-find forest (first tree link, the roots' join round), never the complete
// neighbourhood a point below min_pts needs for the border rule: it may be cut short (NbrOut::core_cap).  On the sparse
// 2-D grids four entries are as good a hint as nine (union 0.30 -> 0.28 ms: shorter walks) and a third of the list volume
// is not written (core count -16 us); in 3-D and on dense grids the full lists pay (3-D union 0.59 against 0.72 ms).
constexpr int CORE_LIST = 4;

struct NbrOut {
  uint32_t* nbr;     // [nblocks * NB * TPB]
  uint16_t* off;     // [n]
  int NB;            // 0..15
  int core_cap;      // entries a core point keeps (<= NB)
};

// exclusive prefix over the wave of a value below 16, and the wave's sum: bit-sliced -- per bit one ballot and the count of
// set lanes below (v_mbcnt), instead of six dependent cross-lane shuffles (each an LDS-pipe round trip on this part)
__device__ __forceinline__ void wave_prefix_small(uint32_t v, uint32_t& pre, uint32_t& sum) {
  pre = 0;
  sum = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const unsigned long long m = __ballot((v >> b) & 1u);
    pre += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << b;
    sum += (uint32_t)__popcll(m) << b;
  }
}

__device__ __forceinline__ const uint32_t* nbr_list(const NbrOut& no, uint32_t p) {
  return no.nbr + (size_t)(p / TPB) * (size_t)(no.NB * TPB) + no.off[p];
}

// lnb: the lanes' staged lists; lout: NB * TPB words of LDS the caller no longer needs.  blk = block of positions.
__device__ __forceinline__ void nbr_flush(const NbrOut& no, const uint32_t* lnb, uint32_t* lout, int nrec, int64_t blk,
                                          int64_t p, bool live) {
  if (no.NB == 0) return;
  __shared__ uint32_t wtot[TPB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t pre, wsum;
  wave_prefix_small((uint32_t)nrec, pre, wsum);
  if (lane == 0) wtot[w] = wsum;
  __syncthreads();  // also: every lane is done with the memory behind lout
  uint32_t total = 0;
  for (int k = 0; k < TPB / 64; k++) {
    if (k < w) pre += wtot[k];
    total += wtot[k];
  }
  for (int k = 0; k < nrec; k++) lout[pre + k] = lnb[k * TPB + threadIdx.x];
  if (live) no.off[p] = (uint16_t)pre;
  __syncthreads();
  uint32_t* dst = no.nbr + (size_t)blk * (size_t)(no.NB * TPB);
  for (uint32_t k = threadIdx.x; k < total; k += TPB) dst[k] = lout[k];
}

template <int GD, int METRIC, bool GROUPED>
__global__ __launch_bounds__(TPB) void k_core(ExactSrc xs, GridP g, double thr, int min_pts,
                                             CellTab ct,
                                             const int32_t* __restrict__ sgroup, uint8_t* __restrict__ flags,
                                             uint32_t* __restrict__ parent, uint32_t* __restrict__ minord,
                                             uint32_t* __restrict__ blkE, uint32_t* __restrict__ blkB, NbrOut no,
                                             const float* __restrict__ sorted32, Screen sc, bool has_cls) {
  extern __shared__ uint32_t lnb[];  // [NB * TPB] staged lists, then [NB * TPB] for their compaction
  const uint32_t nin = *ct.nin;
  const int64_t blk = xcd_block(gridDim.x);
  int64_t p = blk * TPB + threadIdx.x;
  const bool live = p < nin;
  int cc[3] = {0, 0, 0};
  float qf[3] = {0.f, 0.f, 0.f};
  if (live) {
    load_pt32<GD>(sorted32, p, qf);
    cell_of32<GD>(qf, g, cc);
  }
  constexpr int UNR = 4;  // hit nibbles (binary32 candidates: half the registers per candidate in flight)
  const uint32_t scLO = sc.lo < 0.0f ? 0u : __float_as_uint(sc.lo) + 1u;  // integer form of the screen, see k_core_lds
  const uint32_t scS = __float_as_uint(sc.hi) + 1u - scLO;
  const int32_t myg = (GROUPED && live) ? sgroup[p] : 0;
  int cnt = 0, nrec = 0;
  const int NB = no.NB;
  if (live) for_rows<GD, true>(cc, g, ct, [&](uint32_t s, uint32_t e) {
    // batches of UNR candidates: UNR independent loads in flight per lane (the loop is latency bound), the
    // early exit is checked once per batch.  Candidates are screened on their binary32 copies (within_scr).
    for (uint32_t j = s; j < e; j += UNR) {
      float r[UNR][3];
      int32_t gj[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const uint32_t jj = min(j + u, e - 1);
        load_pt32<GD>(sorted32, jj, r[u]);
        if (GROUPED) gj[u] = sgroup[jj];
      }
      // the screen on bit patterns, as in k_core_lds: one nibble of provisional hits per trip, no compare / select pairs
      uint32_t w[UNR], nib = 0;
#pragma unroll
      for (int u = UNR - 1; u >= 0; u--) {
        w[u] = __float_as_uint(value32<METRIC>(qf, r[u])) - scLO;
        nib = __builtin_amdgcn_alignbit(nib, w[u] - scS, 31);
      }
      nib &= 0xFu >> (4u - min(e - j, 4u));  // candidates past the row
      if (GROUPED) {
#pragma unroll
        for (int u = 0; u < UNR; u++) nib &= gj[u] == myg ? 0xFu : ~(1u << u);
      }
      if (min(min(w[0], w[1]), min(w[2], w[3])) < scS) {  // rare: some candidate is undecided
        double q[3];
        load_exact<GD>(xs, (uint32_t)p, q);
#pragma unroll
        for (int u = 0; u < UNR; u++)
          if (((nib >> u) & 1u) && w[u] < scS) {
            double rr[3];
            load_exact<GD>(xs, j + u, rr);
            if (!within<METRIC>(q, rr, thr)) nib &= ~(1u << u);
          }
      }
      cnt += __popc(nib);
      for (uint32_t m = nib; m != 0u && nrec < NB; m &= m - 1u) {
        const uint32_t jj = j + (uint32_t)(__ffs((int)m) - 1);
        if (jj != (uint32_t)p) lnb[(nrec++) * TPB + threadIdx.x] = jj;
      }
      if (cnt >= min_pts) return false;
    }
    return true;
  });
  if (cnt >= min_pts) nrec = min(nrec, no.core_cap);
  uint8_t fl = live && has_cls ? flags[p] : 0;  // has_cls: the build stored F_CLASSED bits (else nothing is there yet)
  bool isE = false, isB = false;
  if (!live) {
  } else if (cnt >= min_pts) {
    fl |= F_CORE;
    if (!(fl & F_CLASSED)) {
      fl |= F_EXPAND;
      isE = true;
    } else {
      isB = true;  // classed core point: takes the largest adjacent cluster like a border point
    }
  } else {
    isB = cnt > 1;  // no early exit happened, so cnt is exact: 1 = nobody but itself within eps
  }
  if (isB) fl |= F_BCAND;
  if (live) {
    flags[p] = fl | (uint8_t)(nrec << 4);
    // union-find start: parent[p] = p for expanding points, NONE for all others, so that the component kernels
    // can tell "expanding, and in which tree" from ONE 4-byte load per candidate
    parent[p] = isE ? (uint32_t)p : NONE;
    if (isE) minord[p] = NONE;  // (read at roots only, and a root is an expanding point)
  }
  nbr_flush(no, lnb, lnb + no.NB * TPB, nrec, blk, p, live);
  wl_count(isE, isB, (uint32_t)blk, blkE, blkB);
}

// ---- LDS-staged candidate rows (ungrouped) -----------------------------------------------------------
// A workgroup covers 256 consecutive positions = a run of cells along x, so the three candidate rows of all its lanes
// are (almost always) three short contiguous position ranges.  Their binary32 screening copies are
// staged once per workgroup (coalesced loads) and every lane scans its own sub-range from LDS instead of re-fetching
// the same lines through L1 per lane.  If a staged range would exceed the tile (dense cells, a block spanning grid rows
// far apart) the workgroup uses the global-memory loop.
template <int GD>
struct CoreTile;
template <>
struct CoreTile<2> {  // 3 rows x 512 x 8 B = 12 KB (+ slack: a lane's last trip of four may read past its range)
  static constexpr int NR = 3, CAP = 512;
  union {
    float2 pt[3][512 + 4];
    uint32_t lout[15 * TPB];  // the neighbour lists are compacted here once the rows have been scanned
  };
  uint32_t lo[3], hi[3];
  __device__ __forceinline__ void put(int r, uint32_t k, const float* __restrict__ s32, uint32_t pos) {
    pt[r][k] = reinterpret_cast<const float2*>(s32)[pos];
  }
  __device__ __forceinline__ void get(int r, uint32_t k, float* c) const {
    const float2 v = pt[r][k];
    c[0] = v.x;
    c[1] = v.y;
  }
};
// (A 3-D tile was built and measured twice.  Round 2: 9 rows x 384 x 12 B with 32-bit hit masks, 1.79 ms against 1.20 ms
// for the global-memory loop on the 10 M-point L2_3D cloud.  Round 3: 9 rows x 320 as three coordinate planes (35 KB, four
// workgroups per CU, 82 VGPRs) with 64-bit hit masks, 1.32 ms against 0.90 ms.  Nine rows of tile bounds, loads and
// barriers per workgroup cost more than the L2 hits of the direct loop save; 3-D keeps k_core.)

// row bounds of this lane (rs >= re for a missing row) and the workgroup's union per row in t.lo / t.hi;
// returns true when every range fits the tile (uniform over the workgroup)
template <int GD>
__device__ __forceinline__ bool tile_bounds(CoreTile<GD>& t, bool live, const int* cc, const GridP& g,
                                            const CellTab& ct, uint32_t* rs, uint32_t* re) {
  constexpr int NR = CoreTile<GD>::NR;
  if (threadIdx.x < NR) {
    t.lo[threadIdx.x] = NONE;
    t.hi[threadIdx.x] = 0u;
  }
  if (live) {
    row_bounds<GD>(cc, g, ct, rs, re);
  } else {
#pragma unroll
    for (int r = 0; r < NR; r++) rs[r] = re[r] = 0u;
  }
  __syncthreads();
  // The lanes of a workgroup hold consecutive positions, i.e. non-decreasing cell ids, and both ends of a row range are
  // non-decreasing in the cell id (cell id = cy * D0 + cx; the row starts at (cy + dy) * D0 + max(cx - 1, 0)): the union
  // over a wave runs from the FIRST non-empty lane's start to the LAST non-empty lane's end -- two lane reads instead of
  // two six-step shuffle reductions per row.
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const unsigned long long m = __ballot(rs[r] < re[r]);
    if (m) {
      const int f = __ffsll((long long)m) - 1, l = 63 - __clzll((long long)m);
      const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)rs[r], f);
      const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)re[r], l);
      if ((threadIdx.x & 63) == 0) {
        atomicMin(&t.lo[r], a);
        atomicMax(&t.hi[r], b);
      }
    }
  }
  __syncthreads();
  bool fits = true;
#pragma unroll
  for (int r = 0; r < NR; r++) fits = fits && (t.hi[r] <= t.lo[r] || t.hi[r] - t.lo[r] <= (uint32_t)CoreTile<GD>::CAP);
  return fits;
}

// tg / sgroup: grouped calls stage the candidates' group ids beside their coordinates
template <int GD>
__device__ __forceinline__ void tile_load(CoreTile<GD>& t, const float* __restrict__ sorted32,
                                          int32_t (*tg)[CoreTile<GD>::CAP + 4] = nullptr,
                                          const int32_t* __restrict__ sgroup = nullptr) {
#pragma unroll
  for (int r = 0; r < CoreTile<GD>::NR; r++) {
    const uint32_t lo = t.lo[r], hi = t.hi[r];
    if (hi > lo)
      for (uint32_t k = threadIdx.x; k < hi - lo; k += TPB) {
        t.put(r, k, sorted32, lo + k);
        if (tg) tg[r][k] = sgroup[lo + k];
      }
  }
  __syncthreads();
}

// LDS-free list recording for k_core_lds: the lane's hits are bit masks (hm) over its first 32 candidates per row, so
// the count is a popcount, the block-wide compaction offset a scan of the counts, and the positions are read off the
// masks straight into the compacted image -- which reuses the row tile once every lane is done with it.  A lane with a
// hit beyond the masks that stays below min_pts (`rescan`) walks its rows again in global memory: rare.
template <int GD, int METRIC>
__device__ __forceinline__ void nbr_flush_masks(const NbrOut& no, uint32_t* lout, const uint32_t* hm, const uint32_t* rs,
                                                const uint32_t* re, bool rescan, int nrec, double thr, const ExactSrc& xs,
                                                int64_t blk, int64_t p, bool live,
                                                const int32_t* __restrict__ sgroup = nullptr) {
  if (no.NB == 0) return;
  constexpr int NR = CoreTile<GD>::NR;
  __shared__ uint32_t wtot[TPB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t pre, wsum;
  wave_prefix_small((uint32_t)nrec, pre, wsum);
  if (lane == 0) wtot[w] = wsum;
  __syncthreads();  // also: every lane is done with the tile behind lout
  uint32_t total = 0;
  for (int k = 0; k < TPB / 64; k++) {
    if (k < w) pre += wtot[k];
    total += wtot[k];
  }
  if (live) no.off[p] = (uint16_t)pre;
  int k = 0;
  if (rescan) {
    double q[3];
    load_exact<GD>(xs, (uint32_t)p, q);
#pragma unroll
    for (int r = 0; r < NR; r++)  // (unrolled: rs / re stay in registers -- a runtime index would put them in scratch)
      for (uint32_t j = rs[r]; j < re[r] && k < nrec; j++) {
        double rr[3];
        load_exact<GD>(xs, j, rr);
        if (j != (uint32_t)p && within<METRIC>(q, rr, thr) && (!sgroup || sgroup[j] == sgroup[p])) lout[pre + (k++)] = j;
      }
  } else {
#pragma unroll
    for (int r = 0; r < NR; r++) {
      uint32_t m = hm[r];
      while (m != 0u && k < nrec) {
        const uint32_t pos = rs[r] + (uint32_t)(__ffs((int)m) - 1);
        m &= m - 1u;
        if (pos != (uint32_t)p) lout[pre + (k++)] = pos;
      }
    }
  }
  __syncthreads();
  uint32_t* dst = no.nbr + (size_t)blk * (size_t)(no.NB * TPB);
  for (uint32_t i = threadIdx.x; i < total; i += TPB) dst[i] = lout[i];
}

template <int GD, int METRIC, bool GROUPED = false>
__global__ __launch_bounds__(TPB) void k_core_lds(ExactSrc xs, GridP g, double thr, int min_pts,
                                                 CellTab ct, const int32_t* __restrict__ sgroup, uint8_t* __restrict__ flags,
                                                 uint32_t* __restrict__ parent, uint32_t* __restrict__ minord,
                                                 uint32_t* __restrict__ blkE, uint32_t* __restrict__ blkB, NbrOut no,
                                                 const float* __restrict__ sorted32, Screen sc, bool has_cls) {
  constexpr int NR = CoreTile<GD>::NR;
  constexpr int OWN = NR / 2;  // the row of the point's own cell (dy = dz = 0)
  __shared__ CoreTile<GD> t;
  __shared__ int32_t tg[GROUPED ? NR : 1][CoreTile<GD>::CAP + 4];  // grouped calls: the candidates' groups
  const uint32_t nin = *ct.nin;
  const int64_t blk = xcd_block(gridDim.x);
  int64_t p = blk * TPB + threadIdx.x;
  const bool live = p < nin;
  int cc[3] = {0, 0, 0};
  float qf[3] = {0.f, 0.f, 0.f};
  int32_t myg = 0;
  if (live) {
    load_pt32<GD>(sorted32, p, qf);
    cell_of32<GD>(qf, g, cc);
    if (GROUPED) myg = sgroup[p];
  }
  uint32_t rs[NR], re[NR];
  const bool fits = tile_bounds<GD>(t, live, cc, g, ct, rs, re);
  constexpr int UNR = 4;  // hit nibbles
  // integer form of the screen (see the trip below): values are non-negative, so their bit patterns order like they do
  const uint32_t scLO = sc.lo < 0.0f ? 0u : __float_as_uint(sc.lo) + 1u;  // bits below this: inside for sure
  const uint32_t scS = __float_as_uint(sc.hi) + 1u - scLO;                 // bits - scLO below this: undecided
  int cnt = 0;
  // per row a bit mask of which of the lane's first 32 candidates were hits (two extra VALU operations per candidate;
  // writing positions to LDS as they were found cost seven, plus the LDS that held them: 0.2 ms on this VALU-bound
  // kernel); ovf = hits beyond the masks
  uint32_t hm[NR], ovf = 0u;
#pragma unroll
  for (int r = 0; r < NR; r++) hm[r] = 0u;
  if (fits) {
    if constexpr (GROUPED) tile_load<GD>(t, sorted32, tg, sgroup);
    else tile_load<GD>(t, sorted32);
    if (live) {
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int r = RowOrder<GD>::at(k);
        if (cnt >= min_pts || rs[r] >= re[r]) continue;
        const uint32_t lo = t.lo[r];
        const uint32_t e = re[r] - lo, a = rs[r] - lo;
        // Branch-free trips of four: `in` = provisional hits (value <= hi), `am` = those of them the binary32 value cannot
        // decide (value > lo); candidates past the lane's range are read (the tile has slack) and masked out.  Only a
        // trip with an undecided candidate -- rare -- takes the exact binary64 test.
        for (uint32_t j = a; j < e; j += UNR) {
          float c[UNR][3];
#pragma unroll
          for (int u = 0; u < UNR; u++) t.get(r, j + u, c[u]);
          // The screen on the BIT PATTERNS of the (non-negative) values: w = bits - LO is negative for a value the screen
          // accepts outright, w < S (unsigned) for one it cannot decide, and the top bit of w - S says "provisional hit";
          // v_alignbit shifts that bit into the trip's nibble -- no compare / select pairs (each costs wait states on
          // this part: the old form spent a quarter of the trip in s_nop).  A NaN (bits above +inf's) is simply outside,
          // which is what the exact test would have answered.
          uint32_t w[UNR], nib = 0;
#pragma unroll
          for (int u = UNR - 1; u >= 0; u--) {
            w[u] = __float_as_uint(value32<METRIC>(qf, c[u])) - scLO;
            if (GROUPED) w[u] = tg[r][j + u] == myg ? w[u] : scS;  // another group's point: outside, and decided
            nib = __builtin_amdgcn_alignbit(nib, w[u] - scS, 31);
          }
          nib &= 0xFu >> (4u - min(e - j, 4u));  // candidates past the lane's range
          if (min(min(w[0], w[1]), min(w[2], w[3])) < scS) {  // rare: some candidate (maybe one past the range) is undecided
            double q[3];
            load_exact<GD>(xs, (uint32_t)p, q);
#pragma unroll
            for (int u = 0; u < UNR; u++)
              if (((nib >> u) & 1u) && w[u] < scS) {
                double rr[3];
                load_exact<GD>(xs, lo + j + u, rr);
                if (!within<METRIC>(q, rr, thr)) nib &= ~(1u << u);
              }
          }
          cnt += __popc(nib);
          const uint32_t sh = j - a;
          const uint32_t first32 = (uint32_t)((int32_t)(sh - 32u) >> 31);  // all ones while the masks still have room
          hm[r] |= (nib << (sh & 31u)) & first32;
          ovf |= nib & ~first32;
          if (cnt >= min_pts) break;
        }
      }
    }
  } else if (live) {
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const int r = RowOrder<GD>::at(k);
      if (cnt >= min_pts) continue;
      for (uint32_t j = rs[r]; j < re[r]; j += UNR) {
        float rr[UNR][3];
#pragma unroll
        for (int u = 0; u < UNR; u++) load_pt32<GD>(sorted32, min(j + u, re[r] - 1), rr[u]);
        uint32_t nib = 0;
#pragma unroll
        for (int u = 0; u < UNR; u++)
          nib |= ((j + u < re[r]) && (!GROUPED || sgroup[j + u] == myg) &&
                  within_scr<GD, METRIC>(qf, rr[u], sc, xs, (uint32_t)p, j + u, thr)) ? (1u << u) : 0u;
        cnt += __popc(nib);
        const uint32_t sh = j - rs[r];
        hm[r] |= sh < 32u ? nib << sh : 0u;
        ovf |= sh < 32u ? 0u : nib;
        if (cnt >= min_pts) break;
      }
    }
  }
  // recorded neighbours: a lane below min_pts with hits beyond the masks re-scans (its count is exact: cnt - itself);
  // everybody else takes what the masks hold, itself excluded
  const bool rescan = live && no.NB > 0 && ovf != 0u && cnt < min_pts;
  int nrec = 0;
  if (live && no.NB > 0) {
    if (rescan) {
      nrec = max(cnt - 1, 0);
    } else {
#pragma unroll
      for (int r = 0; r < NR; r++) nrec += __popc(hm[r]);
      const uint32_t sb = (uint32_t)p - rs[OWN];  // the point itself sits in its own row (a NaN point has no hit at all)
      if (rs[OWN] < re[OWN] && sb < 32u && ((hm[OWN] >> sb) & 1u)) nrec--;
    }
    nrec = min(nrec, cnt >= min_pts ? no.core_cap : no.NB);
  }
  uint8_t fl = live && has_cls ? flags[p] : 0;  // has_cls: the build stored F_CLASSED bits (else nothing is there yet)
  bool isE = false, isB = false;
  if (!live) {
  } else if (cnt >= min_pts) {
    fl |= F_CORE;
    if (!(fl & F_CLASSED)) {
      fl |= F_EXPAND;
      isE = true;
    } else {
      isB = true;
    }
  } else {
    isB = cnt > 1;
  }
  if (isB) fl |= F_BCAND;
  if (live) {
    flags[p] = fl | (uint8_t)(nrec << 4);
    // union-find start: parent[p] = p for expanding points, NONE for all others, so that the component kernels
    // can tell "expanding, and in which tree" from ONE 4-byte load per candidate
    parent[p] = isE ? (uint32_t)p : NONE;
    if (isE) minord[p] = NONE;  // (read at roots only, and a root is an expanding point)
  }
  nbr_flush_masks<GD, METRIC>(no, t.lout, hm, rs, re, rescan, nrec, thr, xs, blk, p, live, GROUPED ? sgroup : nullptr);
  wl_count(isE, isB, (uint32_t)blk, blkE, blkB);
}

// ---- union-find over expanding points -------------------------------------------------------------
// Pointers always go from a larger to a smaller position, so the forest stays acyclic under any
// interleaving.  Loads may be stale (another XCD's L2): a stale value is an OLDER ancestor link, still
// valid; every structural change goes through a device-scope CAS, which is the arbiter.
__device__ __forceinline__ uint32_t ld_parent(const uint32_t* parent, uint32_t x) {
  return __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// plain (L1-cacheable) load that the optimiser may not fold: neighbouring parent[] entries share cache
// lines, and a stale value is still a valid (older) ancestor link
__device__ __forceinline__ uint32_t ld_parent_cached(const uint32_t* parent, uint32_t x) {
  return __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// find with path halving: every other node on the way is re-pointed at its grandparent (an ancestor, so the
// pointers still only decrease and any interleaving stays a forest); later finds through these nodes are shorter
__device__ __forceinline__ uint32_t uf_root(uint32_t* parent, uint32_t x) {
  uint32_t p = ld_parent_cached(parent, x);
  while (p != x) {
    const uint32_t gp = ld_parent_cached(parent, p);
    if (gp == p) return p;
    __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    x = gp;
    p = ld_parent_cached(parent, x);
  }
  return x;
}

// hook the trees of roots ra, rb (either may turn out not to be a root); returns the surviving root
__device__ __forceinline__ uint32_t uf_link(uint32_t* parent, uint32_t ra, uint32_t rb) {
  while (ra != rb) {
    if (ra < rb) {
      uint32_t t = ra;
      ra = rb;
      rb = t;
    }
    uint32_t old = atomicCAS(&parent[ra], ra, rb);  // ra > rb
    if (old == ra) return rb;
    ra = uf_root(parent, old);
    rb = uf_root(parent, rb);
  }
  return ra;
}

// Phase 1 of the component build: every expanding point links to the first expanding neighbour within eps
// it finds at a SMALLER position (any such neighbour keeps the pointers decreasing; early exit).  Plain
// stores, no atomics: each thread writes only its own parent and pointers only go down, so this is a forest.
template <int GD, int METRIC, bool GROUPED>
__global__ __launch_bounds__(TPB) void k_union_init(ExactSrc xs, GridP g, double thr,
                                                   CellTab ct,
                                                   const int32_t* __restrict__ sgroup,
                                                   const uint8_t* __restrict__ flags, uint32_t* __restrict__ parent,
                                                   WorkList wlE, const float* __restrict__ sorted32, Screen sc) {
  const uint32_t p = wl_fetch(wlE);
  if (p == NONE) return;
  float qf[3];
  int cc[3];
  load_pt32<GD>(sorted32, p, qf);
  cell_of32<GD>(qf, g, cc);
  constexpr int UNR = GD == 3 ? VCP_UNR3 : VCP_UNR2;
  const int32_t myg = GROUPED ? sgroup[p] : 0;
  const uint32_t me = (uint32_t)p;
  uint32_t first = me;
  for_rows<GD>(cc, g, ct, [&](uint32_t s, uint32_t e) {
    if (s >= me) return true;  // only smaller positions (pointers must decrease); runs come in any order
    if (e > me) e = me;
    for (uint32_t j0 = s; j0 < e; j0 += UNR) {
      float rr[UNR][3];
      bool cand[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const uint32_t jj = min(j0 + u, e - 1);
        cand[u] = (j0 + u < e) && (flags[jj] & F_EXPAND);
        if (GROUPED) cand[u] = cand[u] && sgroup[jj] == myg;
        load_pt32<GD>(sorted32, jj, rr[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; u++)
        if (first == me && cand[u] && within_scr<GD, METRIC>(qf, rr[u], sc, xs, me, j0 + u, thr)) first = j0 + u;
      if (first != me) return false;
    }
    return true;
  });
  if (first != me) parent[me] = first;
}

// Phase 1 from the recorded lists (no search): the first recorded neighbour that is expanding and sits at a smaller
// position.  An expanding point whose smaller expanding neighbours all lie beyond its NB recorded ones stays a root for
// now; phase 3 scans every edge anyway.
__global__ __launch_bounds__(TPB) void k_union_init_list(const uint8_t* __restrict__ flags, uint32_t* __restrict__ parent,
                                                        NbrOut no, WorkList wlE) {
  const uint32_t p = wl_fetch(wlE);
  if (p == NONE) return;
  const int nrec = flags[p] >> 4;
  const uint32_t* li = nbr_list(no, p);
  for (int k = 0; k < nrec; k++) {
    const uint32_t j = li[k];
    if (j < p && (flags[j] & F_EXPAND)) {
      parent[p] = j;
      return;
    }
  }
}

// Phase 2: flatten the phase-1 forest (no atomics; a racing reader sees an older or a newer ancestor).
// JOIN (first of the two flatten passes when lists were recorded): a lane that finds itself a ROOT joins its tree along
// its recorded edges -- list reads, finds and CAS only, no cell walk and no distance test (the entries are confirmed
// neighbours).  A root is the first point of its tree in position order, so its list reaches into the trees round it:
// most of the merges the full edge scan of phase 3 would otherwise discover one contended CAS at a time happen here
// (C4 cloud: the scan kernel drops from 265 to 109 us; L2_3D: union phase 1.27 -> 0.58 ms).  Letting EVERY expanding
// point join along its list was measured too: the scan falls to 87 us but that round costs 500 us in 2-D (all trees of
// a blob hooked at once).  The second pass flattens what the joins built.
// JOIN = 2 (grids with a point or more per cell, where whole regions are one component): EVERY expanding point joins
// along its list, not only the roots -- on the sparse benchmark cloud that round costs more than it saves the edge scan
// (0.5 ms against 0.02), at eps 0.7 (half of the background is core, one giant component) the edge scan falls from 8.7 ms.
template <int JOIN>
__global__ __launch_bounds__(TPB) void k_flatten0(uint32_t* __restrict__ parent, WorkList wlE,
                                                 const uint8_t* __restrict__ flags, NbrOut no) {
  const uint32_t p = wl_fetch(wlE);  // expanding points only (a quarter of the positions here)
  if (p == NONE) return;
  uint32_t r = (uint32_t)p, x = ld_parent_cached(parent, r);
  if (x == r || JOIN == 2) {
    if (!JOIN) return;
    uint32_t rp = JOIN == 2 ? uf_root(parent, p) : p;
    const int nrec = flags[p] >> 4;
    const uint32_t* li = nbr_list(no, p);
    uint32_t s0 = NONE, s1 = NONE, s2 = NONE;  // words already known to be in my tree
    for (int k = 0; k < nrec; k++) {
      const uint32_t j = li[k];
      if (!(flags[j] & F_EXPAND)) continue;
      const uint32_t y = ld_parent_cached(parent, j);
      if (y == rp || y == s0 || y == s1 || y == s2) continue;
      const uint32_t ry = uf_root(parent, y);
      const uint32_t rm = uf_root(parent, rp);
      s2 = s1;
      s1 = s0;
      s0 = rp;
      rp = (ry != rm) ? uf_link(parent, rm, ry) : rm;
      if (y != rp) s1 = y;
    }
    return;
  }
  while (x != r) {
    r = x;
    x = ld_parent_cached(parent, r);
  }
  __hip_atomic_store(&parent[p], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Dense clouds (eps well above the point spacing: hundreds of candidates per row).  After phases 1-2 nearly every
// candidate of an expanding point already hangs under the point's own root, and phase 3 would still read one parent
// word per candidate: O(n x candidates), e.g. 0.42 s for 445 k points that all lie within eps of each other.  One word
// per chunk of 64 consecutive positions says whether all its expanding points carry the SAME parent word (or whether it
// has none): phase 3 then skips such a chunk with one load when that word is its own root.  Trees only ever merge, so a
// chunk that pointed at my root when the summary was taken is still in my tree.  Used when the grid averages more than
// DENSE_PER_CELL points per cell; sparse clouds (the benchmark clouds: 0.1 per cell) keep the plain loop.
constexpr uint32_t CHUNK_MIXED = 0xFFFFFFFEu;
constexpr int DENSE_PER_CELL = 16;
__global__ __launch_bounds__(TPB) void k_chunkroot(const uint32_t* __restrict__ parent, CellTab ct,
                                                  uint32_t* __restrict__ chunkroot) {
  const uint32_t nin = *ct.nin;
  const int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const uint32_t x = p < nin ? parent[p] : NONE;  // NONE = not expanding
  const unsigned long long have = __ballot(x != NONE);
  uint32_t out = NONE;  // no expanding point in the chunk
  if (have) {
    const uint32_t r = (uint32_t)__shfl((int)x, __ffsll((long long)have) - 1, 64);
    out = __ballot(x != NONE && x != r) ? CHUNK_MIXED : r;
  }
  if ((threadIdx.x & 63) == 0) chunkroot[p >> 6] = out;
}

// Phase 3: all remaining core-core edges.  PRE (2-D, after phases 1-2): the candidate's parent word is read
// FIRST -- NONE = not expanding, my own cached root = already in my tree (the common case inside a cluster) --
// and only candidates in a different tree pay for the coordinate load and the binary64 test.  Confirmed edges
// into other trees are queued (by the other side's parent word, at most two) and joined after the scan, so that
// the find/CAS latency chains of all lanes of a wave overlap instead of each lane stalling the other 63 at a
// different trip of the scan loop.  Without phases 1-2 (3-D) every candidate is in a different tree at the
// start, so the distance test goes first and edges are joined on the spot.
#define VCP_JOIN(x)                                                \
  do {                                                             \
    const uint32_t rx__ = uf_root(parent, (x));                    \
    const uint32_t rm__ = uf_root(parent, rp);                     \
    rp = (rx__ != rm__) ? uf_link(parent, rm__, rx__) : rm__;      \
  } while (0)
#define VCP_FLUSH()                                                \
  do {                                                             \
    if (q0 != NONE) {                                              \
      const uint32_t old__ = rp;                                   \
      VCP_JOIN(q0);                                                \
      a2 = a0;                                                     \
      a0 = q0;                                                     \
      a1 = old__;                                                  \
    }                                                              \
    if (q1 != NONE) {                                              \
      VCP_JOIN(q1);                                                \
      a2 = q1;                                                     \
    }                                                              \
    q0 = q1 = NONE;                                                \
  } while (0)

// (Staging the parent words of a workgroup's three candidate rows in LDS, as the core count does with coordinates, was
// measured: 282 us against 258 us -- the per-workgroup range reduction and barriers cost more than the L1 accesses saved.)
template <int GD, int METRIC, bool GROUPED, bool PRE, bool DENSE = false>
__global__ __launch_bounds__(TPB) void k_union(ExactSrc xs, GridP g, double thr,
                                              CellTab ct,
                                              const int32_t* __restrict__ sgroup, uint32_t* __restrict__ parent,
                                              WorkList wlE, const float* __restrict__ sorted32, Screen sc,
                                              const uint32_t* __restrict__ chunkroot = nullptr) {
  const uint32_t p = wl_fetch(wlE);
  if (p == NONE) return;
  constexpr int NR = GD == 3 ? 9 : 3;
  uint32_t rs[NR], re[NR];
  float qf[3];
  int cc[3];
  load_pt32<GD>(sorted32, p, qf);
  cell_of32<GD>(qf, g, cc);
  row_bounds<GD>(cc, g, ct, rs, re);
  constexpr int UNR = PRE ? VCP_UNRW : (GD == 3 ? VCP_UNR3 : VCP_UNR2);
  const int32_t myg = GROUPED ? sgroup[p] : 0;
  const uint32_t me = (uint32_t)p;
  uint32_t rp = parent[me];  // cached root of my tree (flattened by phase 2)
  // q0,q1 = queued edges; a0..a2 = words known to be in my tree
  uint32_t q0 = NONE, q1 = NONE, a0 = NONE, a1 = NONE, a2 = NONE;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const uint32_t s = rs[r];
    const uint32_t e = min(re[r], me);  // every undirected edge is handled by its larger endpoint
    uint32_t last_c = NONE;
    for (uint32_t j0 = s; j0 < e; j0 += UNR) {
      if (DENSE) {  // entering a chunk of 64 positions that lies wholly in the row: skip it if it is all mine (or empty)
        const uint32_t cj = j0 >> 6;
        if (cj != last_c) {
          last_c = cj;
          if ((cj << 6) >= s && ((cj + 1u) << 6) <= e) {
            const uint32_t cr = chunkroot[cj];
            if (cr == NONE || cr == rp) {
              j0 = ((cj + 1u) << 6) - UNR;  // (positions of this chunk before j0 belonged to the previous trip)
              continue;
            }
          }
        }
      }
      uint32_t pj[UNR];
      float cf[UNR][3];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const uint32_t jj = min(j0 + u, e - 1);
        pj[u] = ld_parent_cached(parent, jj);
        if (!PRE) load_pt32<GD>(sorted32, jj, cf[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const uint32_t j = j0 + u, x = pj[u];
        // NONE = not expanding; my root or a word known to be in my tree = nothing to do; a word already
        // queued = that tree is being joined anyway
        if (j >= e || x == NONE || x == rp) continue;
        if (PRE && (x == a0 || x == a1 || x == a2 || x == q0 || x == q1)) continue;
        if (GROUPED && sgroup[j] != myg) continue;
        if (PRE) load_pt32<GD>(sorted32, j, cf[u]);
        if (!within_scr<GD, METRIC>(qf, cf[u], sc, xs, me, j, thr)) continue;
        if (!PRE) {
          // second hop: j's tree was hooked under my root by an earlier edge (one L2 load instead of two chases)
          const uint32_t x2 = ld_parent_cached(parent, x);
          if (x2 != rp) {
            const uint32_t rx = uf_root(parent, x2);
            const uint32_t rm = uf_root(parent, rp);
            rp = (rx != rm) ? uf_link(parent, rm, rx) : rm;
          }
          // compress j's pointer (rp is an ancestor of j now); cached store, same-XCD readers profit
          if (x != rp && j != rp) __hip_atomic_store(&parent[j], rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else if (q0 == NONE) {
          q0 = x;
        } else if (q1 == NONE) {
          q1 = x;
        } else {
          VCP_FLUSH();
          q0 = x;
        }
      }
    }
  }
  VCP_FLUSH();
  // one compression store per point (rp is an ancestor of me, so the link stays valid)
  if (rp != me) __hip_atomic_store(&parent[me], rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#undef VCP_FLUSH
#undef VCP_JOIN

// flatten + smallest list position per component; lanes of a wave that share a root (the common
// case inside a blob: the wave covers neighbouring cells) combine before one atomicMin
template <bool FILTER>
__global__ __launch_bounds__(TPB) void k_flatten(uint32_t* __restrict__ parent, const uint32_t* __restrict__ sord,
                                                uint32_t* __restrict__ minord, WorkList wlE) {
  const uint32_t p = wl_fetch(wlE);
  uint32_t r = NONE, v = NONE;
  if (p != NONE) {
    r = (uint32_t)p;
    uint32_t x = parent[r];
    while (x != r) {
      r = x;
      x = parent[r];
    }
    parent[p] = r;  // roots keep pointing at themselves, so concurrent flattening is safe
    v = sord[p];
  }
  unsigned long long todo = __ballot(r != NONE);
  const int lane = threadIdx.x & 63;
  while (todo) {
    int leader = __ffsll((long long)todo) - 1;
    uint32_t lr = __shfl(r, leader, 64);
    bool mine = (r == lr);
    uint32_t m = mine ? v : NONE;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, d, 64));
    // FILTER (grids with a point or more per cell): read first -- in a cloud that is one large component every wave would
    // otherwise queue on one word (156 k atomics on the same address took 1.5 ms at eps 0.7; a stale read only lets a
    // needless atomic through).  On sparse grids the extra dependent load costs more than it saves (+0.03 ms at 10 M).
    if (lane == leader &&
        (!FILTER || m < __hip_atomic_load(&minord[lr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
      atomicMin(&minord[lr], m);
    todo &= ~__ballot(mine);
  }
}

// Seeds are marked in a bitmap over list positions (n/32 words); rank(m) = seeds at positions < m comes from
// a scan over the per-word popcounts, 32x smaller than a scan over n flags.
__device__ __forceinline__ uint32_t seed_rank(const uint32_t* __restrict__ bits, const uint32_t* __restrict__ pref,
                                              uint32_t m) {
  const uint32_t w = m >> 5;
  return pref[w] + (uint32_t)__popc(bits[w] & ((1u << (m & 31u)) - 1u));
}
__global__ __launch_bounds__(TPB) void k_seed_popc(const uint32_t* __restrict__ bits, uint32_t nw, uint32_t* __restrict__ cnt) {
  uint32_t w = blockIdx.x * TPB + threadIdx.x;
  if (w < nw) cnt[w] = (uint32_t)__popc(bits[w]);
}

__global__ __launch_bounds__(TPB) void k_seedflag(const uint32_t* __restrict__ parent, const uint32_t* __restrict__ minord,
                                                 uint32_t* __restrict__ seedflag, WorkList wlE) {
  const uint32_t p = wl_fetch(wlE);
  if (p == NONE) return;
  if (parent[p] == p) {
    const uint32_t m = minord[p];
    atomicOr(&seedflag[m >> 5], 1u << (m & 31u));
  }
}

__global__ __launch_bounds__(TPB) void k_rootk(const uint32_t* __restrict__ parent, const uint32_t* __restrict__ minord,
                                              const uint32_t* __restrict__ seedbits, const uint32_t* __restrict__ seedpref,
                                              uint32_t* __restrict__ rootk, uint32_t* __restrict__ clseed, WorkList wlE) {
  const uint32_t p = wl_fetch(wlE);
  if (p == NONE) return;
  if (parent[p] == p) {
    uint32_t k = seed_rank(seedbits, seedpref, minord[p]);
    rootk[p] = k;
    clseed[k] = minord[p];
  }
}

// border points the C# queried twice: one atomic per WORKGROUP, spread over 32 slots (counters[36..68)).  One atomic
// per wave on a single word cost 0.2 ms here: about half of 53 k waves hit it, and same-address atomics serialise at
// ~11 ns each.
__device__ __forceinline__ void twice_add(unsigned twice, unsigned long long* __restrict__ counters) {
  __shared__ unsigned tw[TPB / 64];
  const unsigned long long m2 = __ballot(twice);
  if ((threadIdx.x & 63) == 0) tw[threadIdx.x >> 6] = (unsigned)__popcll(m2);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int k = 0; k < TPB / 64; k++) t += tw[k];
    if (t) atomicAdd(&counters[36 + (blockIdx.x & 31)], (unsigned long long)t);
  }
}

// ---- border rule: labk[p] = 1 + seed rank of the final cluster ------------------------------------
// twice[...] counts border points the C# main loop had already queried before their first cluster's
// seed came up (BaseClass/DBImproved.cs:93-104 then :63-67)
template <int GD, int METRIC, bool GROUPED>
__global__ __launch_bounds__(TPB) void k_border(ExactSrc xs, GridP g, double thr,
                                               CellTab ct,
                                               const int32_t* __restrict__ sgroup, const uint8_t* __restrict__ flags,
                                               const uint32_t* __restrict__ parent, const uint32_t* __restrict__ sord,
                                               const uint32_t* __restrict__ rootk, const uint32_t* __restrict__ clseed,
                                               uint32_t* __restrict__ labk, unsigned long long* __restrict__ counters,
                                               uint32_t* __restrict__ group_twice, WorkList wlB, uint32_t own_lo,
                                               uint32_t own_span, const float* __restrict__ sorted32, Screen sc) {
  const uint32_t p = wl_fetch(wlB);
  unsigned twice = 0;
  if (p != NONE) {
    const uint8_t fl = flags[p];
    uint32_t out = 0;
    {
      float qf[3];
      int cc[3];
      load_pt32<GD>(sorted32, p, qf);
      cell_of32<GD>(qf, g, cc);
      constexpr int UNR = VCP_UNRW;
      const int32_t myg = GROUPED ? sgroup[p] : 0;
      uint32_t mx = 0, mnk = NONE;
      // The cluster rank of a candidate is read FIRST (4 B; NONE = not expanding): a border point sits mostly
      // among non-expanding points, and once some cluster is known to reach it, further members of clusters
      // ranked between the current min and max cannot change the result -- only the remaining candidates pay
      // for the coordinate load and the binary64 test.
      for_rows<GD>(cc, g, ct, [&](uint32_t s, uint32_t e) {
        for (uint32_t j0 = s; j0 < e; j0 += UNR) {
          uint32_t rk[UNR];
#pragma unroll
          for (int u = 0; u < UNR; u++) rk[u] = rootk[min(j0 + u, e - 1)];
#pragma unroll
          for (int u = 0; u < UNR; u++) {
            const uint32_t j = j0 + u, x = rk[u];
            if (j >= e || x == NONE || (x >= mnk && x < mx)) continue;
            if (GROUPED && sgroup[j] != myg) continue;
            float rr[3];
            load_pt32<GD>(sorted32, j, rr);
            if (within_scr<GD, METRIC>(qf, rr, sc, xs, p, j, thr)) {
              mx = max(mx, x + 1u);
              mnk = min(mnk, x);
            }
          }
        }
        return true;
      });
      out = mx;
      // staged (slab) calls count only the caller's own points: list positions in [own_lo, own_lo + own_span)
      if (mx != 0 && !(fl & F_CLASSED) && sord[p] - own_lo < own_span && sord[p] < clseed[mnk]) {
        twice = 1;
        if (GROUPED) atomicAdd(&group_twice[myg], 1u);
      }
    }
    labk[p] = (out << 2) | ((fl & F_CORE) ? 1u : 0u) | ((fl & F_CLASSED) ? 2u : 0u);
  }
  if (!GROUPED) twice_add(twice, counters);
}

// The border rule from the recorded lists: a point below min_pts has recorded ALL its neighbours, so the largest /
// smallest adjacent cluster comes from <= NB rank words, no cell walk, no coordinates.  (Classed core points -- only
// possible with an isClassed input -- exit the count early and have incomplete lists: those calls keep k_border.)
template <bool GROUPED>
__global__ __launch_bounds__(TPB) void k_border_list(const int32_t* __restrict__ sgroup, const uint8_t* __restrict__ flags,
                                                    const uint32_t* __restrict__ sord, const uint32_t* __restrict__ rootk,
                                                    const uint32_t* __restrict__ clseed, uint32_t* __restrict__ labk,
                                                    unsigned long long* __restrict__ counters,
                                                    uint32_t* __restrict__ group_twice, NbrOut no, WorkList wlB) {
  const uint32_t p = wl_fetch(wlB);
  unsigned twice = 0;
  if (p != NONE) {
    const uint8_t fl = flags[p];
    const int nrec = fl >> 4;
    uint32_t mx = 0, mnk = NONE;
    const uint32_t* li = nbr_list(no, p);
    for (int k = 0; k < nrec; k++) {
      const uint32_t x = rootk[li[k]];
      if (x == NONE) continue;
      mx = max(mx, x + 1u);
      mnk = min(mnk, x);
    }
    if (mx != 0 && sord[p] < clseed[mnk]) twice = 1;  // nobody is classed on entry in these calls
    labk[p] = (mx << 2) | ((fl & F_CORE) ? 1u : 0u) | ((fl & F_CLASSED) ? 2u : 0u);
  }
  if (GROUPED) {
    // one add per group the wave's counted points fall into (the lanes of a wave are neighbours in cell order: one to three
    // blocks) instead of one per point -- global atomics execute at the memory side here: a million of them on 27 k
    // counters were 60 of this kernel's 96 us
    const int32_t g = twice ? sgroup[p] : -1;
    unsigned long long todo = __ballot(twice != 0);
    const int lane = threadIdx.x & 63;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int32_t gl = __shfl(g, leader, 64);
      const unsigned long long same = __ballot(twice && g == gl);
      if (lane == leader) atomicAdd(&group_twice[gl], (uint32_t)__popcll(same));
      todo &= ~same;
    }
  } else {
    twice_add(twice, counters);
  }
}

// labk of every position the border list does not cover: expanding points take their component's rank,
// everything else (no neighbour within eps) keeps 0
__global__ __launch_bounds__(TPB) void k_labk_rest(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ parent,
                                                  uint32_t* rootk, uint32_t* __restrict__ labk,
                                                  CellTab ct) {
  const uint32_t nin = *ct.nin;
  int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (p >= nin) return;
  const uint8_t fl = flags[p];
  // rootk is defined at roots (k_rootk); extend it to every position -- the seed rank of the point's cluster,
  // NONE for points that are not expanding -- so that the border search needs one load per candidate
  uint32_t k = NONE;
  if (fl & F_EXPAND) {
    const uint32_t r = parent[p];
    k = rootk[r];
    if (r != (uint32_t)p) rootk[p] = k;
  } else {
    rootk[p] = NONE;
  }
  if (fl & F_BCAND) return;
  const uint32_t out = (fl & F_EXPAND) ? k + 1u : 0u;
  labk[p] = (out << 2) | ((fl & F_CORE) ? 1u : 0u) | ((fl & F_CLASSED) ? 2u : 0u);
}

// ---- outputs in caller order (coalesced writes, gather from the sorted arrays) ------------------------
template <bool GROUPED>
__global__ __launch_bounds__(TPB) void k_output(int64_t n, const uint32_t* __restrict__ pos,
                                               const uint32_t* __restrict__ labk,
                                               const uint8_t* __restrict__ in_classed, const int32_t* __restrict__ group,
                                               const uint32_t* __restrict__ groupstart,
                                               const uint32_t* __restrict__ seedbits,
                                               const uint32_t* __restrict__ seedpref, int32_t cf_in,
                                               int32_t* __restrict__ labels, uint8_t* __restrict__ is_core,
                                               uint8_t* __restrict__ is_classed,
                                               unsigned long long* __restrict__ counters) {
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  unsigned unclassed = 0;
  (void)counters;
  if (i < n) {
    const uint32_t p = pos[i];
    if (p == NONE) {  // excluded from this call (grouped: outside the group range -- left untouched, another
                      // call or another rank owns that slice of the label array)
      if (GROUPED) return;
      if (!in_classed) labels[i] = 0;
      if (is_core) is_core[i] = 0;
      if (is_classed) is_classed[i] = in_classed ? in_classed[i] : 0;
    } else {
      const uint32_t w = labk[p];
      const uint32_t k1 = w >> 2;
      const bool core = w & 1u, classed = w & 2u;
      int32_t lab = 0;
      if (k1) {
        uint32_t base = GROUPED ? seed_rank(seedbits, seedpref, groupstart[group[i]]) : 0u;
        lab = cf_in + (int32_t)(k1 - base);
      }
      if (lab != 0 || !in_classed) labels[i] = lab;
      if (is_core) is_core[i] = (core && !classed) ? 1 : 0;
      if (is_classed) is_classed[i] = (classed || lab != 0) ? 1 : 0;
      if (!classed) unclassed = 1;
    }
  }
  // points not classed on entry: only worth counting when the caller passed in_classed (otherwise it is
  // every clustered point); one atomic per workgroup, spread over 32 slots (a single hot word serialises)
  if (in_classed) {
    __shared__ unsigned wcnt[TPB / 64];
    unsigned long long m1 = __ballot(unclassed);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = (unsigned)__popcll(m1);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned t = 0;
      for (int k = 0; k < TPB / 64; k++) t += wcnt[k];
      if (t) atomicAdd(&counters[4 + (blockIdx.x & 31)], (unsigned long long)t);
    }
  }
}

// per-group statistics: clusters per group and the op counter of the per-block DBImproved instances
__global__ __launch_bounds__(TPB) void k_group_stats(int32_t G, int glo, int ghi, const uint32_t* __restrict__ groupstart,
                                                    const uint32_t* __restrict__ seedbits,
                                                    const uint32_t* __restrict__ seedpref,
                                                    const uint32_t* __restrict__ group_twice,
                                                    uint32_t* __restrict__ group_nclus,
                                                    unsigned long long* __restrict__ evals, uint32_t skip_upto) {
  int g = blockIdx.x * TPB + threadIdx.x;
  if (g >= G) return;
  if (g < glo || g >= ghi) {
    if (group_nclus) group_nclus[g] = 0;
    return;
  }
  unsigned long long ng = groupstart[g + 1] - groupstart[g];
  if (ng <= skip_upto) return;  // (DbscanExt::skip_upto)
  unsigned long long kg = seed_rank(seedbits, seedpref, groupstart[g + 1]) - seed_rank(seedbits, seedpref, groupstart[g]);
  if (group_nclus) group_nclus[g] = (uint32_t)kg;
  unsigned long long ev = ng * (ng + kg + group_twice[g]);
  if (ev) atomicAdd(evals, ev);
}

// ---- the whole cloud fits inside one eps-ball: every pair is within eps -------------------------------
// (decided on the host from the bounding box; without this, one cell would hold everything and the search
// would be O(n^2)).  count = n for every point, so either nobody is core, or everybody is and the unclassed
// points form ONE cluster cf_in+1 that also relabels every classed point (BaseClass/DBImproved.cs:87).
__global__ __launch_bounds__(TPB) void k_all_pairs(const uint8_t* __restrict__ in_classed, int64_t n, int core,
                                                  int have_seed, int32_t cf_in, int32_t* __restrict__ labels,
                                                  uint8_t* __restrict__ is_core, uint8_t* __restrict__ is_classed) {
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const bool cls = in_classed && in_classed[i];
  const bool hit = core && have_seed;
  if (hit) labels[i] = cf_in + 1;
  else if (!in_classed) labels[i] = 0;
  if (is_core) is_core[i] = (core && !cls) ? 1 : 0;
  if (is_classed) is_classed[i] = (cls || hit) ? 1 : 0;
}

// ---- eps < 0 or NaN: nobody has a neighbour, not even itself ---------------------------------------
__global__ __launch_bounds__(TPB) void k_unclassed_flag(const uint8_t* __restrict__ in_classed, uint32_t* __restrict__ f,
                                                       int64_t n) {
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (i < n) f[i] = (in_classed && in_classed[i]) ? 0u : 1u;
}
__global__ __launch_bounds__(TPB) void k_degenerate(const uint8_t* __restrict__ in_classed, const uint32_t* __restrict__ rk,
                                                   int64_t n, int32_t cf_in, int all_core, int fresh,
                                                   int32_t* __restrict__ labels, uint8_t* __restrict__ is_core,
                                                   uint8_t* __restrict__ is_classed) {
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  bool cls = in_classed && in_classed[i];
  if (!cls && all_core) labels[i] = cf_in + 1 + (int32_t)rk[i];
  else if (fresh) labels[i] = 0;
  if (is_core) is_core[i] = (!cls && all_core) ? 1 : 0;
  if (is_classed) is_classed[i] = cls ? 1 : 0;  // expandCluster never marks the seed itself
}

// ---- min_pts <= 0 with non-finite coordinates -------------------------------------------------------------
// A point with a NaN / infinite coordinate has no neighbour, not even itself.  With min_pts <= 0 it is still "core"
// (0 >= minPts, BaseClass/DBImproved.cs:105): it seeds a cluster of its own, gets the id (:58), but expandCluster's
// loop over the EMPTY neighbour list never marks it classed (:63-65).  The label word has no spare bit for this, and
// the host knows both conditions before the first kernel, so the rare case gets its own pass over caller order.
template <int GD>
__global__ __launch_bounds__(TPB) void k_lonely_seeds(const double* __restrict__ c, int64_t n, int stride,
                                                     const uint8_t* __restrict__ in_classed,
                                                     uint8_t* __restrict__ is_classed,
                                                     unsigned long long* __restrict__ lonely_seeds) {
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  bool finite = true;
#pragma unroll
  for (int a = 0; a < GD; a++) finite = finite && isfinite(c[i * stride + a]);
  if (finite) return;
  const bool cls = in_classed && in_classed[i];
  if (is_classed) is_classed[i] = cls ? 1 : 0;
  // such a seed is queried once (main loop), not twice: it is never popped from its own, empty, list
  if (!cls) atomicAdd(lonely_seeds, 1ull);
}

// ---- staged (slab) calls: exact DBSCAN over several GPUs (distributed.exact_slabs) ---------------------
// After the component build the caller needs, per point, the seed of its LOCAL component (smallest ord) and
// the list of local components; it resolves them against the other ranks' and comes back with, per local
// component, an index into a table of global clusters sorted by cluster id (vcp_slab_finish).
__global__ __launch_bounds__(TPB) void k_slab_count(const uint32_t* __restrict__ parent, const uint8_t* __restrict__ flags,
                                                   CellTab ct,
                                                   uint32_t* __restrict__ blkcnt) {
  const uint32_t nin = *ct.nin;
  int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const bool root = p < nin && (flags[p] & F_EXPAND) && parent[p] == (uint32_t)p;
  __shared__ unsigned wc[TPB / 64];
  const unsigned long long m = __ballot(root);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = 0;
    for (int k = 0; k < TPB / 64; k++) a += wc[k];
    blkcnt[blockIdx.x] = a;
  }
}

__global__ __launch_bounds__(TPB) void k_slab_fill(const uint32_t* __restrict__ parent, const uint8_t* __restrict__ flags,
                                                  const uint32_t* __restrict__ minord,
                                                  CellTab ct,
                                                  const uint32_t* __restrict__ blkscan, uint32_t* __restrict__ comps) {
  const uint32_t nin = *ct.nin;
  int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const bool root = p < nin && (flags[p] & F_EXPAND) && parent[p] == (uint32_t)p;
  __shared__ unsigned wc[TPB / 64];
  const unsigned long long m = __ballot(root);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wc[w] = (unsigned)__popcll(m);
  __syncthreads();
  if (root) {
    unsigned before = 0;
    for (int k = 0; k < w; k++) before += wc[k];
    before += (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    comps[blkscan[blockIdx.x] + before] = minord[p];
  }
}

__global__ __launch_bounds__(TPB) void k_slab_out(int64_t n, const uint32_t* __restrict__ pos,
                                                 const uint8_t* __restrict__ flags, const uint32_t* __restrict__ parent,
                                                 const uint32_t* __restrict__ minord, uint32_t* __restrict__ rep,
                                                 uint8_t* __restrict__ is_core) {
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = pos[i];
  const uint8_t fl = flags[p];
  rep[i] = (fl & F_EXPAND) ? minord[parent[p]] : NONE;
  if (is_core) is_core[i] = (fl & F_CORE) ? 1 : 0;
}

// per root: index of its global cluster in the caller's table (binary search of the local seed)
__global__ __launch_bounds__(TPB) void k_slab_rootk(const uint32_t* __restrict__ parent, const uint8_t* __restrict__ flags,
                                                   const uint32_t* __restrict__ minord,
                                                   CellTab ct,
                                                   const uint32_t* __restrict__ map_rep, const uint32_t* __restrict__ map_k,
                                                   uint32_t nmap, uint32_t* __restrict__ rootk,
                                                   unsigned long long* __restrict__ missing) {
  const uint32_t nin = *ct.nin;
  int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (p >= nin) return;
  if ((flags[p] & F_EXPAND) && parent[p] == (uint32_t)p) {
    const uint32_t key = minord[p];
    uint32_t lo = 0, hi = nmap;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (map_rep[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo < nmap && map_rep[lo] == key) {
      rootk[p] = map_k[lo];
    } else {
      rootk[p] = 0;
      atomicAdd(missing, 1ull);
    }
  }
}

__global__ __launch_bounds__(TPB) void k_slab_output(int64_t n, const uint32_t* __restrict__ pos,
                                                    const uint32_t* __restrict__ labk, const int32_t* __restrict__ tab_gid,
                                                    int32_t* __restrict__ labels, uint8_t* __restrict__ is_classed) {
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const uint32_t k1 = labk[pos[i]] >> 2;
  const int32_t lab = k1 ? tab_gid[k1 - 1] : 0;
  labels[i] = lab;
  if (is_classed) is_classed[i] = lab != 0;
}

// largest binary64 s such that sqrt(s) <= eps (sqrt is correctly rounded and monotone), so that
// `sqrt(s) <= eps` can be tested as `s <= thr` without a device sqrt.
double l2_threshold(double eps) {
  if (std::isinf(eps)) return eps;
  double t = eps * eps;
  if (std::isinf(t)) t = std::numeric_limits<double>::max();
  while (std::sqrt(t) > eps) t = std::nextafter(t, 0.0);
  for (;;) {
    double u = std::nextafter(t, std::numeric_limits<double>::infinity());
    if (std::isinf(u) || std::sqrt(u) > eps) break;
    t = u;
  }
  return t;
}

// Screen bounds (see grid_common.hpp).  E = largest |coordinate - grid origin| over the finite input (from the TRUE
// bounding box: the grid range may have been trimmed), u = 2^-24.  A screening copy is off by at most u E (1 + u); a
// coordinate difference a of two copies satisfies |a - d| <= alpha + u |d| with alpha = 2 u E (1 + 2u).
//   L1:  |s - m| <= A + B m,  A = 2 alpha (1 + u),  B = 2 u (1 + u)        (m = exact |dx| + |dy|, s = binary32 value)
//   L2:  |s - m| <= 2 alpha sqrt(GD m)(1 + u) + GD alpha^2 + 6 u m (1 + u) (m = exact sum of squares)
// plus 1e-14 m for the roundings of the binary64 reference expression itself.  err() grows with m, so s > thr + err(thr)
// proves m > thr; m - err(m) grows with m as long as sqrt(m) dominates alpha, so s <= thr - err(thr) proves m <= thr (when
// it does not, lo = -1: nothing is accepted on binary32 evidence).  lo is rounded down, hi up.
Screen screen_bounds(int metric, int gd, double thr, double E) {
  Screen sc;
  sc.lo = -1.0f;
  sc.hi = INFINITY;
  if (!(thr >= 0.0) || !std::isfinite(E)) return sc;  // NaN / negative thresholds and unbounded clouds: all exact
  // The error model below is RELATIVE rounding (u = 2^-24).  It does not hold where binary32 underflows (conversions
  // and squares of magnitude below 2^-126 carry an absolute error of 2^-150) or overflows (a copy or a square beyond
  // FLT_MAX is inf): thresholds that small (thr = 0 is fine: identical points have identical copies, value 0) and
  // clouds that large take the exact path for every candidate.
  if ((thr > 0.0 && thr < 1e-30) || E > 1e18) return sc;
  const double u = 5.9604644775390625e-08;
  const double alpha = 2.0 * u * E * (1.0 + 2.0 * u);
  double err;
  bool lo_ok = true;
  if (metric == VCP_L1_2D) {
    err = 2.0 * alpha * (1.0 + u) + 2.0 * u * (1.0 + u) * thr;
  } else {
    err = 2.0 * alpha * std::sqrt((double)gd * thr) * (1.0 + u) + (double)gd * alpha * alpha + 6.0 * u * (1.0 + u) * thr;
    lo_ok = std::sqrt(thr) > 4.0 * alpha * std::sqrt((double)gd);
  }
  err = err * 1.0625 + 1e-14 * thr + 1e-300;
  const double lo = thr - err, hi = thr + err;
  if (std::isfinite(hi)) {
    float h = (float)hi;
    if ((double)h < hi) h = std::nextafterf(h, INFINITY);
    sc.hi = h;
  }
  if (lo_ok && lo >= 0.0 && std::isfinite(lo)) {
    float l = (float)lo;
    if ((double)l > lo) l = std::nextafterf(l, -INFINITY);
    sc.lo = l;
  } else if (lo_ok && std::isinf(thr) && thr > 0) {
    sc.lo = 3.4028234663852886e38f;  // eps = +inf: every finite value is inside
  }
  return sc;
}

// workgroups per band of a work list: a band holds at most ceil(n / (8*LCHUNK)) entries
unsigned list_perblk(int64_t n) {
  const int64_t band = (n + 8 * LCHUNK - 1) / (8 * LCHUNK) + 1;
  return (unsigned)((band + TPB - 1) / TPB);
}

// GD = dimension of the metric (grid and sorted copy); `stride` = doubles per input point
template <int GD, int METRIC, bool GROUPED>
int run_dbscan(vcp_ctx* ctx, const double* d_coords, int64_t n, int stride, double eps, int min_pts, int32_t cf_in,
               const uint8_t* d_in_classed, int32_t* d_labels, uint8_t* d_is_core, uint8_t* d_is_classed,
               int32_t* cf_out, int64_t* dist_evals, const DbscanExt* ext) {
  hipStream_t st = ctx->stream;
  const unsigned nb = vcp_blocks(n, TPB);
  const int32_t* d_group = GROUPED ? ext->d_group : nullptr;
  const uint32_t* d_ord = ext ? ext->d_ord : nullptr;
  const int G = GROUPED ? ext->G : 0;
  const int glo = GROUPED ? ext->only_lo : 0;
  const int ghi = GROUPED ? (ext->only_hi < 0 ? G : ext->only_hi) : 0;

  // 1. bounds over finite coordinates
  vcp_phase(ctx, "bounds");
  const int rb = (int)vcp_blocks(n, TPB, 1024);
  VCP_TRY(vcp_ensure(ctx, ctx->b_misc, (size_t)(rb * 8 + 96) * sizeof(double)));
  double* d_part = ctx->b_misc.as<double>();
  double* d_bounds = d_part + (size_t)rb * 8;
  double* h = reinterpret_cast<double*>(ctx->pinned);
  if (ext && ext->h_bbox) {  // the caller has seen every point: finite, inside this box
    for (int a = 0; a < 6; a++) h[a] = ext->h_bbox[a];
    h[6] = 0.0;
  } else {
    hipLaunchKernelGGL((k_bounds<GD, GROUPED>), dim3(rb), dim3(TPB), 0, st, d_coords, n, stride, d_group, glo, ghi, d_part);
    hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(TPB), 0, st, d_part, rb, d_bounds);
    VCP_HIP(ctx, hipMemcpyAsync(h, d_bounds, 7 * sizeof(double), hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
  }
  const bool all_finite = h[6] == 0.0;

  const double thr = (METRIC == VCP_L1_2D) ? eps : l2_threshold(eps);
  if (!GROUPED && all_finite && !(ext && ext->slab)) {
    // monotone rounding: |dx| <= hi-lo on every axis, so the box measure bounds every pair's distance form
    const double wx = h[3] - h[0], wy = h[4] - h[1], wz = GD == 3 ? h[5] - h[2] : 0.0;
    const double box = METRIC == VCP_L1_2D ? std::fabs(wx) + std::fabs(wy)
                       : METRIC == VCP_L2_2D ? wx * wx + wy * wy : wx * wx + wy * wy + wz * wz;
    if (box <= thr) {
      vcp_phase(ctx, "all_pairs");
      VCP_TRY(vcp_ensure(ctx, ctx->b_seedflag, (size_t)(n + 2) * 4));
      uint32_t* f = ctx->b_seedflag.as<uint32_t>();
      uint32_t* d_tot = reinterpret_cast<uint32_t*>(d_bounds + 8);
      unsigned long long unclassed = (unsigned long long)n;
      if (d_in_classed) {
        hipLaunchKernelGGL(k_unclassed_flag, dim3(nb), dim3(TPB), 0, st, d_in_classed, f, n);
        VCP_TRY(vcp_exclusive_scan_u32(ctx, f, f, n, d_tot));
        uint32_t* hu = reinterpret_cast<uint32_t*>(ctx->pinned) + 32;
        VCP_HIP(ctx, hipMemcpyAsync(hu, d_tot, 4, hipMemcpyDeviceToHost, st));
        VCP_HIP(ctx, hipStreamSynchronize(st));
        unclassed = hu[0];
      }
      const int core = n >= (int64_t)min_pts;
      const int have_seed = unclassed > 0;
      hipLaunchKernelGGL(k_all_pairs, dim3(nb), dim3(TPB), 0, st, d_in_classed, n, core, have_seed, cf_in, d_labels,
                         d_is_core, d_is_classed);
      VCP_HIP(ctx, hipGetLastError());
      VCP_TRY(vcp_phase_finish(ctx));
      VCP_HIP(ctx, hipStreamSynchronize(st));
      const int made = core && have_seed;
      if (cf_out) *cf_out = cf_in + made;
      if (dist_evals) *dist_evals = (int64_t)(unclassed + (unsigned long long)made) * n;
      return VCP_OK;
    }
  }

  // 2. grid geometry (host): cell edge a hair above eps; coarsen until the cell count fits
  double bbox[6];  // the true bounding box of the finite coordinates (the grid range below may get trimmed)
  for (int a = 0; a < 3; a++) {
    const bool have = a < GD && h[3 + a] >= h[a];
    bbox[a] = have ? h[a] : 0.0;
    bbox[3 + a] = have ? h[3 + a] : 0.0;
  }
  GridP g;
  double range = 0.0;
  for (int a = 0; a < 3; a++) {
    double lo = a < GD ? h[a] : 0.0, hi = a < GD ? h[3 + a] : 0.0;
    if (!(hi >= lo)) lo = hi = 0.0;  // no finite value on this axis
    g.mn[a] = lo;
    h[a] = lo;
    h[3 + a] = hi;
    range = std::fmax(range, hi - lo);
  }
  double cellw = eps * (1.0 + 1.0 / 1048576.0);
  static const int64_t cells_per_point = [] {  // (VCP_CELL_BUDGET: test switch)
    const char* e = getenv("VCP_CELL_BUDGET");
    const long v = e ? atol(e) : 32;
    return (int64_t)(v < 1 ? 1 : v > 4096 ? 4096 : v);
  }();
  int64_t budget = n * cells_per_point;
  if (budget < (1 << 16)) budget = 1 << 16;
  if (budget > ((int64_t)1 << 31) - 16) budget = ((int64_t)1 << 31) - 16;  // cell ids and ncells + 1 stay in 31 bits
  // Robust range.  Cell indices are clamped to the grid, so ANY origin and extent give correct results (a point
  // beyond the grid shares the edge cell with its neighbours); the full bounding box is only the natural choice.
  // When the eps-grid over it does not fit the budget, the range is trimmed to mean +- 8 sigma of the points
  // inside it, repeatedly: a handful of far outliers then end up in the edge cells instead of coarsening the
  // grid for the whole cloud.
  {
    // (the cell edge is at least range / 2^20 -- eps = 0 asks for exact duplicates only -- and the trimming has to see
    // that edge, not a zero one: with eps = 0 and a few far outliers the untrimmed range put 700 k points into ONE cell,
    // 1.2 s instead of 1 ms)
    auto cells_needed = [&](const double* lo, const double* hi) {
      double r = 0.0;
      for (int a = 0; a < GD; a++) r = std::fmax(r, hi[a] - lo[a]);
      const double cw = std::fmax(cellw, r / 1048575.0);
      double c = 1.0;
      for (int a = 0; a < GD; a++) c *= std::floor((hi[a] - lo[a]) / cw) + 1.0;
      return c;
    };
    double lo[3] = {h[0], h[1], h[2]}, hi[3] = {h[3], h[4], h[5]};
    const bool no_trim = ext && ext->no_trim;
    for (int it = 0; it < 8 && !no_trim && cellw >= 0.0 && std::isfinite(cellw) && !(cells_needed(lo, hi) <= (double)budget); it++) {
      Range3 R;
      for (int a = 0; a < 3; a++) {
        R.lo[a] = lo[a];
        R.hi[a] = hi[a];
        R.mid[a] = 0.5 * lo[a] + 0.5 * hi[a];
      }
      VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, (size_t)rb * 9 * sizeof(double)));
      double* d_mom = ctx->b_aux0.as<double>();
      hipLaunchKernelGGL((k_moments<GD, GROUPED>), dim3(rb), dim3(TPB), 0, st, d_coords, n, stride, d_group, glo, ghi, R,
                         d_mom);
      std::vector<double> hm((size_t)rb * 9);
      VCP_HIP(ctx, hipMemcpyAsync(hm.data(), d_mom, hm.size() * sizeof(double), hipMemcpyDeviceToHost, st));
      VCP_HIP(ctx, hipStreamSynchronize(st));
      bool changed = false;
      for (int a = 0; a < GD; a++) {
        double cnt = 0, s1 = 0, s2 = 0;
        for (int b = 0; b < rb; b++) {
          cnt += hm[(size_t)b * 9 + 3 * a];
          s1 += hm[(size_t)b * 9 + 3 * a + 1];
          s2 += hm[(size_t)b * 9 + 3 * a + 2];
        }
        if (!(cnt > 0)) continue;
        const double mean = s1 / cnt, var = std::fmax(s2 / cnt - mean * mean, 0.0);
        const double c0 = R.mid[a] + mean, w = 8.0 * std::sqrt(var) + 4.0 * cellw;
        const double nlo = std::fmax(lo[a], c0 - w), nhi = std::fmin(hi[a], c0 + w);
        if (nlo > lo[a] || nhi < hi[a]) changed = true;
        if (nlo <= nhi) {
          lo[a] = nlo;
          hi[a] = nhi;
        }
      }
      if (!changed) break;
    }
    for (int a = 0; a < GD; a++) {
      h[a] = lo[a];
      h[3 + a] = hi[a];
      g.mn[a] = lo[a];
    }
    range = 0.0;
    for (int a = 0; a < GD; a++) range = std::fmax(range, h[3 + a] - h[a]);
  }
  const double min_w = range / 1048575.0;
  if (!(cellw >= min_w)) cellw = min_w;
  if (!(cellw > 0.0)) cellw = 1.0;
  // The cells are taken from binary32 relative coordinates (grid_common.hpp: rel32): each is within 2^-23.9 of its
  // magnitude of the exact difference, and wherever the cell index is not clamped that magnitude is below
  // range + 2 cells.  Two points within eps on an axis must end up at most one cell apart: cellw >= eps + both roundings.
  {
    const double need = (eps + (range + 4.0 * cellw) * (1.0 / 4194304.0)) * (1.0 + 1.0 / 1048576.0);
    if (std::isfinite(need) && !(cellw >= need)) cellw = need;
  }
  int64_t ncells = 0;
  for (int it = 0; it < 400; it++) {
    ncells = 1;
    for (int a = 0; a < 3; a++) {
      double ext_a = a < GD ? (h[3 + a] - h[a]) / cellw : 0.0;
      int64_t d = std::isfinite(ext_a) ? (int64_t)ext_a + 1 : 1;
      if (d > 1048576) d = 1048576;
      g.D[a] = (int)d;
      ncells *= d;
    }
    if (ncells <= budget || std::isinf(cellw)) break;
    cellw *= 1.25;
  }
  if (ncells > budget) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "grid does not fit the cell budget");
  // largest |coordinate - grid origin| over the finite input: the true bounding box, not the (possibly trimmed) grid
  double Emax = 0.0;
  for (int a = 0; a < GD; a++) Emax = std::fmax(Emax, std::fmax(std::fabs(bbox[a] - g.mn[a]), std::fabs(bbox[3 + a] - g.mn[a])));
  if (!std::isfinite(Emax))
    return vcp_fail(ctx, VCP_ERR_UNSUPPORTED, "the cloud's extent overflows binary64 (coordinate - origin is infinite)");
  // binary32 keeps the relative precision the cell edge and the screen bounds count on only inside its normal range:
  // a cloud whose extent is far outside it (coordinates x 1e150: the copies would be inf, every point in an edge cell;
  // x 1e-40: subnormal copies) is binned and screened on (coordinate - origin) * 2^k with the extent brought to [1, 2).
  // A power of two is exact, so everything above holds for the scaled values; clouds of ordinary size keep k = 0.
  g.scale = 1.0;
  if (Emax > 1e30 || (Emax > 0.0 && Emax < 1e-20)) g.scale = std::ldexp(1.0, -std::ilogb(Emax));
  g.inv_h = std::isinf(cellw) ? 0.0 : 1.0 / (cellw * g.scale);
  g.ncells = (uint32_t)ncells;

  // 3. workspace
  // the cell table (grid_common.hpp: CellTab): 16 bytes per word of 32 cells, the full starts of the populous words, and
  // two counters
  const size_t ctw = vcp_ct_words(g.ncells), ctd = vcp_ct_dense_cap(n, g.ncells);
  VCP_TRY(vcp_ensure(ctx, ctx->b_ctw, ctw * 16 + 64));
  VCP_TRY(vcp_ensure(ctx, ctx->b_ctd, ctd * 128));
  VCP_TRY(vcp_ensure(ctx, ctx->b_cellof, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_pos, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_sorted32, (size_t)n * (GD == 2 ? 2 : 4) * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_sidx, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_flags, (size_t)n));
  VCP_TRY(vcp_ensure(ctx, ctx->b_parent, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_minord, (size_t)n * 4));
  const uint32_t nw = (uint32_t)(n / 32 + 2);  // bitmap words: positions 0..n (rank(n) = seed total)
  VCP_TRY(vcp_ensure(ctx, ctx->b_seedflag, ((size_t)nw * 2 + 8) * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_rootcl, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_clseed, (size_t)n * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_labk, (size_t)n * 4));
  if (GROUPED) VCP_TRY(vcp_ensure(ctx, ctx->b_sgroup, (size_t)n * 4));
  uint4* ctwords = ctx->b_ctw.as<uint4>();
  uint32_t* ctcount = reinterpret_cast<uint32_t*>(ctx->b_ctw.as<char>() + ctw * 16);  // [0] populous words, [1] points in the grid
  const CellTab ct{ctwords, ctx->b_ctd.as<uint32_t>(), ctcount + 1};
  uint32_t* cellof = ctx->b_cellof.as<uint32_t>();
  uint32_t* pos = ctx->b_pos.as<uint32_t>();
  float* sorted32 = ctx->b_sorted32.as<float>();
  static const bool screen_off = getenv("VCP_NO_SCREEN") != nullptr;
  // the screen compares values of the SCALED copies: threshold and extent in the same units
  Screen sc = screen_bounds(METRIC, GD, METRIC == VCP_L1_2D ? thr * g.scale : thr * g.scale * g.scale, Emax * g.scale);
  if (screen_off) sc = Screen{-1.0f, INFINITY};
  uint32_t* sord = ctx->b_sidx.as<uint32_t>();
  uint8_t* flags = ctx->b_flags.as<uint8_t>();
  uint32_t* parent = ctx->b_parent.as<uint32_t>();
  uint32_t* minord = ctx->b_minord.as<uint32_t>();
  uint32_t* seedflag = ctx->b_seedflag.as<uint32_t>();  // seed bitmap [nw]
  uint32_t* seedpref = seedflag + ((nw + 3) & ~3u);      // per-word prefix [nw] (16-B aligned)
  uint32_t* rootk = ctx->b_rootcl.as<uint32_t>();
  uint32_t* clseed = ctx->b_clseed.as<uint32_t>();
  uint32_t* labk = ctx->b_labk.as<uint32_t>();
  int32_t* sgroup = GROUPED ? ctx->b_sgroup.as<int32_t>() : nullptr;
  // [0] lonely seeds (min_pts <= 0, non-finite points), [1] border points queried twice, [2] seed total (u32), [3] grouped evals, [4..36) unclassed slots
  unsigned long long* counters = reinterpret_cast<unsigned long long*>(d_bounds + 8);
  uint32_t* d_total = reinterpret_cast<uint32_t*>(counters + 2);

  // 4. cell order: two-level partition that carries the binary32 coordinates and emits the cell table (gridbuild.hip);
  //    its output pass needs no caller-order -> cell-order map (pos).  The binary64 coordinates stay in the caller's
  //    array and are read by index where a screened pair needs the exact test.
  // flags before the core count: the build stores the callers' isClassed bits; without them the core count starts every
  // byte itself and nothing has to be there
  const bool flags_set = d_in_classed != nullptr;
  const bool part_out = !GROUPED && !d_ord && !(ext && ext->slab) && n <= ((int64_t)1 << 27);
  {
    GridBuildArgs ga;
    ga.d_coords = d_coords;
    ga.n = n;
    ga.stride = stride;
    ga.gd = GD;
    ga.g = g;
    ga.d_group = d_group;
    ga.glo = glo;
    ga.ghi = ghi;
    ga.d_ord = d_ord;
    ga.d_in_classed = d_in_classed;
    ga.ctwords = ctwords;
    ga.ctdense = ctx->b_ctd.as<uint32_t>();
    ga.ctcount = ctcount;
    ga.sidx = d_ord ? cellof : nullptr;  // the point's index where sord holds the caller's list position instead
    ga.sorted32 = sorted32;
    ga.sord = sord;
    ga.sgroup = sgroup;
    ga.flags = flags;
    ga.pos = part_out ? nullptr : pos;
    VCP_TRY(vcp_grid_build_partition(ctx, ga));
  }
  const ExactSrc xs{d_coords, d_ord ? cellof : sord, stride};  // staged calls: kept for vcp_slab_finish (SlabState.xs)

  // 5. core flags + work lists (expanding points; non-core points that have a neighbour)
  vcp_phase(ctx, "core_count");
  WorkList wlE, wlB;
  VCP_TRY(vcp_ensure(ctx, ctx->b_wl, ((size_t)n * 2 + (size_t)(nb + 2) * 2) * 4 + 256));
  uint32_t* blkE = ctx->b_wl.as<uint32_t>();
  uint32_t* blkB = blkE + (nb + 2);
  wlE.list = blkB + (nb + 2);
  wlB.list = wlE.list + n;
  wlE.scan = blkE;
  wlB.scan = blkB;
  wlE.nblk = wlB.nblk = nb;
  wlE.perblk = wlB.perblk = list_perblk(n);
  const unsigned nbl = 8u * LCHUNK * wlE.perblk;  // list kernels: see wl_fetch
  // neighbour lists (see NbrOut): off when the caller passes isClassed (classed core points need the full search), for
  // staged calls (vcp_slab_finish searches again with the resolved ids) and for min_pts outside 2..16
  static const bool lists_off = getenv("VCP_NO_LISTS") != nullptr;
  NbrOut no{nullptr, nullptr, 0, 0};
  if (!lists_off && !d_in_classed && !(ext && ext->slab) && min_pts >= 2 && min_pts <= 16) {
    no.NB = min_pts - 1;
    VCP_TRY(vcp_ensure(ctx, ctx->b_nbr, (size_t)no.NB * (size_t)nb * TPB * 4));
    VCP_TRY(vcp_ensure(ctx, ctx->b_nboff, (size_t)nb * TPB * 2));
    no.nbr = ctx->b_nbr.as<uint32_t>();
    no.off = ctx->b_nboff.as<uint16_t>();
    static const int cap_env = getenv("VCP_CORE_CAP") ? atoi(getenv("VCP_CORE_CAP")) : -1;  // test switch
    no.core_cap = (GD == 2 && (uint64_t)n < (uint64_t)g.ncells) ? std::min(no.NB, CORE_LIST) : no.NB;
    if (cap_env >= 0) no.core_cap = std::min(no.NB, cap_env);
  }
  const size_t lds_nb = (size_t)no.NB * TPB * 4;
  static const bool core_global = getenv("VCP_CORE_GLOBAL") != nullptr;  // test switch: grouped calls through k_core
  if constexpr (GD == 2) {
    if (!GROUPED || !core_global)
      hipLaunchKernelGGL((k_core_lds<GD, METRIC, GROUPED>), dim3(nb), dim3(TPB), 0, st, xs, g, thr, min_pts, ct, sgroup, flags,
                         parent, minord, blkE, blkB, no, sorted32, sc, flags_set);
    else
      hipLaunchKernelGGL((k_core<GD, METRIC, GROUPED>), dim3(nb), dim3(TPB), 2 * lds_nb, st, xs, g, thr, min_pts, ct,
                         sgroup, flags, parent, minord, blkE, blkB, no, sorted32, sc, flags_set);
  } else
    hipLaunchKernelGGL((k_core<GD, METRIC, GROUPED>), dim3(nb), dim3(TPB), 2 * lds_nb, st, xs, g, thr, min_pts, ct,
                       sgroup, flags, parent, minord, blkE, blkB, no, sorted32, sc, flags_set);
  // ONE scan over both count arrays (they are adjacent): the B half comes out offset by everything before it, which
  // its readers take off again (scan[0]); the two pad words between the halves are never written and cancel the same way
  VCP_TRY(vcp_exclusive_scan_u32(ctx, blkE, blkE, 2 * ((int64_t)nb + 2), nullptr));
  hipLaunchKernelGGL(k_wl_fill, dim3(nb), dim3(TPB), 0, st, flags, ct, blkE, blkB, wlE.list, wlB.list,
                     seedflag, nw, counters, GROUPED ? ext->d_group_twice : nullptr, (uint32_t)(GROUPED ? G : 0));

  // 6. components of the expanding points
  vcp_phase(ctx, "union");
  // phases 1-2 pay for their extra search pass in 2-D (3 rows); in 3-D (9 rows) they do not (measured: +28 %)
  const bool pre = GD == 2 || no.NB > 0;  // with lists the forest costs no search, so it pays in 3-D too
  if (no.NB > 0) {
    hipLaunchKernelGGL(k_union_init_list, dim3(nbl), dim3(TPB), 0, st, flags, parent, no, wlE);
    static const int join_all = getenv("VCP_JOIN_ALL") ? atoi(getenv("VCP_JOIN_ALL")) : -1;  // test switch
    const bool all = join_all >= 0 ? join_all != 0 : (uint64_t)n >= (uint64_t)g.ncells;
    if (all) hipLaunchKernelGGL(k_flatten0<2>, dim3(nbl), dim3(TPB), 0, st, parent, wlE, flags, no);
    else hipLaunchKernelGGL(k_flatten0<1>, dim3(nbl), dim3(TPB), 0, st, parent, wlE, flags, no);
    hipLaunchKernelGGL(k_flatten0<0>, dim3(nbl), dim3(TPB), 0, st, parent, wlE, flags, no);
  } else if (GD == 2) {
    hipLaunchKernelGGL((k_union_init<GD, METRIC, GROUPED>), dim3(nbl), dim3(TPB), 0, st, xs, g, thr, ct, sgroup,
                       flags, parent, wlE, sorted32, sc);
    hipLaunchKernelGGL(k_flatten0<0>, dim3(nbl), dim3(TPB), 0, st, parent, wlE, flags, no);
  }
  const bool dense = pre && (uint64_t)n > (uint64_t)DENSE_PER_CELL * g.ncells;
  if (dense) {
    VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, ((size_t)nb * (TPB / 64) + 2) * 4));  // one word per wave of k_chunkroot
    uint32_t* chunkroot = ctx->b_aux0.as<uint32_t>();
    hipLaunchKernelGGL(k_chunkroot, dim3(nb), dim3(TPB), 0, st, parent, ct, chunkroot);
    hipLaunchKernelGGL((k_union<GD, METRIC, GROUPED, true, true>), dim3(nbl), dim3(TPB), 0, st, xs, g, thr, ct, sgroup,
                       parent, wlE, sorted32, sc, chunkroot);
  } else if (pre)
    hipLaunchKernelGGL((k_union<GD, METRIC, GROUPED, true>), dim3(nbl), dim3(TPB), 0, st, xs, g, thr, ct, sgroup,
                       parent, wlE, sorted32, sc);
  else
    hipLaunchKernelGGL((k_union<GD, METRIC, GROUPED, false>), dim3(nbl), dim3(TPB), 0, st, xs, g, thr, ct, sgroup,
                       parent, wlE, sorted32, sc);
  vcp_phase(ctx, "flatten_number");
  if ((uint64_t)n >= (uint64_t)g.ncells)
    hipLaunchKernelGGL(k_flatten<true>, dim3(nbl), dim3(TPB), 0, st, parent, sord, minord, wlE);
  else
    hipLaunchKernelGGL(k_flatten<false>, dim3(nbl), dim3(TPB), 0, st, parent, sord, minord, wlE);
  if (!GROUPED && ext && ext->slab) {
    // staged call: hand the local components to the caller and keep the grid state for vcp_slab_finish
    vcp_phase(ctx, "slab_components");
    hipLaunchKernelGGL(k_slab_count, dim3(nb), dim3(TPB), 0, st, parent, flags, ct, blkE);
    VCP_TRY(vcp_exclusive_scan_u32(ctx, blkE, blkE, (int64_t)nb + 1, nullptr));
    hipLaunchKernelGGL(k_slab_fill, dim3(nb), dim3(TPB), 0, st, parent, flags, minord, ct, blkE, clseed);
    hipLaunchKernelGGL(k_slab_out, dim3(nb), dim3(TPB), 0, st, n, pos, flags, parent, minord, ext->d_slab_rep, d_is_core);
    VCP_HIP(ctx, hipGetLastError());
    uint32_t* hn = reinterpret_cast<uint32_t*>(ctx->pinned) + 64;
    VCP_HIP(ctx, hipMemcpyAsync(hn, blkE + nb, 4, hipMemcpyDeviceToHost, st));
    VCP_TRY(vcp_phase_finish(ctx));
    VCP_HIP(ctx, hipStreamSynchronize(st));
    if (!ctx->slab) ctx->slab = new SlabState();
    SlabState& ss = *ctx->slab;
    ss.valid = true;
    ss.gd = GD;
    ss.metric = METRIC;
    ss.n = n;
    ss.n_comp = hn[0];
    ss.g = g;
    ss.thr = thr;
    ss.sc = sc;
    ss.xs = xs;
    ss.nb = nb;
    if (cf_out) *cf_out = (int32_t)hn[0];
    return VCP_OK;
  }
  hipLaunchKernelGGL(k_seedflag, dim3(nbl), dim3(TPB), 0, st, parent, minord, seedflag, wlE);
  hipLaunchKernelGGL(k_seed_popc, dim3(vcp_blocks(nw, TPB)), dim3(TPB), 0, st, seedflag, nw, seedpref);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, seedpref, seedpref, nw, d_total));
  hipLaunchKernelGGL(k_rootk, dim3(nbl), dim3(TPB), 0, st, parent, minord, seedflag, seedpref, rootk, clseed, wlE);

  // 7. border rule, then outputs in caller order
  vcp_phase(ctx, "border");
  hipLaunchKernelGGL(k_labk_rest, dim3(nb), dim3(TPB), 0, st, flags, parent, rootk, labk, ct);
  if (no.NB > 0)
    hipLaunchKernelGGL(k_border_list<GROUPED>, dim3(nbl), dim3(TPB), 0, st, sgroup, flags, sord, rootk, clseed, labk, counters,
                       GROUPED ? ext->d_group_twice : nullptr, no, wlB);
  else
    hipLaunchKernelGGL((k_border<GD, METRIC, GROUPED>), dim3(nbl), dim3(TPB), 0, st, xs, g, thr, ct, sgroup, flags,
                       parent, sord, rootk, clseed, labk, counters, GROUPED ? ext->d_group_twice : nullptr, wlB, 0u, NONE,
                       sorted32, sc);
  if (part_out) {
    GridOutputArgs oa;
    oa.n = n;
    oa.sord = sord;
    oa.labk = labk;
    oa.have_in_classed = d_in_classed != nullptr;
    oa.cf_in = cf_in;
    oa.labels = d_labels;
    oa.is_core = d_is_core;
    oa.is_classed = d_is_classed;
    oa.counters = counters;
    VCP_TRY(vcp_grid_output_partition(ctx, oa));
  } else {
    vcp_phase(ctx, "output");
    hipLaunchKernelGGL((k_output<GROUPED>), dim3(nb), dim3(TPB), 0, st, n, pos, labk, d_in_classed, d_group,
                       GROUPED ? ext->d_groupstart : nullptr, seedflag, seedpref, cf_in, d_labels, d_is_core, d_is_classed,
                       counters);
  }
  if (!GROUPED && min_pts <= 0 && !all_finite)
    hipLaunchKernelGGL(k_lonely_seeds<GD>, dim3(nb), dim3(TPB), 0, st, d_coords, n, stride, d_in_classed, d_is_classed,
                       counters);
  if (GROUPED) {
    hipLaunchKernelGGL(k_group_stats, dim3(vcp_blocks(G, TPB)), dim3(TPB), 0, st, G, glo, ghi, ext->d_groupstart,
                       seedflag, seedpref, ext->d_group_twice, ext->d_group_nclus, counters + 3, ext->skip_upto);
    if (ext->d_group_evals)
      VCP_HIP(ctx, hipMemcpyAsync(ext->d_group_evals, counters + 3, 8, hipMemcpyDeviceToDevice, st));
  }
  VCP_HIP(ctx, hipGetLastError());
  unsigned long long* hc = reinterpret_cast<unsigned long long*>(ctx->pinned) + 8;
  VCP_HIP(ctx, hipMemcpyAsync(hc, counters, 68 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  VCP_TRY(vcp_phase_finish(ctx));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  const uint32_t K = *reinterpret_cast<uint32_t*>(hc + 2);
  unsigned long long unclassed = (unsigned long long)n;  // nobody classed on entry
  if (d_in_classed) {
    unclassed = 0;
    for (int k = 0; k < 32; k++) unclassed += hc[4 + k];
  }
  if (cf_out) *cf_out = cf_in + (int32_t)K;
  unsigned long long twice_total = 0;
  for (int k = 0; k < 32; k++) twice_total += hc[36 + k];
  if (dist_evals) *dist_evals = GROUPED ? (int64_t)hc[3] : (int64_t)(unclassed + twice_total + K - hc[0]) * n;
  return VCP_OK;
}

template <int GD, int METRIC>
int run_slab_finish(vcp_ctx* ctx, const SlabState& ss, const uint32_t* d_map_rep, const uint32_t* d_map_k,
                    const int32_t* d_tab_gid, uint32_t own_lo, uint32_t own_span, int32_t* d_labels,
                    uint8_t* d_is_classed, int64_t* twice) {
  hipStream_t st = ctx->stream;
  const int64_t n = ss.n;
  const unsigned nb = ss.nb;
  const GridP g = ss.g;
  const size_t ctw = vcp_ct_words(g.ncells);
  const CellTab ct{ctx->b_ctw.as<uint4>(), ctx->b_ctd.as<uint32_t>(),
                   reinterpret_cast<uint32_t*>(ctx->b_ctw.as<char>() + ctw * 16) + 1};
  uint32_t* pos = ctx->b_pos.as<uint32_t>();
  uint32_t* sord = ctx->b_sidx.as<uint32_t>();
  uint8_t* flags = ctx->b_flags.as<uint8_t>();
  uint32_t* parent = ctx->b_parent.as<uint32_t>();
  uint32_t* minord = ctx->b_minord.as<uint32_t>();
  uint32_t* rootk = ctx->b_rootcl.as<uint32_t>();
  uint32_t* clseed = ctx->b_clseed.as<uint32_t>();  // the caller's table of global seeds (copied in by now)
  uint32_t* labk = ctx->b_labk.as<uint32_t>();
  WorkList wlB;
  uint32_t* blkE = ctx->b_wl.as<uint32_t>();
  uint32_t* blkB = blkE + (nb + 2);
  wlB.list = blkB + (nb + 2) + n;
  wlB.scan = blkB;
  wlB.nblk = nb;
  wlB.perblk = list_perblk(n);
  const unsigned nbl = 8u * LCHUNK * wlB.perblk;
  const int rb = (int)vcp_blocks(n, TPB, 1024);
  unsigned long long* counters = reinterpret_cast<unsigned long long*>(ctx->b_misc.as<double>() + (size_t)rb * 8 + 8);
  vcp_phase(ctx, "slab_roots");
  VCP_HIP(ctx, hipMemsetAsync(counters, 0, 68 * sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_slab_rootk, dim3(nb), dim3(TPB), 0, st, parent, flags, minord, ct, d_map_rep, d_map_k,
                     (uint32_t)ss.n_comp, rootk, counters);
  vcp_phase(ctx, "border");
  hipLaunchKernelGGL(k_labk_rest, dim3(nb), dim3(TPB), 0, st, flags, parent, rootk, labk, ct);
  hipLaunchKernelGGL((k_border<GD, METRIC, false>), dim3(nbl), dim3(TPB), 0, st, ss.xs, g, ss.thr,
                     ct, nullptr, flags, parent, sord, rootk, clseed, labk, counters, nullptr, wlB, own_lo, own_span,
                     ctx->b_sorted32.as<float>(), ss.sc);
  vcp_phase(ctx, "output");
  hipLaunchKernelGGL(k_slab_output, dim3(nb), dim3(TPB), 0, st, n, pos, labk, d_tab_gid, d_labels, d_is_classed);
  VCP_HIP(ctx, hipGetLastError());
  unsigned long long* hc = reinterpret_cast<unsigned long long*>(ctx->pinned) + 8;
  VCP_HIP(ctx, hipMemcpyAsync(hc, counters, 68 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  VCP_TRY(vcp_phase_finish(ctx));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (hc[0] != 0)
    return vcp_fail(ctx, VCP_ERR_ARG, "%llu local components are missing from the map", hc[0]);
  if (twice) {
    *twice = 0;
    for (int k = 0; k < 32; k++) *twice += (int64_t)hc[36 + k];
  }
  return VCP_OK;
}

int run_degenerate(vcp_ctx* ctx, int64_t n, int min_pts, int32_t cf_in, const uint8_t* d_in_classed,
                   int32_t* d_labels, uint8_t* d_is_core, uint8_t* d_is_classed, int32_t* cf_out,
                   int64_t* dist_evals) {
  hipStream_t st = ctx->stream;
  const unsigned nb = vcp_blocks(n, TPB);
  vcp_phase(ctx, "degenerate");
  VCP_TRY(vcp_ensure(ctx, ctx->b_seedflag, (size_t)(n + 2) * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_misc, 64));
  uint32_t* f = ctx->b_seedflag.as<uint32_t>();
  uint32_t* d_total = ctx->b_misc.as<uint32_t>();
  hipLaunchKernelGGL(k_unclassed_flag, dim3(nb), dim3(TPB), 0, st, d_in_classed, f, n);
  VCP_TRY(vcp_exclusive_scan_u32(ctx, f, f, n, d_total));
  const int all_core = 0 >= min_pts;  // tmpList.Count (0) >= minPts, BaseClass/DBImproved.cs:105
  hipLaunchKernelGGL(k_degenerate, dim3(nb), dim3(TPB), 0, st, d_in_classed, f, n, cf_in, all_core,
                     d_in_classed == nullptr, d_labels, d_is_core, d_is_classed);
  VCP_HIP(ctx, hipGetLastError());
  uint32_t* hu = reinterpret_cast<uint32_t*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(hu, d_total, 4, hipMemcpyDeviceToHost, st));
  VCP_TRY(vcp_phase_finish(ctx));
  VCP_HIP(ctx, hipStreamSynchronize(st));
  if (cf_out) *cf_out = cf_in + (all_core ? (int32_t)hu[0] : 0);
  if (dist_evals) *dist_evals = (int64_t)hu[0] * n;
  return VCP_OK;
}

}  // namespace

int vcp_dbscan_engine(vcp_ctx* ctx, const double* d_coords, int64_t n, int stride, int metric, double eps,
                      int min_pts, int32_t cf_in, const uint8_t* d_in_classed, int32_t* d_labels,
                      uint8_t* d_is_core, uint8_t* d_is_classed, int32_t* cf_out, int64_t* dist_evals,
                      const DbscanExt* ext) {
  if (!ctx) return VCP_ERR_ARG;
  if (n < 0) return vcp_fail(ctx, VCP_ERR_ARG, "n < 0");
  if (stride != 2 && stride != 3) return vcp_fail(ctx, VCP_ERR_ARG, "dim must be 2 or 3");
  if (metric < 0 || metric > 3) return vcp_fail(ctx, VCP_ERR_ARG, "unknown metric %d", metric);
  if (metric == VCP_SIGNED_SUM_2D) {  // the dead v1.0 class DB (BaseClass/DB.cs): csrc/dbdead.hip
    if (ext) return vcp_fail(ctx, VCP_ERR_ARG, "DB has no grouped / staged form");
    if (n >= 0x7FFFFFF0LL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "n beyond 32-bit indexing");
    if (n > 0 && (!d_coords || !d_labels)) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
    VCP_TRY(vcp_bind(ctx));
    vcp_phase_reset(ctx);
    if (ctx->slab) ctx->slab->valid = false;
    if (n == 0) {
      if (cf_out) *cf_out = cf_in;
      if (dist_evals) *dist_evals = 0;
      ctx->last_timing.clear();
      return VCP_OK;
    }
    return vcp_db_engine(ctx, d_coords, n, stride, eps, min_pts, cf_in, nullptr, d_in_classed, d_labels, d_is_core,
                         d_is_classed, cf_out, dist_evals);
  }
  if (metric == VCP_L2_3D && stride != 3) return vcp_fail(ctx, VCP_ERR_ARG, "VCP_L2_3D needs dim 3");
  if (n >= 0x3FFFFFF0LL) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "n beyond 30-bit indexing");
  if (n > 0 && (!d_coords || !d_labels)) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  const bool grouped = ext && ext->d_group;
  if (grouped && (metric != VCP_L1_2D || !ext->d_groupstart || !ext->d_group_twice || d_in_classed))
    return vcp_fail(ctx, VCP_ERR_ARG, "grouped DBSCAN needs groupstart, group_twice and the L1 metric");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  if (ctx->slab) ctx->slab->valid = false;  // the workspace is about to be overwritten
  if (n == 0) {
    if (cf_out) *cf_out = cf_in;
    if (dist_evals) *dist_evals = 0;
    ctx->last_timing.clear();
    return VCP_OK;
  }
  if (!(eps >= 0.0)) {
    if (grouped) return vcp_fail(ctx, VCP_ERR_ARG, "grouped DBSCAN needs eps >= 0");
    return run_degenerate(ctx, n, min_pts, cf_in, d_in_classed, d_labels, d_is_core, d_is_classed, cf_out, dist_evals);
  }
#define VCP_RUN(D, M, GR)                                                                                        \
  return run_dbscan<D, M, GR>(ctx, d_coords, n, stride, eps, min_pts, cf_in, d_in_classed, d_labels, d_is_core, \
                              d_is_classed, cf_out, dist_evals, ext)
  if (grouped) VCP_RUN(2, VCP_L1_2D, true);
  if (metric == VCP_L1_2D) VCP_RUN(2, VCP_L1_2D, false);
  if (metric == VCP_L2_2D) VCP_RUN(2, VCP_L2_2D, false);
  VCP_RUN(3, VCP_L2_3D, false);
#undef VCP_RUN
}

extern "C" {

void vcp_slab_state_free(vcp_ctx* ctx) {
  delete ctx->slab;
  ctx->slab = nullptr;
}

int vcp_slab_begin(vcp_ctx* ctx, const double* d_coords, int64_t n, int dim, int metric, double eps, int min_pts,
                   const uint8_t* d_noexpand, const uint32_t* d_ord, uint32_t* d_rep, uint8_t* d_is_core,
                   int64_t* n_comp) {
  if (!ctx) return VCP_ERR_ARG;
  if (n <= 0) return vcp_fail(ctx, VCP_ERR_ARG, "vcp_slab_begin needs n > 0");
  if (!d_ord || !d_rep || !n_comp) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  if (!(eps >= 0.0) || std::isinf(eps)) return vcp_fail(ctx, VCP_ERR_ARG, "staged DBSCAN needs a finite eps >= 0");
  DbscanExt ext;
  ext.d_ord = d_ord;
  ext.slab = true;
  ext.d_slab_rep = d_rep;
  int32_t nc = 0;
  // the labels argument is not written by a staged call; d_rep stands in for the null check
  int rc = vcp_dbscan_engine(ctx, d_coords, n, dim, metric, eps, min_pts, 0, d_noexpand,
                             reinterpret_cast<int32_t*>(d_rep), d_is_core, nullptr, &nc, nullptr, &ext);
  if (rc != VCP_OK) return rc;
  if (!ctx->slab || !ctx->slab->valid) return vcp_fail(ctx, VCP_ERR_ARG, "staged call did not reach the grid path");
  *n_comp = ctx->slab->n_comp;
  return VCP_OK;
}

int vcp_slab_comps(vcp_ctx* ctx, uint32_t* comp_rep) {
  if (!ctx) return VCP_ERR_ARG;
  if (!ctx->slab || !ctx->slab->valid) return vcp_fail(ctx, VCP_ERR_ARG, "no vcp_slab_begin state in this context");
  if (ctx->slab->n_comp == 0) return VCP_OK;
  if (!comp_rep) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(vcp_bind(ctx));
  VCP_HIP(ctx, hipMemcpyAsync(comp_rep, ctx->b_clseed.p, (size_t)ctx->slab->n_comp * 4, hipMemcpyDeviceToHost,
                              ctx->stream));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VCP_OK;
}

int vcp_slab_finish(vcp_ctx* ctx, const uint32_t* map_rep, const uint32_t* map_k, int64_t n_tab,
                    const int32_t* tab_gid, const uint32_t* tab_seed, uint32_t own_lo, uint32_t own_count,
                    int32_t* d_labels, uint8_t* d_is_classed, int64_t* twice) {
  if (!ctx) return VCP_ERR_ARG;
  if (!ctx->slab || !ctx->slab->valid) return vcp_fail(ctx, VCP_ERR_ARG, "no vcp_slab_begin state in this context");
  SlabState& ss = *ctx->slab;
  if (!d_labels) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  if (n_tab < 0 || n_tab > ss.n_comp) return vcp_fail(ctx, VCP_ERR_ARG, "cluster table larger than the component list");
  if (ss.n_comp > 0 && (!map_rep || !map_k || !tab_gid || !tab_seed)) return vcp_fail(ctx, VCP_ERR_ARG, "null map");
  for (int64_t k = 0; k < ss.n_comp; k++) {
    if (k > 0 && map_rep[k] <= map_rep[k - 1]) return vcp_fail(ctx, VCP_ERR_ARG, "map_rep must be strictly ascending");
    if ((int64_t)map_k[k] >= n_tab) return vcp_fail(ctx, VCP_ERR_ARG, "map_k out of range");
  }
  for (int64_t k = 1; k < n_tab; k++)
    if (tab_gid[k] <= tab_gid[k - 1]) return vcp_fail(ctx, VCP_ERR_ARG, "tab_gid must be strictly ascending");
  VCP_TRY(vcp_bind(ctx));
  vcp_phase_reset(ctx);
  hipStream_t st = ctx->stream;
  const size_t L = (size_t)(ss.n_comp > 0 ? ss.n_comp : 1);
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux0, L * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux1, L * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_aux2, L * 4));
  if (ss.n_comp > 0) {
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_aux0.p, map_rep, (size_t)ss.n_comp * 4, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_aux1.p, map_k, (size_t)ss.n_comp * 4, hipMemcpyHostToDevice, st));
  }
  if (n_tab > 0) {
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_aux2.p, tab_gid, (size_t)n_tab * 4, hipMemcpyHostToDevice, st));
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_clseed.p, tab_seed, (size_t)n_tab * 4, hipMemcpyHostToDevice, st));
  }
  const uint32_t* mr = ctx->b_aux0.as<uint32_t>();
  const uint32_t* mk = ctx->b_aux1.as<uint32_t>();
  const int32_t* tg = ctx->b_aux2.as<int32_t>();
  int rc;
  if (ss.metric == VCP_L1_2D)
    rc = run_slab_finish<2, VCP_L1_2D>(ctx, ss, mr, mk, tg, own_lo, own_count, d_labels, d_is_classed, twice);
  else if (ss.metric == VCP_L2_2D)
    rc = run_slab_finish<2, VCP_L2_2D>(ctx, ss, mr, mk, tg, own_lo, own_count, d_labels, d_is_classed, twice);
  else
    rc = run_slab_finish<3, VCP_L2_3D>(ctx, ss, mr, mk, tg, own_lo, own_count, d_labels, d_is_classed, twice);
  ss.valid = false;
  return rc;
}

int vcp_dbscan_dev(vcp_ctx* ctx, const double* d_coords, int64_t n, int dim, int metric, double eps,
                   int min_pts, int32_t cf_in, const uint8_t* d_in_classed, int32_t* d_labels,
                   uint8_t* d_is_core, uint8_t* d_is_classed, int32_t* cf_out, int64_t* dist_evals) {
  return vcp_dbscan_engine(ctx, d_coords, n, dim, metric, eps, min_pts, cf_in, d_in_classed, d_labels, d_is_core,
                           d_is_classed, cf_out, dist_evals, nullptr);
}

int vcp_dbscan(vcp_ctx* ctx, const double* coords, int64_t n, int dim, int metric, double eps, int min_pts,
               int32_t cf_in, const uint8_t* in_mask, const uint8_t* in_classed, int32_t* labels,
               uint8_t* is_core, uint8_t* is_classed, int32_t* cf_out, int64_t* dist_evals) {
  if (!ctx) return VCP_ERR_ARG;
  if (in_mask && metric != VCP_SIGNED_SUM_2D)
    return vcp_fail(ctx, VCP_ERR_UNSUPPORTED,
                    "ifShown masks belong to the DB class (BaseClass/DB.cs:40): use metric VCP_SIGNED_SUM_2D");
  if (n < 0) return vcp_fail(ctx, VCP_ERR_ARG, "n < 0");
  if (dim != 2 && dim != 3) return vcp_fail(ctx, VCP_ERR_ARG, "dim must be 2 or 3");
  if (n > 0 && (!coords || !labels)) return vcp_fail(ctx, VCP_ERR_ARG, "null buffer");
  VCP_TRY(vcp_bind(ctx));
  hipStream_t st = ctx->stream;
  size_t nn = (size_t)(n > 0 ? n : 1);
  VCP_TRY(vcp_ensure(ctx, ctx->b_in0, nn * dim * 8));
  VCP_TRY(vcp_ensure(ctx, ctx->b_in1, nn));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out0, nn * 4));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out1, nn));
  VCP_TRY(vcp_ensure(ctx, ctx->b_out2, nn));
  if (n > 0) {
    VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in0.p, coords, (size_t)n * dim * 8, hipMemcpyHostToDevice, st));
    if (in_classed) {
      VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in1.p, in_classed, (size_t)n, hipMemcpyHostToDevice, st));
      VCP_HIP(ctx, hipMemcpyAsync(ctx->b_out0.p, labels, (size_t)n * 4, hipMemcpyHostToDevice, st));
    }
  }
  int rc;
  if (metric == VCP_SIGNED_SUM_2D && n > 0) {
    if (in_mask) {
      VCP_TRY(vcp_ensure(ctx, ctx->b_in2, nn));
      VCP_HIP(ctx, hipMemcpyAsync(ctx->b_in2.p, in_mask, (size_t)n, hipMemcpyHostToDevice, st));
    }
    vcp_phase_reset(ctx);
    if (ctx->slab) ctx->slab->valid = false;
    rc = vcp_db_engine(ctx, ctx->b_in0.as<double>(), n, dim, eps, min_pts, cf_in,
                       in_mask ? ctx->b_in2.as<uint8_t>() : nullptr, in_classed ? ctx->b_in1.as<uint8_t>() : nullptr,
                       ctx->b_out0.as<int32_t>(), ctx->b_out1.as<uint8_t>(), ctx->b_out2.as<uint8_t>(), cf_out, dist_evals);
  } else {
    rc = vcp_dbscan_dev(ctx, ctx->b_in0.as<double>(), n, dim, metric, eps, min_pts, cf_in,
                        in_classed ? ctx->b_in1.as<uint8_t>() : nullptr, ctx->b_out0.as<int32_t>(),
                        ctx->b_out1.as<uint8_t>(), ctx->b_out2.as<uint8_t>(), cf_out, dist_evals);
  }
  if (rc != VCP_OK) return rc;
  if (n > 0) {
    VCP_HIP(ctx, hipMemcpyAsync(labels, ctx->b_out0.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (is_core) VCP_HIP(ctx, hipMemcpyAsync(is_core, ctx->b_out1.p, (size_t)n, hipMemcpyDeviceToHost, st));
    if (is_classed) VCP_HIP(ctx, hipMemcpyAsync(is_classed, ctx->b_out2.p, (size_t)n, hipMemcpyDeviceToHost, st));
    VCP_HIP(ctx, hipStreamSynchronize(st));
  }
  return VCP_OK;
}

}  // extern "C"
