// grid_common.hpp -- what the grid build (gridbuild.hip) and the search kernels (dbscan.hip) share: the grid
// geometry, cell arithmetic and point loads.  Everything is internal linkage (each translation unit its own copy).
#pragma once
#include "vcp_ctx.hpp"

namespace vcpg {

constexpr int TPB = 256;
constexpr uint8_t F_CORE = 1, F_CLASSED = 2, F_EXPAND = 4, F_BCAND = 8;
constexpr uint32_t NONE = 0xFFFFFFFFu;

struct GridP {
  double mn[3];
  double inv_h;     // cells per unit of the SCALED relative coordinate
  double scale;     // power of two applied to (coordinate - origin) before it is rounded to binary32: 1 unless the cloud's
                    // extent is outside the range where binary32 keeps its relative precision (run_dbscan)
  int D[3];         // cells per axis
  uint32_t ncells;
};

// linear id of cell (cx,cy,cz), x fastest: the 3 cells of a neighbour row are one contiguous position range.
// (A tile-major order -- tiles of 4..16 cells per axis -- was measured 5-25 % SLOWER on MI355X for these
// latency-bound search loops, with or without the XCD-aware block map, and was dropped.)
template <int GD>
__device__ __forceinline__ uint32_t cell_id(const GridP& g, int cx, int cy, int cz) {
  uint32_t id = (uint32_t)cy * (uint32_t)g.D[0] + (uint32_t)cx;
  if (GD == 3) id += (uint32_t)cz * (uint32_t)g.D[0] * (uint32_t)g.D[1];
  return id;
}

// The grid bins on the BINARY32 value of the coordinate relative to the grid origin -- the same number the screening
// copies (sorted32) hold -- so a kernel that has only the 8/16-byte screening copy of a point knows its cell, and the
// partition build can carry binary32 records.  rel32 is monotone in x, hence so is the cell index; the cell width
// includes the rounding of two such values (run(): cellw), so two points within eps of each other on an axis still
// land in the same or in adjacent cells.
// (the factor is a power of two: exact, so the scaled value rounds exactly like the unscaled one would in a wider format)
__device__ __forceinline__ float rel32(double x, double mn, double scale) { return (float)((x - mn) * scale); }

__device__ __forceinline__ int cell_coord32(float f, double inv_h, int D) {
  double u = (double)f * inv_h;
  if (u >= 0.0 && u < (double)D) return (int)u;
  if (u >= (double)D) return D - 1;
  return 0;  // below the minimum or NaN
}

__device__ __forceinline__ int cell_coord(double x, double mn, double scale, double inv_h, int D) {
  return cell_coord32(rel32(x, mn, scale), inv_h, D);
}

template <int GD>
__device__ __forceinline__ void load_pt(const double* __restrict__ c, int64_t i, double* q) {
  if (GD == 2) {
    double2 v = *reinterpret_cast<const double2*>(c + 2 * i);
    q[0] = v.x;
    q[1] = v.y;
  } else {
    q[0] = c[3 * i];
    q[1] = c[3 * i + 1];
    q[2] = c[3 * i + 2];
  }
}

template <int GD>
__device__ __forceinline__ void store_pt(double* __restrict__ c, int64_t i, const double* q) {
  if (GD == 2) {
    *reinterpret_cast<double2*>(c + 2 * i) = make_double2(q[0], q[1]);
  } else {
    c[3 * i] = q[0];
    c[3 * i + 1] = q[1];
    c[3 * i + 2] = q[2];
  }
}

// caller-order input: `stride` doubles per point, the metric reads the first GD of them
template <int GD>
__device__ __forceinline__ void load_in(const double* __restrict__ c, int64_t i, int stride, double* q) {
  if (GD == 2 && stride == 2) {
    double2 v = *reinterpret_cast<const double2*>(c + 2 * i);
    q[0] = v.x;
    q[1] = v.y;
  } else {
#pragma unroll
    for (int a = 0; a < GD; a++) q[a] = c[i * stride + a];
  }
}

template <int GD>
__device__ __forceinline__ uint32_t cell_of(const double* q, const GridP& g, int* cc) {
  cc[0] = cell_coord(q[0], g.mn[0], g.scale, g.inv_h, g.D[0]);
  cc[1] = cell_coord(q[1], g.mn[1], g.scale, g.inv_h, g.D[1]);
  cc[2] = 0;
  if (GD == 3) cc[2] = cell_coord(q[2], g.mn[2], g.scale, g.inv_h, g.D[2]);
  return cell_id<GD>(g, cc[0], cc[1], cc[2]);
}

template <int GD>
__device__ __forceinline__ uint32_t cell_of32(const float* qf, const GridP& g, int* cc) {
  cc[0] = cell_coord32(qf[0], g.inv_h, g.D[0]);
  cc[1] = cell_coord32(qf[1], g.inv_h, g.D[1]);
  cc[2] = 0;
  if (GD == 3) cc[2] = cell_coord32(qf[2], g.inv_h, g.D[2]);
  return cell_id<GD>(g, cc[0], cc[1], cc[2]);
}

// ---- the cell table ---------------------------------------------------------------------------------------------
// start(c) = cell-ordered position of the first point whose cell id is >= c, for 0 <= c <= ncells.  A dense array of
// starts costs 4 bytes per CELL -- on the sparse grids of this path (10-30 cells per point: 40-120 bytes per point) it
// was the largest stream of the build and of the region query.  The table is kept per WORD of 32 consecutive cells
// instead, 16 bytes each (0.5 byte per cell):
//   lo, hi   two bit planes: the cell's population, saturated at 3 (lo & hi = "3 or more")
//   wpos     start of the word's first cell
//   slot     where a word with a populous cell keeps its 32 starts in full (`dense`, 128 bytes), else unused
// start(c) = wpos + popcount(lo below c) + 2 popcount(hi below c) while no cell below c in the word is populous -- ONE
// 16-byte load and no dependent one; otherwise dense[32 slot + (c & 31)].  Populous words are the inside of blobs: few
// (every one holds >= 3 points), hot in cache, and 4 bytes per cell is what they would have cost anyway.
struct CellTab {
  const uint4* words;      // [(ncells + 32) / 32]
  const uint32_t* dense;   // [32 * number of populous words]
  const uint32_t* nin;     // number of points in the grid (= start(ncells))
};

__device__ __forceinline__ uint32_t ct_start(const CellTab& t, uint32_t cell) {
  const uint4 w = t.words[cell >> 5];
  const uint32_t i = cell & 31u, below = (1u << i) - 1u;
  if ((w.x & w.y & below) == 0u) return w.z + (uint32_t)__popc(w.x & below) + 2u * (uint32_t)__popc(w.y & below);
  return t.dense[(size_t)w.w * 32u + i];
}
// two bounds of one word-local range (rows: x0 .. x1 + 1 are at most 3 cells apart and usually share a word)
__device__ __forceinline__ uint32_t ct_from_word(const CellTab& t, const uint4 w, uint32_t cell) {
  const uint32_t i = cell & 31u, below = (1u << i) - 1u;
  if ((w.x & w.y & below) == 0u) return w.z + (uint32_t)__popc(w.x & below) + 2u * (uint32_t)__popc(w.y & below);
  return t.dense[(size_t)w.w * 32u + i];
}
__device__ __forceinline__ void ct_range(const CellTab& t, uint32_t c0, uint32_t c1, uint32_t& s, uint32_t& e) {
  const uint4 w0 = t.words[c0 >> 5];
  s = ct_from_word(t, w0, c0);
  e = (c1 >> 5) == (c0 >> 5) ? ct_from_word(t, w0, c1) : ct_start(t, c1);  // 29 rows in 32 share the word: one load
}

// ---- binary32 screening of the distance predicate ----------------------------------------------------------
// The search kernels decide `d(p, j) <= eps` on binary32 copies of the coordinates, taken relative to the grid origin,
// wherever that decision is provably the binary64 one: value <= lo -> inside, value > hi -> outside, anything else
// (also NaN) is re-tested with the reference's exact binary64 expression.  lo / hi come from a rounding analysis on
// the host (screen_bounds in dbscan.hip); half the bytes per candidate and a quarter of the FP64 work in 3-D.
struct Screen {
  float lo, hi;
};

// coordinates of cell-ordered point p relative to the grid origin, in binary32 (2-D: float2, 3-D: float4 with w unused)
template <int GD>
__device__ __forceinline__ void load_pt32(const float* __restrict__ c, int64_t i, float* q) {
  if (GD == 2) {
    const float2 v = *reinterpret_cast<const float2*>(c + 2 * i);
    q[0] = v.x;
    q[1] = v.y;
  } else {
    const float4 v = *reinterpret_cast<const float4*>(c + 4 * i);
    q[0] = v.x;
    q[1] = v.y;
    q[2] = v.z;
  }
}
template <int GD>
__device__ __forceinline__ void store_pt32(float* __restrict__ c, int64_t i, const double* q, const GridP& g) {
  if (GD == 2) {
    *reinterpret_cast<float2*>(c + 2 * i) = make_float2(rel32(q[0], g.mn[0], g.scale), rel32(q[1], g.mn[1], g.scale));
  } else {
    *reinterpret_cast<float4*>(c + 4 * i) =
        make_float4(rel32(q[0], g.mn[0], g.scale), rel32(q[1], g.mn[1], g.scale), rel32(q[2], g.mn[2], g.scale), 0.0f);
  }
}

// Where the binary64 coordinates of cell-ordered position p are: a cell-ordered copy (idx == NULL, the sort-based
// build and the staged multi-GPU calls), or the CALLER's array through the point's index (the partition build carries
// binary32 records only).  Read by the exact re-test of a pair the binary32 screen cannot decide -- a handful per
// million candidates on real-valued clouds -- so the gather costs nothing there; on inputs where many pairs sit exactly
// on the threshold (lattices with eps on the lattice) it is the price of the lighter build.
struct ExactSrc {
  const double* base;
  const uint32_t* idx;
  int stride;
};
template <int GD>
__device__ __forceinline__ void load_exact(const ExactSrc& xs, uint32_t p, double* q) {
  const int64_t i = xs.idx ? (int64_t)xs.idx[p] : (int64_t)p;
  load_in<GD>(xs.base, i, xs.stride, q);
}

}  // namespace vcpg

// ---- grid build by two-level partition (gridbuild.hip) ------------------------------------------------------
// Caller order -> cell order without a (key, index) sort and without a per-point random gather: the coordinates travel
// with the index through one coarse partition (buckets = contiguous ranges of cell ids) and one per-bucket counting
// sort in LDS that also emits the bucket's slice of the cell table.
struct GridBuildArgs {
  const double* d_coords = nullptr;  // [n * stride] caller order
  int64_t n = 0;
  int stride = 2, gd = 2;
  vcpg::GridP g;
  const int32_t* d_group = nullptr;  // grouped calls: points with group outside [glo, ghi) are left out
  int glo = 0, ghi = 0;
  const uint32_t* d_ord = nullptr;          // list position of point i (NULL = i)
  const uint8_t* d_in_classed = nullptr;    // NULL = flags are zero-filled by the caller
  // outputs (cell order unless noted)
  uint4* ctwords = nullptr;       // [(ncells + 32) / 32] the cell table (CellTab)
  uint32_t* ctdense = nullptr;    // [32 * min(words, n / 3 + 1)] full starts of the populous words
  uint32_t* ctcount = nullptr;    // [2]: [0] populous words handed out so far (zero on entry), [1] <- points in the grid
  float* sorted32 = nullptr;      // [nin * (gd == 2 ? 2 : 4)] binary32 coordinates relative to g.mn (binning, screening)
  uint32_t* sord = nullptr;       // [nin] list position (d_ord of the point's index)
  uint32_t* sidx = nullptr;       // [nin] the point's index in d_coords; NULL = not wanted (it is sord without d_ord)
  int32_t* sgroup = nullptr;      // [nin], grouped only
  uint8_t* flags = nullptr;       // [nin], written only with d_in_classed
  uint32_t* pos = nullptr;        // [n] caller order, NONE for left-out points; NULL = not wanted
};
// words of the cell table of a grid of ncells cells, and populous words a cloud of n points can have at most
__host__ __device__ static inline size_t vcp_ct_words(uint32_t ncells) { return ((size_t)ncells + 32) / 32; }
static inline size_t vcp_ct_dense_cap(int64_t n, uint32_t ncells) {
  const size_t w = vcp_ct_words(ncells), byn = (size_t)n / 3 + 1;
  return w < byn ? w : byn;
}
int vcp_grid_build_partition(vcp_ctx* ctx, const GridBuildArgs& a);

// Cell order -> caller order the same way (no per-point random gather): (list position, label word) pairs are
// partitioned by windows of 2^13 list positions, then every window is assembled in LDS and stored as full lines.
struct GridOutputArgs {
  int64_t n = 0;
  const uint32_t* sord = nullptr;  // [n] list position of cell-order position p (a permutation of [0, n))
  const uint32_t* labk = nullptr;  // [n] (1 + seed rank) << 2 | core | classed << 1
  bool have_in_classed = false;
  int32_t cf_in = 0;
  int32_t* labels = nullptr;
  uint8_t* is_core = nullptr;
  uint8_t* is_classed = nullptr;
  unsigned long long* counters = nullptr;  // [4..36): points not classed on entry (only with have_in_classed)
};
int vcp_grid_output_partition(vcp_ctx* ctx, const GridOutputArgs& a);
