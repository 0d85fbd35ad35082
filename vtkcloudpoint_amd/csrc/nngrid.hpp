// nngrid.hpp -- exact nearest neighbour against a STATIC 3-D point set through a uniform grid (MI355X, gfx950).
//
// Used where the reference scans a whole list per query: ICP.FindClosestPointSet (BaseClass/ICP.cs:224-250) once the
// model has thousands of points (MainForm.ICP's real use: cluster centroids against the truth list, FrmMain.cs:841-907)
// and RecorrectMatchingPtsByDistance (FrmMain.cs:3588-3618).  The set is binned once per call (it does not move); a
// query searches the cells within Chebyshev distance r = 1, 2, 4, ... (while the block is small against the grid) of its own cell and stops as soon as the best
// distance found is provably smaller than anything outside the searched block (every unexplored point lies at least
// r * h away), falling back to a scan of the whole set.  Candidates are evaluated with the reference's own binary64
// expression and compared as (value, original index) pairs, so the result is the index the sequential strict-`<`
// scan returns -- lowest index among equal values, also when the values are compared after a sqrt.
#pragma once
#include "vcp_ctx.hpp"

struct NNGrid {
  double mn[3];
  double h, inv_h;
  int D[3];
  uint32_t ncells;
  int n;
  const uint32_t* cellstart;  // [ncells + 1]
  const double4* rec;         // [n] in cell order: (x, y, z, original index in the low word of w)
};

// Bins d_pts [n*3] (finite coordinates) on the context's stream; the grid's arrays live in ctx->b_nn_* until the next
// build.  Returns VCP_ERR_UNSUPPORTED when the set has non-finite coordinates (callers keep their full scan).
int vcp_nngrid_build(vcp_ctx* ctx, const double* d_pts, int64_t n, NNGrid* out);

#if defined(__HIPCC__)
namespace nng {

__device__ __forceinline__ int cell1(double x, double mn, double inv_h, int D) {
  const double u = (x - mn) * inv_h;
  if (u >= 0.0 && u < (double)D) return (int)u;
  if (u >= (double)D) return D - 1;
  return 0;  // below the minimum or NaN
}

// value of a candidate: squared distance summed left to right like the C#, optionally through the correctly rounded
// sqrt (getDisP, FrmMain.cs:829-835)
template <bool USE_SQRT>
__device__ __forceinline__ double value(const double* p, const double4& m) {
  const double e0 = p[0] - m.x, e1 = p[1] - m.y, e2 = p[2] - m.z;
  const double dd = e0 * e0 + e1 * e1 + e2 * e2;
  return USE_SQRT ? sqrt(dd) : dd;
}

template <bool USE_SQRT>
__device__ __forceinline__ void scan_range(const NNGrid& g, const double* p, uint32_t s, uint32_t e, double& best,
                                           int& order) {
  for (uint32_t j = s; j < e; j += 2) {  // two candidates in flight
    const double4 a = g.rec[j];
    const double4 b = g.rec[min(j + 1, e - 1)];
    const double va = value<USE_SQRT>(p, a);
    const int ia = __double2loint(a.w);
    if (va < best || (va == best && ia < order)) {
      best = va;
      order = ia;
    }
    if (j + 1 < e) {
      const double vb = value<USE_SQRT>(p, b);
      const int ib = __double2loint(b.w);
      if (vb < best || (vb == best && ib < order)) {
        best = vb;
        order = ib;
      }
    }
  }
}

// order = index the sequential scan `min = f(0); for i: if (f(i) < min) ...` ends with; best = its value.
// A query is worked by a GROUP of NNG lanes (consecutive lanes, all holding the same p): the rows of the searched block
// are dealt round-robin to the lanes and the (value, index) pairs reduced with shuffles after every block, so a
// query's chain of dependent loads (row bounds, then candidates) is an eighth as long -- the searches are latency
// bound: 27 k queries are only 430 waves with one lane per query (27 k x 27 k matching: 297 -> see DESIGN.md).
constexpr int NNG = 8;

template <bool USE_SQRT>
__device__ __forceinline__ void query(const NNGrid& g, const double* p, int sub, int& order, double& best) {
  order = 0;
  best = INFINITY;  // (INFINITY, 0): what the C# keeps when no candidate compares smaller (non-finite query)
  const int cx = cell1(p[0], g.mn[0], g.inv_h, g.D[0]);
  const int cy = cell1(p[1], g.mn[1], g.inv_h, g.D[1]);
  const int cz = cell1(p[2], g.mn[2], g.inv_h, g.D[2]);
  const int maxd = max(g.D[0], max(g.D[1], g.D[2]));
  auto reduce = [&]() {
#pragma unroll
    for (int d = 1; d < NNG; d <<= 1) {
      const double ob = __shfl_xor(best, d, 64);
      const int oo = __shfl_xor(order, d, 64);
      if (ob < best || (ob == best && oo < order)) {
        best = ob;
        order = oo;
      }
    }
  };
  for (int r = 1; r <= 64; r <<= 1) {
    const int x0 = max(cx - r, 0), x1 = min(cx + r, g.D[0] - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, g.D[1] - 1);
    const int z0 = max(cz - r, 0), z1 = min(cz + r, g.D[2] - 1);
    // a block of half the grid or more: the linear scan of the whole set below is the cheaper way to finish (the rings
    // keep doubling while they are small against the set: a query with no neighbour nearby -- the first rounds of an
    // ICP that starts far off -- used to fall from r = 8 straight to the whole set)
    if (r > 1 && 2ll * (x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) > (long long)g.ncells) break;
    int turn = 0;
    for (int z = z0; z <= z1; z++)
      for (int y = y0; y <= y1; y++, turn++) {
        if ((turn & (NNG - 1)) != sub) continue;
        const uint32_t row = ((uint32_t)z * (uint32_t)g.D[1] + (uint32_t)y) * (uint32_t)g.D[0];
        scan_range<USE_SQRT>(g, p, g.cellstart[row + x0], g.cellstart[row + x1 + 1], best, order);
      }
    reduce();
    // every point outside the block is at least r*h away (the margin covers the roundings of the cell arithmetic)
    const double lb = (double)r * g.h * (1.0 - 9.5367431640625e-07);
    const double lbv = USE_SQRT ? lb : lb * lb;
    if (best < lbv * (1.0 - 9.5367431640625e-07)) return;
    if (r >= maxd) return;  // the block covered the whole grid
  }
  // far from everything: the whole set, dealt in slices
  const uint32_t n = (uint32_t)g.n, per = (n + NNG - 1) / NNG;
  const uint32_t lo = min(n, per * (uint32_t)sub), hi = min(n, lo + per);
  scan_range<USE_SQRT>(g, p, lo, hi, best, order);
  reduce();
}

}  // namespace nng
#endif
