/*
 * vcp_oracle.cpp -- CPU ORACLE (test infrastructure only; see vcp_oracle.h).
 *
 * Single thread, strict binary64, compiled with -ffp-contract=off so that every
 * a*b+c below is two roundings exactly as the C# (SSE2) would do.
 * Citations: file:line into /root/reference/vtkPointCloud/ (BC = BaseClass).
 */
#include "vcp_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------
// distance forms
// ---------------------------------------------------------------------------------
struct Dist {
  const double* c;
  int dim;
  int metric;
  int64_t evals = 0;
  // getDisP(p1 = a, p2 = b): BC/DBImproved.cs:14-25 (L1, live) with the commented-out
  // Euclidean alternatives of :20 and :24; BC/DB.cs:14-25 for the signed sum.
  inline double operator()(int64_t a, int64_t b) {
    const double* p1 = c + a * dim;
    const double* p2 = c + b * dim;
    double dx = p1[0] - p2[0];
    double dy = p1[1] - p2[1];
    ++evals;  // iritatorNum++
    switch (metric) {
      case ORC_L1_2D:
        return std::fabs(dx) + std::fabs(dy);
      case ORC_L2_2D:
        return std::sqrt(dx * dx + dy * dy);
      case ORC_L2_3D: {
        double dz = p1[2] - p2[2];
        return std::sqrt(dx * dx + dy * dy + dz * dz);
      }
      default:  // ORC_SIGNED_SUM_2D
        return dx + dy;
    }
  }
};

bool metric_ok(int metric, int dim) {
  if (metric < 0 || metric > 3) return false;
  if (metric == ORC_L2_3D) return dim >= 3;
  return dim >= 2;
}

// ---------------------------------------------------------------------------------
// literal DBImproved
// ---------------------------------------------------------------------------------
struct Literal {
  Dist d;
  int64_t n;
  double e;
  int minPts;
  uint8_t* classed;
  int32_t* lab;
  uint8_t* key;
  bool dedupe;
  uint64_t next_box = 1;  // identity of each boxed int (ArrayList stores objects)

  // BC/DBImproved.cs:33-54
  void isKeyPoint(int64_t p, std::vector<int64_t>& tmp, std::vector<uint64_t>& box) {
    int count = 0;
    tmp.clear();
    box.clear();
    for (int64_t i = 0; i < n; i++) {
      if (d(p, i) <= e) {
        ++count;
        tmp.push_back(i);
        box.push_back(next_box++);
      }
    }
    if (count >= minPts) key[p] = 1;
  }

  // BC/DBImproved.cs:56-90
  void expandCluster(int64_t p, std::vector<int64_t>& nei, std::vector<uint64_t>& neibox, int32_t c) {
    lab[p] = c;
    std::vector<int64_t> tmp;
    std::vector<uint64_t> tbox;
    for (size_t i = 0; i < nei.size(); i++) {
      int64_t dpp = nei[i];
      if (!classed[dpp]) {
        classed[dpp] = 1;
        isKeyPoint(dpp, tmp, tbox);
        if ((int64_t)tmp.size() >= (int64_t)minPts) {
          for (size_t k = 0; k < tmp.size(); k++) {
            bool flag = false;
            if (dedupe) {
              // :73-80 `nei[j] == tmpList[k]` compares two boxed objects BY REFERENCE;
              // a fresh box is never identical to one already in nei.
              const uint64_t b = tbox[k];
              const uint64_t* nb = neibox.data();
              for (size_t j = 0, m = neibox.size(); j < m; j++) {
                if (nb[j] == b) {
                  flag = true;
                  break;
                }
              }
            }
            if (!flag) {
              nei.push_back(tmp[k]);
              if (dedupe) neibox.push_back(tbox[k]);
            }
          }
        }
      }
      lab[dpp] = c;  // :87 unconditional
    }
  }
};

// ---------------------------------------------------------------------------------
// CPU grid for the canonical formulation
// ---------------------------------------------------------------------------------
struct Grid {
  int gdim = 2;
  double h = 1.0;
  double mn[3] = {0, 0, 0};
  int64_t D[3] = {1, 1, 1};
  std::vector<int64_t> cell;    // per point cell coords, gdim each
  std::vector<int64_t> skey;    // sorted keys
  std::vector<int64_t> sidx;    // point index per sorted slot

  static constexpr int64_t kMaxDim = (int64_t)1 << 20;

  inline int64_t coord(double x, int a) const {
    double u = (x - mn[a]) / h;
    if (u >= 0.0 && u < (double)D[a]) return (int64_t)u;
    if (u >= (double)D[a]) return D[a] - 1;
    return 0;  // negative or NaN
  }

  void build(const double* c, int64_t n, int dim, int gd, double eps) {
    gdim = gd;
    double mx[3] = {0, 0, 0};
    bool any[3] = {false, false, false};
    for (int64_t i = 0; i < n; i++)
      for (int a = 0; a < gdim; a++) {
        double v = c[i * dim + a];
        if (!std::isfinite(v)) continue;
        if (!any[a]) {
          mn[a] = mx[a] = v;
          any[a] = true;
        } else {
          mn[a] = std::min(mn[a], v);
          mx[a] = std::max(mx[a], v);
        }
      }
    double range = 0.0;
    for (int a = 0; a < gdim; a++) range = std::max(range, mx[a] - mn[a]);
    // cell edge a hair LARGER than eps so that |dx| <= eps (up to rounding) always lands
    // within +-1 cell; the exact predicate is re-applied to every candidate.
    h = eps * (1.0 + 1.0 / 1048576.0);
    double hmin = range / (double)(kMaxDim - 1);
    if (!(h >= hmin)) h = hmin;
    if (!(h > 0.0) || !std::isfinite(h)) h = std::isfinite(h) ? 1.0 : h;
    for (int a = 0; a < gdim; a++) {
      double ext = (mx[a] - mn[a]) / h;
      int64_t dd = std::isfinite(ext) ? (int64_t)ext + 1 : 1;
      D[a] = std::max<int64_t>(1, std::min<int64_t>(dd, kMaxDim));
    }
    cell.resize((size_t)n * gdim);
    std::vector<int64_t> key(n);
    for (int64_t i = 0; i < n; i++) {
      int64_t k = 0;
      for (int a = gdim - 1; a >= 0; a--) {
        int64_t cc = coord(c[i * dim + a], a);
        cell[i * gdim + a] = cc;
        k = k * D[a] + cc;
      }
      key[i] = k;
    }
    sidx.resize(n);
    std::iota(sidx.begin(), sidx.end(), (int64_t)0);
    std::stable_sort(sidx.begin(), sidx.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
    skey.resize(n);
    for (int64_t t = 0; t < n; t++) skey[t] = key[sidx[t]];
  }

  template <class F>
  inline void for_candidates(int64_t i, F&& f) const {
    const int64_t* cc = &cell[i * gdim];
    int64_t x0 = std::max<int64_t>(0, cc[0] - 1), x1 = std::min<int64_t>(D[0] - 1, cc[0] + 1);
    int64_t y0 = std::max<int64_t>(0, cc[1] - 1), y1 = std::min<int64_t>(D[1] - 1, cc[1] + 1);
    int64_t z0 = 0, z1 = 0;
    if (gdim == 3) {
      z0 = std::max<int64_t>(0, cc[2] - 1);
      z1 = std::min<int64_t>(D[2] - 1, cc[2] + 1);
    }
    for (int64_t z = z0; z <= z1; z++)
      for (int64_t y = y0; y <= y1; y++) {
        int64_t base = (gdim == 3 ? (z * D[1] + y) : y) * D[0];
        int64_t lo = base + x0, hi = base + x1;
        auto it = std::lower_bound(skey.begin(), skey.end(), lo);
        for (size_t t = it - skey.begin(); t < skey.size() && skey[t] <= hi; t++) f(sidx[t]);
      }
  }
};

struct UF {
  std::vector<int64_t> p;
  explicit UF(int64_t n) : p(n) { std::iota(p.begin(), p.end(), (int64_t)0); }
  int64_t find(int64_t x) {
    while (p[x] != x) {
      p[x] = p[p[x]];
      x = p[x];
    }
    return x;
  }
  void unite(int64_t a, int64_t b) {
    a = find(a);
    b = find(b);
    if (a == b) return;
    if (a < b) p[b] = a; else p[a] = b;
  }
};

int dbscan_canonical_impl(const double* coords, int64_t n, int dim, int metric, double eps,
                          int min_pts, int32_t cf_in, uint8_t* classed, int32_t* labels,
                          uint8_t* is_key, int32_t* cf_out, int64_t* dist_evals) {
  if (n < 0 || !metric_ok(metric, dim) || metric == ORC_SIGNED_SUM_2D) return ORC_ERR_ARG;
  int32_t cf = cf_in;
  int64_t queries = 0;
  if (n == 0) {
    if (cf_out) *cf_out = cf;
    if (dist_evals) *dist_evals = 0;
    return ORC_OK;
  }
  Dist d{coords, dim, metric};
  if (!(eps >= 0.0)) {
    // negative or NaN eps: `getDisP(..) <= e` is never true, not even for the point
    // itself, so every neighbour list is empty (BC/DBImproved.cs:41).
    for (int64_t i = 0; i < n; i++) {
      if (classed[i]) continue;
      queries++;
      if (0 >= min_pts) {  // tmpList.Count (0) >= minPts, :105
        is_key[i] = 1;
        cf++;
        labels[i] = cf;  // expandCluster sets p.clusterId only; isClassed stays false
      }
    }
    if (cf_out) *cf_out = cf;
    if (dist_evals) *dist_evals = queries * n;
    return ORC_OK;
  }
  Grid g;
  g.build(coords, n, dim, metric == ORC_L2_3D ? 3 : 2, eps);
  std::vector<uint8_t> core(n, 0), expanding(n, 0), lonely(n, 0);
  for (int64_t i = 0; i < n; i++) {
    int64_t cnt = 0;
    g.for_candidates(i, [&](int64_t j) {
      if (d(i, j) <= eps) cnt++;
    });
    core[i] = cnt >= (int64_t)min_pts;
    expanding[i] = core[i] && !classed[i];
    lonely[i] = cnt == 0;  // non-finite coordinate: not even its own neighbour
  }
  UF uf(n);
  for (int64_t i = 0; i < n; i++) {
    if (!expanding[i]) continue;
    g.for_candidates(i, [&](int64_t j) {
      if (j < i && expanding[j] && d(i, j) <= eps) uf.unite(i, j);
    });
  }
  // clusters numbered cf+1.. in increasing order of the smallest expanding index
  std::vector<int32_t> comp_id(n, 0);
  std::vector<int64_t> seed_of;  // seed index of cluster (cf_in + 1 + k)
  for (int64_t i = 0; i < n; i++) {
    if (!expanding[i]) continue;
    int64_t r = uf.find(i);
    if (comp_id[r] == 0) {
      cf++;
      comp_id[r] = cf;
      seed_of.push_back(i);
    }
  }
  int64_t border_twice = 0;
  std::vector<int32_t> newlab(n, 0);
  for (int64_t i = 0; i < n; i++) {
    if (expanding[i]) {
      newlab[i] = comp_id[uf.find(i)];
      continue;
    }
    int32_t mx = 0, mnid = std::numeric_limits<int32_t>::max();
    g.for_candidates(i, [&](int64_t j) {
      if (expanding[j] && d(i, j) <= eps) {
        int32_t id = comp_id[uf.find(j)];
        mx = std::max(mx, id);
        mnid = std::min(mnid, id);
      }
    });
    newlab[i] = mx;
    if (mx != 0 && !classed[i]) {
      // reached first by cluster mnid; queried twice iff the main loop met it before
      // that cluster's seed (BC/DBImproved.cs:93-104 then :63-67).
      if (i < seed_of[mnid - cf_in - 1]) border_twice++;
    }
  }
  for (int64_t i = 0; i < n; i++) {
    if (!classed[i]) {
      queries++;
      if (core[i]) is_key[i] = 1;
    }
    if (newlab[i] != 0) {
      labels[i] = newlab[i];
      // a seed with an EMPTY neighbour list (min_pts <= 0, non-finite point) gets its id (:58) but is never popped, so
      // never marked classed (:63-65)
      if (!(expanding[i] && lonely[i])) classed[i] = 1;
    }
  }
  // a seed is queried a second time when it is popped from its own neighbour list; a lonely seed never is
  int64_t lonely_seeds = 0;
  for (int64_t i : seed_of) lonely_seeds += lonely[i];
  queries += (int64_t)seed_of.size() - lonely_seeds + border_twice;
  if (cf_out) *cf_out = cf;
  if (dist_evals) *dist_evals = queries * n;
  return ORC_OK;
}

// LINQ Average over a list of indices: sequential sum then / count (Enumerable.Average)
inline double average(const double* base, int stride, int comp, const std::vector<int64_t>& li) {
  double sum = 0.0;
  long long count = 0;
  for (int64_t i : li) {
    sum += base[i * stride + comp];
    count++;
  }
  return sum / (double)count;
}

// ---------------------------------------------------------------------------------
// small dense helpers for ICP
// ---------------------------------------------------------------------------------
// Matrix.StupidMultiply (BC/Matrix.cs:500-510): result starts at 0 and accumulates k ascending.
inline void mul33(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}
inline void mul31(const double A[9], const double v[3], double r[3]) {
  for (int i = 0; i < 3; i++) {
    double s = 0.0;
    for (int k = 0; k < 3; k++) s += A[i * 3 + k] * v[k];
    r[i] = s;
  }
}

}  // namespace

// =====================================================================================
extern "C" {

int orc_dbscan_literal(const double* coords, int64_t n, int dim, int metric, double eps,
                       int min_pts, int32_t cf_in, uint8_t* classed, int32_t* labels,
                       uint8_t* is_key, int32_t* cf_out, int64_t* dist_evals,
                       int run_dead_dedupe_scan) {
  if (n < 0 || !metric_ok(metric, dim)) return ORC_ERR_ARG;
  Literal L{Dist{coords, dim, metric}, n, eps, min_pts, classed, labels, is_key,
            run_dead_dedupe_scan != 0};
  int32_t cf = cf_in;
  std::vector<int64_t> tmp;
  std::vector<uint64_t> tbox;
  // BC/DBImproved.cs:91-114
  for (int64_t i = 0; i < n; i++) {
    if (classed[i]) continue;
    L.isKeyPoint(i, tmp, tbox);
    if ((int64_t)tmp.size() >= (int64_t)min_pts) {
      cf++;
      std::vector<int64_t> nei(tmp);
      std::vector<uint64_t> nbox(tbox);
      L.expandCluster(i, nei, nbox, cf);
    }
  }
  if (cf_out) *cf_out = cf;
  if (dist_evals) *dist_evals = L.d.evals;
  return ORC_OK;
}

int orc_dbscan_canonical(const double* coords, int64_t n, int dim, int metric, double eps,
                         int min_pts, int32_t cf_in, uint8_t* classed, int32_t* labels,
                         uint8_t* is_key, int32_t* cf_out, int64_t* dist_evals) {
  return dbscan_canonical_impl(coords, n, dim, metric, eps, min_pts, cf_in, classed, labels, is_key,
                               cf_out, dist_evals);
}

// -------------------------------------------------------------------------------------
// Staged form of the canonical formulation for a cloud spread over several ranks (SURVEY.md 8e mode 2):
// the same three facts of BC/DBImproved.cs as dbscan_canonical_impl -- core test :33-54, transitive
// expansion :56-90, border rule :87 -- with "index in the list" replaced by the caller's global position
// `ord`, and cluster numbers supplied by the caller once all ranks have agreed on them.
int orc_slab_begin(const double* coords, int64_t n, int dim, int metric, double eps, int min_pts,
                   const uint8_t* noexpand, const uint32_t* ord, uint32_t* rep, uint8_t* is_core,
                   int64_t* n_comp) {
  if (n <= 0 || !metric_ok(metric, dim) || metric == ORC_SIGNED_SUM_2D || !(eps >= 0.0)) return ORC_ERR_ARG;
  Dist d{coords, dim, metric};
  Grid g;
  g.build(coords, n, dim, metric == ORC_L2_3D ? 3 : 2, eps);
  std::vector<uint8_t> expanding(n, 0);
  for (int64_t i = 0; i < n; i++) {
    int64_t cnt = 0;
    g.for_candidates(i, [&](int64_t j) {
      if (d(i, j) <= eps) cnt++;
    });
    const bool core = cnt >= (int64_t)min_pts;
    if (is_core) is_core[i] = core;
    expanding[i] = core && !(noexpand && noexpand[i]);
  }
  UF uf(n);
  for (int64_t i = 0; i < n; i++) {
    if (!expanding[i]) continue;
    g.for_candidates(i, [&](int64_t j) {
      if (j < i && expanding[j] && d(i, j) <= eps) uf.unite(i, j);
    });
  }
  std::vector<uint32_t> mn(n, 0xFFFFFFFFu);
  for (int64_t i = 0; i < n; i++)
    if (expanding[i]) {
      int64_t r = uf.find(i);
      mn[r] = std::min(mn[r], ord[i]);
    }
  int64_t comps = 0;
  for (int64_t i = 0; i < n; i++) {
    rep[i] = expanding[i] ? mn[uf.find(i)] : 0xFFFFFFFFu;
    if (expanding[i] && uf.find(i) == i) comps++;
  }
  if (n_comp) *n_comp = comps;
  return ORC_OK;
}

int orc_slab_finish(const double* coords, int64_t n, int dim, int metric, double eps, const uint8_t* noexpand,
                    const uint32_t* ord, const uint32_t* rep, const uint32_t* map_rep, const uint32_t* map_k,
                    int64_t n_map, const int32_t* tab_gid, const uint32_t* tab_seed, int64_t n_tab, uint32_t own_lo,
                    uint32_t own_count, int32_t* labels, uint8_t* is_classed, int64_t* twice) {
  if (n <= 0 || !metric_ok(metric, dim) || metric == ORC_SIGNED_SUM_2D || !(eps >= 0.0)) return ORC_ERR_ARG;
  Dist d{coords, dim, metric};
  Grid g;
  g.build(coords, n, dim, metric == ORC_L2_3D ? 3 : 2, eps);
  auto table_index = [&](uint32_t r, int64_t* k) {
    const uint32_t* e = map_rep + n_map;
    const uint32_t* it = std::lower_bound(map_rep, e, r);
    if (it == e || *it != r) return false;
    *k = map_k[it - map_rep];
    return *k < n_tab;
  };
  int64_t tw = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t k = 0;
    if (rep[i] != 0xFFFFFFFFu) {
      if (!table_index(rep[i], &k)) return ORC_ERR_ARG;
      labels[i] = tab_gid[k];
      if (is_classed) is_classed[i] = 1;
      continue;
    }
    int64_t kmax = -1, kmin = n_tab;
    bool bad = false;
    g.for_candidates(i, [&](int64_t j) {
      if (rep[j] != 0xFFFFFFFFu && d(i, j) <= eps) {
        int64_t kj = 0;
        if (!table_index(rep[j], &kj)) bad = true;
        kmax = std::max(kmax, kj);
        kmin = std::min(kmin, kj);
      }
    });
    if (bad) return ORC_ERR_ARG;
    labels[i] = kmax >= 0 ? tab_gid[kmax] : 0;  // the table is ascending in cluster id: :87, last writer = largest id
    if (is_classed) is_classed[i] = kmax >= 0;
    const bool own = (uint32_t)(ord[i] - own_lo) < own_count;
    if (kmax >= 0 && own && !(noexpand && noexpand[i]) && ord[i] < tab_seed[kmin]) tw++;
  }
  if (twice) *twice = tw;
  return ORC_OK;
}

int orc_db_literal(const double* coords, int64_t n, int dim, double eps, int min_pts,
                   const uint8_t* shown, uint8_t* classed, int32_t* labels, uint8_t* is_key,
                   int32_t* cluster_amount, int32_t* points_amount, int64_t* dist_evals) {
  if (n < 0 || dim < 2) return ORC_ERR_ARG;
  Dist d{coords, dim, ORC_SIGNED_SUM_2D};
  // BC/DB.cs:33-55
  auto isKeyPoint = [&](int64_t p, std::vector<int64_t>& tmp) {
    int count = 0;
    tmp.clear();
    for (int64_t i = 0; i < n; i++) {
      if (!shown[i]) continue;
      if (d(p, i) <= eps) {
        ++count;
        tmp.push_back(i);
      }
    }
    if (count >= min_pts) is_key[p] = 1;
  };
  int32_t c = 0, pts = 0;
  std::vector<int64_t> tmp, tmp2;
  // BC/DB.cs:92-115
  for (int64_t i = 0; i < n; i++) {
    if (!shown[i]) continue;
    pts++;
    if (classed[i]) continue;
    isKeyPoint(i, tmp);
    if ((int64_t)tmp.size() >= (int64_t)min_pts) {
      c++;
      // BC/DB.cs:57-91
      std::vector<int64_t> nei(tmp);
      labels[i] = c;
      for (size_t t = 0; t < nei.size(); t++) {
        int64_t dpp = nei[t];
        if (!shown[dpp]) continue;
        if (!classed[dpp]) {
          classed[dpp] = 1;
          isKeyPoint(dpp, tmp2);
          if ((int64_t)tmp2.size() >= (int64_t)min_pts)
            nei.insert(nei.end(), tmp2.begin(), tmp2.end());  // dedupe scan never matches
        }
        labels[dpp] = c;
      }
    }
  }
  if (cluster_amount) *cluster_amount = c;
  if (points_amount) *points_amount = pts;
  if (dist_evals) *dist_evals = d.evals;
  return ORC_OK;
}

// -------------------------------------------------------------------------------------
// Block pipeline in three stages (partition / per-block DBImproved / CompleteWork3) so that the staged
// multi-GPU path can be checked stage by stage; orc_block_pipeline is their composition.
int orc_block_partition(const double* motor, int64_t n, int pts_in_cell, int brute_partition, int32_t* block_of,
                        int64_t* raw_o, int64_t* bl_o, int64_t* blockstart_o, int64_t blockstart_cap, int32_t* rows_o,
                        int32_t* cols_o, int64_t* m_o) {
  if (n < 0) return ORC_ERR_ARG;
  // FrmMain.cs:1224-1227: Min()/Max() on an empty list throw before the Count check of :1228
  if (n == 0) return ORC_ERR_EMPTY;
  if (pts_in_cell <= 0) return ORC_ERR_EMPTY;  // Take(0) -> cell.Max() throws (:1255)
  for (int64_t i = 0; i < n * 2; i++)
    if (!std::isfinite(motor[i])) return ORC_ERR_ARG;
  for (int64_t i = 0; i < n; i++) block_of[i] = -1;
  // :1224-1227
  double x_Min = motor[0], x_Max = motor[0], y_Min = motor[1], y_Max = motor[1];
  for (int64_t i = 1; i < n; i++) {
    x_Min = std::min(x_Min, motor[2 * i]);
    x_Max = std::max(x_Max, motor[2 * i]);
    y_Min = std::min(y_Min, motor[2 * i + 1]);
    y_Max = std::max(y_Max, motor[2 * i + 1]);
  }
  // :1229-1251 sort by max(x - x_Min, y - y_Min).  DEVIATION: stable (ties keep input order).
  std::vector<double> skey(n);
  for (int64_t i = 0; i < n; i++)
    skey[i] = std::max(motor[2 * i] - x_Min, motor[2 * i + 1] - y_Min);
  std::vector<int64_t> raw(n);  // rawData after the sort: raw[pos] = original index
  std::iota(raw.begin(), raw.end(), (int64_t)0);
  std::stable_sort(raw.begin(), raw.end(), [&](int64_t a, int64_t b) { return skey[a] < skey[b]; });
  // :1253-1258
  int64_t take = std::min<int64_t>(pts_in_cell, n);
  double cmx = motor[2 * raw[0]], cmy = motor[2 * raw[0] + 1];
  for (int64_t t = 1; t < take; t++) {
    cmx = std::max(cmx, motor[2 * raw[t]]);
    cmy = std::max(cmy, motor[2 * raw[t] + 1]);
  }
  double cell_x = cmx - x_Min, cell_y = cmy - y_Min;
  double fr = (y_Max - y_Min) / cell_y, fc = (x_Max - x_Min) / cell_x;
  // (int) of +inf / NaN is unspecified in C#; a zero-extent first block is an error here.
  if (!std::isfinite(fr) || !std::isfinite(fc)) return ORC_ERR_DEGENERATE;
  if (fr >= 2147483646.0 || fc >= 2147483646.0) return ORC_ERR_TOO_LARGE;
  int rows = (int)fr + 1, cols = (int)fc + 1;
  if ((int64_t)rows * cols > ((int64_t)1 << 31) - 1) return ORC_ERR_TOO_LARGE;
  int64_t nblocks = (int64_t)rows * cols;
  if (rows_o) *rows_o = rows;
  if (cols_o) *cols_o = cols;
  if (blockstart_o && blockstart_cap < nblocks + 1) return ORC_ERR_TOO_LARGE;

  // :1259-1285 blocks.  cells[0] = the first `take` points; every other (p,q) is the
  // rectangle filter of Tools.getListByScale2 (BC/Tools.cs:510-513): (lo, hi] on both axes,
  // last row / column stretched to the max.
  auto lo_x = [&](int q) { return x_Min + q * cell_x; };
  auto lo_y = [&](int p) { return y_Min + p * cell_y; };
  auto hi_x = [&](int q) { return q == cols - 1 ? x_Max : x_Min + (q + 1) * cell_x; };
  auto hi_y = [&](int p) { return p == rows - 1 ? y_Max : y_Min + (p + 1) * cell_y; };
  std::vector<uint8_t> in_first(n, 0);
  for (int64_t t = 0; t < take; t++) {
    in_first[raw[t]] = 1;
    block_of[raw[t]] = 0;
  }
  if (brute_partition) {
    for (int p = 0; p < rows; p++)
      for (int q = 0; q < cols; q++) {
        int64_t index = (int64_t)p * cols + q;
        if (index == 0) continue;  // :1266
        double a = lo_x(q), b = lo_y(p), c = hi_x(q), dd = hi_y(p);
        for (int64_t t = 0; t < n; t++) {
          int64_t i = raw[t];
          double x = motor[2 * i], y = motor[2 * i + 1];
          if (x > a && y > b && x <= c && y <= dd) {
            if (in_first[i]) continue;                // DEVIATION: block 0 keeps its points
            if (block_of[i] < 0) block_of[i] = (int32_t)index;
          }
        }
      }
  } else {
    auto find_axis = [&](double v, double vmin, double cellw, int cnt, bool is_x) -> int {
      // unique q with lo(q) < v <= hi(q); candidates around the arithmetic guess
      double g = (v - vmin) / cellw;
      int64_t q0 = std::isfinite(g) ? (int64_t)std::floor(g) : 0;
      for (int64_t q = q0 - 2; q <= q0 + 2; q++) {
        if (q < 0 || q >= cnt) continue;
        double lo = is_x ? lo_x((int)q) : lo_y((int)q);
        double hi = is_x ? hi_x((int)q) : hi_y((int)q);
        if (v > lo && v <= hi) return (int)q;
      }
      // stretched last interval can be far from the arithmetic guess
      {
        int q = cnt - 1;
        double lo = is_x ? lo_x(q) : lo_y(q);
        double hi = is_x ? hi_x(q) : hi_y(q);
        if (v > lo && v <= hi) return q;
      }
      return -1;
    };
    for (int64_t i = 0; i < n; i++) {
      if (in_first[i]) continue;
      int q = find_axis(motor[2 * i], x_Min, cell_x, cols, true);
      int p = find_axis(motor[2 * i + 1], y_Min, cell_y, rows, false);
      if (p < 0 || q < 0) continue;
      int64_t index = (int64_t)p * cols + q;
      if (index == 0) continue;
      block_of[i] = (int32_t)index;
    }
  }
  // block lists in rawData (sorted) order -- FindAll preserves list order
  std::vector<int64_t> bcount(nblocks + 1, 0);
  for (int64_t i = 0; i < n; i++)
    if (block_of[i] >= 0) bcount[block_of[i] + 1]++;
  for (int64_t b = 0; b < nblocks; b++) bcount[b + 1] += bcount[b];
  {
    std::vector<int64_t> fill(bcount.begin(), bcount.end() - 1);
    for (int64_t t = 0; t < n; t++) {
      int64_t i = raw[t];
      if (block_of[i] >= 0) bl_o[fill[block_of[i]]++] = i;
    }
  }
  if (raw_o) std::copy(raw.begin(), raw.end(), raw_o);
  if (blockstart_o) std::copy(bcount.begin(), bcount.end(), blockstart_o);
  if (m_o) *m_o = bcount[nblocks];
  return ORC_OK;
}

// FrmMain.cs:2782-2794 per block: new DBImproved().dbscan(cell, eps, minPts) for block_lo <= b < block_hi;
// local[t] = block-local cluster id of bl[t].  Returns the distance evaluations of these blocks.
int orc_block_cluster(const double* motor, const int64_t* bl, const int64_t* blockstart, int64_t block_lo,
                      int64_t block_hi, double eps, int min_pts, int use_canonical, int32_t* local,
                      int64_t* evals_o) {
  int64_t evals = 0;
  std::vector<int32_t> blab;
  std::vector<uint8_t> bcls, bkey;
  std::vector<double> bc;
  for (int64_t b = block_lo; b < block_hi; b++) {
    int64_t s = blockstart[b], cnt = blockstart[b + 1] - s;
    if (cnt == 0) continue;
    bc.resize(cnt * 2);
    blab.assign(cnt, 0);
    bcls.assign(cnt, 0);
    bkey.assign(cnt, 0);
    for (int64_t t = 0; t < cnt; t++) {
      bc[2 * t] = motor[2 * bl[s + t]];
      bc[2 * t + 1] = motor[2 * bl[s + t] + 1];
    }
    int32_t cf = 0;
    int64_t ev = 0;
    if (use_canonical)
      dbscan_canonical_impl(bc.data(), cnt, 2, ORC_L1_2D, eps, min_pts, 0, bcls.data(), blab.data(),
                            bkey.data(), &cf, &ev);
    else
      orc_dbscan_literal(bc.data(), cnt, 2, ORC_L1_2D, eps, min_pts, 0, bcls.data(), blab.data(),
                         bkey.data(), &cf, &ev, 0);
    evals += ev;
    for (int64_t t = 0; t < cnt; t++) local[s + t] = blab[t];
  }
  if (evals_o) *evals_o = evals;
  return ORC_OK;
}

// FrmMain.cs:1442-1520 CompleteWork3 on the block-local ids `local` [m] (block-major order).
int orc_block_finish(const double* motor, int64_t n, const int64_t* bl, const int64_t* blockstart, int64_t nblocks,
                     const int32_t* local, double eps, int min_pts, int small_max, int use_canonical,
                     int64_t evals_blocks, int32_t* labels, int64_t* merge_order, int64_t* m_out, int32_t* kept_o,
                     int32_t* del_sum_o, int32_t* cluster_amount_o, int64_t* dist_evals_o) {
  const int64_t mtot = blockstart[nblocks];
  for (int64_t i = 0; i < n; i++) labels[i] = 0;
  for (int64_t t = 0; t < mtot; t++) labels[bl[t]] = local[t];
  // clusterSum = 1 + sum of per-block clusterAmount (:1346, :2789) = 1 + sum of the largest local id
  int64_t clusterSum = 1;
  for (int64_t b = 0; b < nblocks; b++) {
    int32_t mx = 0;
    for (int64_t t = blockstart[b]; t < blockstart[b + 1]; t++) mx = std::max(mx, local[t]);
    clusterSum += mx;
  }
  if (blockstart[1] - blockstart[0] == 0) return ORC_ERR_INDEX;  // cells[0][0] :1442 (cannot happen: take>=1)
  int idLast, idNow = 0, id, clusLen = 0, delSum = 0;
  std::vector<int64_t> clusForMerge;
  clusForMerge.reserve(mtot);
  std::vector<int64_t> cellsorted;
  for (int64_t b = 0; b < nblocks; b++) {
    int64_t s = blockstart[b], cnt = blockstart[b + 1] - s;
    if (cnt == 0) continue;  // :1448
    // :1449-1459 sort by clusterId.  DEVIATION: stable.
    cellsorted.assign(bl + s, bl + s + cnt);
    std::stable_sort(cellsorted.begin(), cellsorted.end(),
                     [&](int64_t a, int64_t c) { return labels[a] < labels[c]; });
    idLast = labels[cellsorted[0]];
    if (idLast != 0) {
      idNow++;
      clusLen = 1;
    } else {
      clusLen = 0;
    }
    for (int64_t j = 0; j < cnt; j++) {
      int64_t pt = cellsorted[j];
      id = labels[pt];
      if (id == 0) {
        clusForMerge.push_back(pt);
      } else {
        if (id != idLast) {
          if (clusLen <= small_max && idLast != 0) {
            delSum++;
            for (int k = 0; k < clusLen; k++) {
              int64_t at = (int64_t)clusForMerge.size() - 1 - k;
              if (at < 0) return ORC_ERR_INDEX;  // ArgumentOutOfRangeException
              labels[clusForMerge[at]] = 0;      // :1487
            }
          } else {
            idNow++;
          }
          clusLen = 1;
        } else {
          clusLen++;
        }
        labels[pt] = idNow;  // :1500
        clusForMerge.push_back(pt);
        idLast = id;
      }
    }
  }
  // :1507-1520 global noise pass
  int32_t cf0 = (int32_t)(clusterSum - delSum - 1);
  if (kept_o) *kept_o = cf0;
  if (del_sum_o) *del_sum_o = delSum;
  std::vector<int64_t> zeroList, rest;
  for (int64_t pt : clusForMerge) (labels[pt] == 0 ? zeroList : rest).push_back(pt);
  int64_t Z = (int64_t)zeroList.size();
  std::vector<double> zc(Z * 2);
  std::vector<int32_t> zl(Z, 0);
  std::vector<uint8_t> zcl(Z, 0), zk(Z, 0);
  for (int64_t t = 0; t < Z; t++) {
    zc[2 * t] = motor[2 * zeroList[t]];
    zc[2 * t + 1] = motor[2 * zeroList[t] + 1];
  }
  int32_t cf = cf0;
  int64_t ev = 0;
  if (use_canonical)
    dbscan_canonical_impl(zc.data(), Z, 2, ORC_L1_2D, eps, min_pts, cf0, zcl.data(), zl.data(), zk.data(),
                          &cf, &ev);
  else
    orc_dbscan_literal(zc.data(), Z, 2, ORC_L1_2D, eps, min_pts, cf0, zcl.data(), zl.data(), zk.data(),
                       &cf, &ev, 0);
  for (int64_t t = 0; t < Z; t++) labels[zeroList[t]] = zl[t];
  if (cluster_amount_o) *cluster_amount_o = cf;
  if (dist_evals_o) *dist_evals_o = evals_blocks + ev;
  int64_t m = 0;
  if (merge_order) {
    for (int64_t pt : rest) merge_order[m++] = pt;
    for (int64_t pt : zeroList) merge_order[m++] = pt;  // :1517-1520
  } else {
    m = (int64_t)rest.size() + Z;
  }
  if (m_out) *m_out = m;
  return ORC_OK;
}

int orc_block_pipeline(const double* motor, int64_t n, double eps, int min_pts, int pts_in_cell,
                       int small_max, int use_canonical, int brute_partition, int32_t* labels,
                       int32_t* block_of, int64_t* merge_order, int64_t* m_out, int32_t* rows_o,
                       int32_t* cols_o, int32_t* kept_o, int32_t* del_sum_o,
                       int32_t* cluster_amount_o, int64_t* dist_evals_o) {
  return orc_block_pipeline_keyed(motor, motor, n, eps, min_pts, pts_in_cell, small_max, use_canonical, brute_partition,
                                  labels, block_of, merge_order, m_out, rows_o, cols_o, kept_o, del_sum_o,
                                  cluster_amount_o, dist_evals_o);
}

// getClusterFromList (FrmMain.cs:1136-1213): the partition reads `key` = (X, Y) (Tools.getListByScale, BC/Tools.cs:507-509),
// StartCode and CompleteWork3 cluster on motor exactly as in getClusterFromMotor.
int orc_block_pipeline_keyed(const double* key, const double* motor, int64_t n, double eps, int min_pts, int pts_in_cell,
                             int small_max, int use_canonical, int brute_partition, int32_t* labels,
                             int32_t* block_of, int64_t* merge_order, int64_t* m_out, int32_t* rows_o,
                             int32_t* cols_o, int32_t* kept_o, int32_t* del_sum_o,
                             int32_t* cluster_amount_o, int64_t* dist_evals_o) {
  int32_t rows = 0, cols = 0;
  int64_t m = 0;
  std::vector<int64_t> bl(n > 0 ? n : 1);
  // two calls: the first sizes rows*cols, the second fills blockstart
  int rc = orc_block_partition(key, n, pts_in_cell, brute_partition, block_of, nullptr, bl.data(), nullptr, 0, &rows,
                               &cols, &m);
  if (rc) return rc;
  int64_t nblocks = (int64_t)rows * cols;
  std::vector<int64_t> blockstart(nblocks + 1, 0);
  for (int64_t i = 0; i < n; i++)
    if (block_of[i] >= 0) blockstart[block_of[i] + 1]++;
  for (int64_t b = 0; b < nblocks; b++) blockstart[b + 1] += blockstart[b];
  if (rows_o) *rows_o = rows;
  if (cols_o) *cols_o = cols;
  std::vector<int32_t> local(m > 0 ? m : 1, 0);
  int64_t ev = 0;
  rc = orc_block_cluster(motor, bl.data(), blockstart.data(), 0, nblocks, eps, min_pts, use_canonical, local.data(), &ev);
  if (rc) return rc;
  return orc_block_finish(motor, n, bl.data(), blockstart.data(), nblocks, local.data(), eps, min_pts, small_max,
                          use_canonical, ev, labels, merge_order, m_out, kept_o, del_sum_o, cluster_amount_o,
                          dist_evals_o);
}

// -------------------------------------------------------------------------------------
int orc_centroids(const double* xyz, const double* motor, const int32_t* labels,
                  const int64_t* order, int64_t m, int32_t K, double* c3, double* c2,
                  int64_t* counts) {
  if (m < 0 || K < 0) return ORC_ERR_ARG;
  std::vector<std::vector<int64_t>> li(K);
  for (int64_t t = 0; t < m; t++) {
    int64_t i = order ? order[t] : t;
    int32_t id = labels[i];
    if (id != 0) {
      if (id < 1 || id > K) return ORC_ERR_INDEX;  // clusList[p.clusterId - 1] BC/Tools.cs:185
      li[id - 1].push_back(i);
    }
  }
  const double nan = std::numeric_limits<double>::quiet_NaN();
  for (int32_t k = 0; k < K; k++) {
    counts[k] = (int64_t)li[k].size();
    if (li[k].empty()) {  // :191 skipped
      if (c3) c3[3 * k] = c3[3 * k + 1] = c3[3 * k + 2] = nan;
      if (c2) c2[2 * k] = c2[2 * k + 1] = nan;
      continue;
    }
    if (c3 && xyz)
      for (int a = 0; a < 3; a++) c3[3 * k + a] = average(xyz, 3, a, li[k]);  // :192
    if (c2 && motor)
      for (int a = 0; a < 2; a++) c2[2 * k + a] = average(motor, 2, a, li[k]);  // :193
  }
  return ORC_OK;
}

// Tools.getFixedPtsCentroid (BC/Tools.cs:78-111), statement by statement: sequential sums in list order.
int orc_fixed_centroids(const double* xyz, const int32_t* group, const int32_t* cluster_id, const int32_t* pts_count,
                        int64_t n, int32_t K, int ignore_dup, double* c3, int64_t* inside_num) {
  if (n < 0 || K < 0) return ORC_ERR_ARG;
  std::vector<std::vector<int64_t>> li(K);
  for (int64_t i = 0; i < n; i++) {
    const int32_t g = group[i];
    if (g == 0) continue;
    if (g < 1 || g > K) return ORC_ERR_INDEX;
    li[g - 1].push_back(i);
  }
  for (int32_t k = 0; k < K; k++) {
    double X = 0, Y = 0, Z = 0;  // new Point3D(): :82
    int64_t insideNum = 0;       // int in the C#
    for (int64_t i : li[k]) {
      const int32_t cid = cluster_id ? cluster_id[i] : group[i];
      if (cid != 0 && ignore_dup) {  // :86-92
        X += xyz[3 * i];
        Y += xyz[3 * i + 1];
        Z += xyz[3 * i + 2];
        insideNum++;
      } else {  // :95-100
        X += xyz[3 * i] * pts_count[i];
        Y += xyz[3 * i + 1] * pts_count[i];
        Z += xyz[3 * i + 2] * pts_count[i];
        insideNum += pts_count[i];
      }
    }
    c3[3 * k] = X / (double)insideNum;  // :102-104 (double / int)
    c3[3 * k + 1] = Y / (double)insideNum;
    c3[3 * k + 2] = Z / (double)insideNum;
    if (inside_num) inside_num[k] = insideNum;
    if (li[k].empty()) return ORC_ERR_INDEX;  // :105 clusList[i].li[0] throws ArgumentOutOfRangeException
  }
  return ORC_OK;
}

int orc_merge_ids(const double* cxy, const int32_t* ids, int32_t K, double thr, int32_t* map_to,
                  int32_t* merge_count) {
  if (K < 0) return ORC_ERR_ARG;
  // BC/Tools.cs:584-592
  std::vector<int32_t> lab(K, 0);
  std::vector<uint8_t> cls(K, 0), key(K, 0);
  int32_t cf = 0;
  int64_t ev = 0;
  int rc = orc_dbscan_literal(cxy, K, 2, ORC_L1_2D, thr, 2, 0, cls.data(), lab.data(), key.data(), &cf,
                              &ev, 0);
  if (rc) return rc;
  // :593-618
  std::vector<int32_t> set;
  auto contains = [&](int32_t v) { return std::find(set.begin(), set.end(), v) != set.end(); };
  std::vector<std::pair<int32_t, int32_t>> dick;
  int mc = 0;
  for (int32_t k = 0; k < K; k++) map_to[k] = 0;
  for (int32_t p = 0; p < K; p++) {
    if (lab[p] != 0) {
      if (!contains(ids[p])) {
        set.push_back(ids[p]);
        for (int32_t q = 0; q < K; q++) {
          if (lab[q] == lab[p] && ids[q] != ids[p]) {
            set.push_back(ids[q]);
            for (auto& kv : dick)
              if (kv.first == ids[q]) return ORC_ERR_ARG;  // Dictionary.Add duplicate key throws
            dick.emplace_back(ids[q], ids[p]);
            map_to[q] = ids[p];
            mc++;
          }
        }
      }
    } else {
      set.push_back(ids[p]);
    }
  }
  if (merge_count) *merge_count = mc;
  return ORC_OK;
}

int orc_refresh_by_dictionary(const double* xyz, const double* motor, int32_t* labels,
                              const int64_t* order, int64_t m, int32_t K,
                              const int32_t* map_by_id, int32_t* new_k, double* c3, double* c2,
                              int64_t* counts) {
  if (m < 0 || K < 0) return ORC_ERR_ARG;
  // clusList as built by GetClusList: position id-1, li in visiting order
  std::vector<std::vector<int64_t>> li(K);
  for (int64_t t = 0; t < m; t++) {
    int64_t i = order ? order[t] : t;
    int32_t id = labels[i];
    if (id != 0) {
      if (id < 1 || id > K) return ORC_ERR_INDEX;
      li[id - 1].push_back(i);
    }
  }
  // BC/Tools.cs:525-533: append merged clusters' points to their target (clusList order)
  for (int32_t k = 0; k < K; k++) {
    int32_t tgt = map_by_id[k];
    if (tgt != 0) {
      if (tgt < 1 || tgt > K) return ORC_ERR_INDEX;
      // the loop reads ob.li while appending to ANOTHER list; a self-map would modify the
      // collection being enumerated (InvalidOperationException)
      if (tgt - 1 == k) return ORC_ERR_ARG;
      for (int64_t pt : li[k]) li[tgt - 1].push_back(pt);
    }
  }
  // :534-565 remove merged, sort by clusId (already ascending), renumber 1..K'
  int32_t nk = 0;
  for (int32_t k = 0; k < K; k++) {
    if (map_by_id[k] != 0) continue;
    if (li[k].empty()) return ORC_ERR_EMPTY;  // :568 Average() of an empty list throws
    for (int64_t pt : li[k]) labels[pt] = nk + 1;
    counts[nk] = (int64_t)li[k].size();
    if (c3 && xyz)
      for (int a = 0; a < 3; a++) c3[3 * nk + a] = average(xyz, 3, a, li[k]);
    if (c2 && motor)
      for (int a = 0; a < 2; a++) c2[2 * nk + a] = average(motor, 2, a, li[k]);
    nk++;
  }
  if (new_k) *new_k = nk;
  return ORC_OK;
}

// -------------------------------------------------------------------------------------
void orc_find_closest(const double* model, int64_t nm, const double* p, int64_t nd, int32_t* idx) {
  for (int64_t i = 0; i < nd; i++) {
    const double* d = p + 3 * i;
    int64_t j = 0, order = 0;
    double mn = (d[0] - model[0]) * (d[0] - model[0]) + (d[1] - model[1]) * (d[1] - model[1]) +
                (d[2] - model[2]) * (d[2] - model[2]);
    for (j = 1; j < nm; j++) {
      const double* q = model + 3 * j;
      double dd = (d[0] - q[0]) * (d[0] - q[0]) + (d[1] - q[1]) * (d[1] - q[1]) +
                  (d[2] - q[2]) * (d[2] - q[2]);
      if (dd < mn) {  // strict: lowest index wins ties
        mn = dd;
        order = j;
      }
    }
    idx[i] = (int32_t)order;
  }
}

void orc_mean3(const double* p, int64_t n, double mean[3]) {
  double x = 0, y = 0, z = 0;
  for (int64_t i = 0; i < n; i++) {
    x += p[3 * i];
    y += p[3 * i + 1];
    z += p[3 * i + 2];
  }
  mean[0] = x / (double)n;
  mean[1] = y / (double)n;
  mean[2] = z / (double)n;
}

void orc_trans_point(const double* src, int64_t n, const double R[9], const double T[3], double* dst) {
  for (int64_t i = 0; i < n; i++) {
    double r[3];
    mul31(R, src + 3 * i, r);
    for (int a = 0; a < 3; a++) dst[3 * i + a] = r[a] + T[a];
  }
}

void orc_calc_rotation(const double q[4], double R[9]) {
  R[0] = q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3];
  R[1] = 2.0 * (q[1] * q[2] - q[0] * q[3]);
  R[2] = 2.0 * (q[1] * q[3] + q[0] * q[2]);
  R[3] = 2.0 * (q[1] * q[2] + q[0] * q[3]);
  R[4] = q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3];
  R[5] = 2.0 * (q[2] * q[3] - q[0] * q[1]);
  R[6] = 2.0 * (q[1] * q[3] - q[0] * q[2]);
  R[7] = 2.0 * (q[2] * q[3] + q[0] * q[1]);
  R[8] = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
}

void orc_icp_sums(const double* model, int64_t nm, const double* p, int64_t nd, double s[16]) {
  std::vector<int32_t> idx(nd);
  orc_find_closest(model, nm, p, nd, idx.data());
  for (int k = 0; k < 16; k++) s[k] = 0.0;
  for (int64_t i = 0; i < nd; i++) {
    const double* a = p + 3 * i;
    const double* y = model + 3 * (int64_t)idx[i];
    for (int r = 0; r < 3; r++) {
      s[r] += a[r];
      s[3 + r] += y[r];
      for (int c = 0; c < 3; c++) s[6 + 3 * r + c] += a[r] * y[c];  // BC/ICP.cs:38-52
    }
    double e0 = a[0] - y[0], e1 = a[1] - y[1], e2 = a[2] - y[2];
    s[15] += e0 * e0 + e1 * e1 + e2 * e2;  // BC/ICP.cs:129-133
  }
}

int orc_jacobi_sym(double* A, int n, double* evals, double* V, int max_sweeps) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  double scale = 0.0;
  for (int i = 0; i < n * n; i++) scale = std::max(scale, std::fabs(A[i]));
  if (scale == 0.0) {
    for (int i = 0; i < n; i++) evals[i] = 0.0;
    return 0;
  }
  for (int it = 0; it < max_sweeps * n * n; it++) {
    int p = 0, q = 1;
    double fm = 0.0;
    for (int i = 1; i < n; i++)
      for (int j = 0; j < i; j++)
        if (std::fabs(A[i * n + j]) > fm) {
          fm = std::fabs(A[i * n + j]);
          p = j;
          q = i;
        }
    if (fm <= 1e-300 || fm < scale * 1e-17) break;
    double app = A[p * n + p], aqq = A[q * n + q], apq = A[p * n + q];
    double theta = (aqq - app) / (2.0 * apq);
    double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
    double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
    for (int k = 0; k < n; k++) {
      double akp = A[k * n + p], akq = A[k * n + q];
      A[k * n + p] = c * akp - s * akq;
      A[k * n + q] = s * akp + c * akq;
    }
    for (int k = 0; k < n; k++) {
      double apk = A[p * n + k], aqk = A[q * n + k];
      A[p * n + k] = c * apk - s * aqk;
      A[q * n + k] = s * apk + c * aqk;
    }
    for (int k = 0; k < n; k++) {
      double vkp = V[k * n + p], vkq = V[k * n + q];
      V[k * n + p] = c * vkp - s * vkq;
      V[k * n + q] = s * vkp + c * vkq;
    }
  }
  for (int i = 0; i < n; i++) evals[i] = A[i * n + i];
  return 0;
}

int orc_horn_from_sums(const double s[16], int64_t nd, double R1[9], double T1[3]) {
  double N = (double)nd;
  double muP[3] = {s[0] / N, s[1] / N, s[2] / N};
  double muY[3] = {s[3] / N, s[4] / N, s[5] / N};
  double m[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) m[3 * r + c] = s[6 + 3 * r + c] / N - muP[r] * muY[c];
  // A = m - m^T; delta = (A12, A20, A01)   (BC/ICP.cs:68-76 intent)
  double delta[3] = {m[1 * 3 + 2] - m[2 * 3 + 1], m[2 * 3 + 0] - m[0 * 3 + 2], m[0 * 3 + 1] - m[1 * 3 + 0]};
  double tr = m[0] + m[4] + m[8];
  double Q[16];
  Q[0] = tr;
  for (int i = 0; i < 3; i++) {
    Q[1 + i] = delta[i];
    Q[4 * (1 + i)] = delta[i];
    for (int j = 0; j < 3; j++) Q[4 * (1 + i) + 1 + j] = m[3 * i + j] + m[3 * j + i] - (i == j ? tr : 0.0);
  }
  double ev[4], V[16];
  orc_jacobi_sym(Q, 4, ev, V, 64);
  int best = 0;
  for (int i = 1; i < 4; i++)
    if (ev[i] > ev[best]) best = i;
  double q[4] = {V[0 * 4 + best], V[1 * 4 + best], V[2 * 4 + best], V[3 * 4 + best]};
  double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (!(nrm > 0.0)) return ORC_ERR_ARG;
  for (int i = 0; i < 4; i++) q[i] /= nrm;
  orc_calc_rotation(q, R1);
  double rp[3];
  mul31(R1, muP, rp);
  for (int i = 0; i < 3; i++) T1[i] = muY[i] - rp[i];  // BC/ICP.cs:114-124
  return ORC_OK;
}

int orc_icp(const double* model, int64_t nm, const double* data, int64_t nd, double tol, int max_iter,
            int stop_rule, double R[9], double T[3], double* sse_o, double* rmse_o, int32_t* iters_o) {
  if (nm <= 0) return ORC_ERR_EMPTY;  // model[0] BC/ICP.cs:233 throws
  if (nd < 0 || max_iter < 1) return ORC_ERR_ARG;
  std::vector<double> P(data, data + 3 * nd);
  double pre_d = 0.0, d = 0.0;
  int round = 0;
  bool go;
  do {
    pre_d = d;
    double s[16], R1[9], T1[3];
    orc_icp_sums(model, nm, P.data(), nd, s);
    int rc = nd > 0 ? orc_horn_from_sums(s, nd, R1, T1) : ORC_OK;
    d = s[15];
    round++;
    if (stop_rule == ORC_STOP_RMSE)
      go = nd > 0 && std::sqrt(d / (double)nd) >= tol;
    else
      go = std::fabs(d - pre_d) >= tol;  // BC/ICP.cs:149,180
    if (go && nd > 0) {
      if (rc) return rc;
      if (round == 1) {
        std::memcpy(R, R1, sizeof(double) * 9);
        std::memcpy(T, T1, sizeof(double) * 3);
      } else {
        double tR[9], tT[3];
        mul33(R1, R, tR);   // BC/ICP.cs:167
        mul31(R1, T, tT);   // :168
        std::memcpy(R, tR, sizeof(tR));
        for (int i = 0; i < 3; i++) T[i] = tT[i] + T1[i];  // :175-176
      }
      orc_trans_point(data, nd, R, T, P.data());  // :178
    }
  } while (go && round < max_iter);
  if (sse_o) *sse_o = d;
  if (rmse_o) *rmse_o = nd > 0 ? std::sqrt(d / (double)nd) : 0.0;
  if (iters_o) *iters_o = round;
  return ORC_OK;
}

// -------------------------------------------------------------------------------------
// Minimal bounding circle of a cluster: Geometry.FindMinimalBoundingCircle,
// BC/Geometry.cs:247-319, with MakeConvexHull :122-208, AngleValue :220-246, CircleEnclosesPoints :322-337,
// FindCircle :340-372, FindIntersection :373-432, line by line.  HullCull (:83-120) culls nothing: the
// Rectangle2D it tests against never gets Left/Right/Top/Bottom set (BC/DataModel.cs:191-208), so they are 0
// and the "outside the box" test is always true; only NaN coordinates would be dropped.
namespace {
double AngleValue(double x1, double y1, double x2, double y2) {
  double dx = x2 - x1, ax = std::fabs(dx), dy = y2 - y1, ay = std::fabs(dy), t;
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
struct P2 {
  double x, y;
};
bool CircleEnclosesPoints(P2 c, double radius2, const std::vector<P2>& pts, int s1, int s2, int s3) {
  for (int i = 0; i < (int)pts.size(); i++)
    if (i != s1 && i != s2 && i != s3) {
      double dx = c.x - pts[i].x, dy = c.y - pts[i].y;
      if (dx * dx + dy * dy > radius2) return false;
    }
  return true;
}
void FindCircle(P2 a, P2 b, P2 c, P2* center, double* radius2) {
  double x1 = (b.x + a.x) / 2, y1 = (b.y + a.y) / 2, dy1 = b.x - a.x, dx1 = -(b.y - a.y);
  double x2 = (c.x + b.x) / 2, y2 = (c.y + b.y) / 2, dy2 = c.x - b.x, dx2 = -(c.y - b.y);
  // FindIntersection(p1=(x1,y1), p2=(x1+dx1,y1+dy1), p3=(x2,y2), p4=(x2+dx2,y2+dy2))
  P2 p1{x1, y1}, p2{x1 + dx1, y1 + dy1}, p3{x2, y2}, p4{x2 + dx2, y2 + dy2};
  double dx12 = p2.x - p1.x, dy12 = p2.y - p1.y, dx34 = p4.x - p3.x, dy34 = p4.y - p3.y;
  double denominator = (dy12 * dx34 - dx12 * dy34);
  double t1 = ((p1.x - p3.x) * dy34 + (p3.y - p1.y) * dx34) / denominator;  // double division never throws
  center->x = p1.x + dx12 * t1;
  center->y = p1.y + dy12 * t1;
  double dx = center->x - a.x, dy = center->y - a.y;
  *radius2 = dx * dx + dy * dy;
}
}  // namespace

int orc_min_circle(const double* pts, int64_t cnt, double center[2], double* radius, double* hull_xy,
                   int64_t hull_cap, int32_t* hull_n) {
  if (cnt <= 0) return ORC_ERR_EMPTY;  // points[0]
  // HullCull: keeps every point whose coordinates are not NaN (x <= 0 || x >= 0 || ...)
  std::vector<int64_t> rem;
  for (int64_t i = 0; i < cnt; i++) {
    double x = pts[2 * i], y = pts[2 * i + 1];
    if (x <= 0 || x >= 0 || y <= 0 || y >= 0) rem.push_back(i);
  }
  if (rem.empty()) return ORC_ERR_EMPTY;
  // :129-150 smallest Y, then smallest X (first wins exact ties)
  size_t bi = 0;
  for (size_t k = 0; k < rem.size(); k++) {
    double y = pts[2 * rem[k] + 1], x = pts[2 * rem[k]];
    double by = pts[2 * rem[bi] + 1], bx = pts[2 * rem[bi]];
    if (y < by || (y == by && x < bx)) bi = k;
  }
  std::vector<P2> hull;
  hull.push_back(P2{pts[2 * rem[bi]], pts[2 * rem[bi] + 1]});
  rem.erase(rem.begin() + bi);
  double sweep_angle = 0;
  while (!rem.empty()) {  // the C# indexes points[0] at :168 and would throw on an empty list
    double X = hull.back().x, Y = hull.back().y;
    size_t best = 0;
    double best_angle = 3600;
    for (size_t k = 0; k < rem.size(); k++) {
      double ta = AngleValue(X, Y, pts[2 * rem[k]], pts[2 * rem[k] + 1]);
      if (ta >= sweep_angle && best_angle > ta) {
        best_angle = ta;
        best = k;
      }
    }
    double first_angle = AngleValue(X, Y, hull[0].x, hull[0].y);
    if (first_angle >= sweep_angle && best_angle >= first_angle) break;
    hull.push_back(P2{pts[2 * rem[best]], pts[2 * rem[best] + 1]});
    rem.erase(rem.begin() + best);
    sweep_angle = best_angle;
  }
  if (hull_n) *hull_n = (int32_t)hull.size();
  if (hull_xy)
    for (size_t k = 0; k < hull.size() && (int64_t)k < hull_cap; k++) {
      hull_xy[2 * k] = hull[k].x;
      hull_xy[2 * k + 1] = hull[k].y;
    }
  // :252-312
  P2 best_center{pts[0], pts[1]};
  double best_radius2 = std::numeric_limits<double>::max();
  const int h = (int)hull.size();
  for (int i = 0; i < h - 1; i++)
    for (int j = i + 1; j < h; j++) {
      P2 tc{(hull[i].x + hull[j].x) / 2.0, (hull[i].y + hull[j].y) / 2.0};
      double dx = tc.x - hull[i].x, dy = tc.y - hull[i].y;
      double tr2 = dx * dx + dy * dy;
      if (tr2 < best_radius2 && CircleEnclosesPoints(tc, tr2, hull, i, j, -1)) {
        best_center = tc;
        best_radius2 = tr2;
      }
    }
  for (int i = 0; i < h - 2; i++)
    for (int j = i + 1; j < h - 1; j++)
      for (int k = j + 1; k < h; k++) {
        P2 tc;
        double tr2;
        FindCircle(hull[i], hull[j], hull[k], &tc, &tr2);
        if (tr2 < best_radius2 && CircleEnclosesPoints(tc, tr2, hull, i, j, k)) {
          best_center = tc;
          best_radius2 = tr2;
        }
      }
  center[0] = best_center.x;
  center[1] = best_center.y;
  *radius = best_radius2 == std::numeric_limits<double>::max() ? 0.0 : std::sqrt(best_radius2);
  return ORC_OK;
}

// Tools.getCircles (BC/Tools.cs:394-409): one circle per cluster with more than 3 points.
int orc_get_circles(const double* xy, const int32_t* labels, const int64_t* order, int64_t m, int32_t K,
                    double* centers, double* radius, uint8_t* valid, int32_t* hull_n) {
  if (m < 0 || K < 0) return ORC_ERR_ARG;
  std::vector<std::vector<double>> li(K);
  for (int64_t t = 0; t < m; t++) {
    int64_t i = order ? order[t] : t;
    int32_t id = labels[i];
    if (id != 0) {
      if (id < 1 || id > K) return ORC_ERR_INDEX;
      li[id - 1].push_back(xy[2 * i]);
      li[id - 1].push_back(xy[2 * i + 1]);
    }
  }
  for (int32_t k = 0; k < K; k++) {
    valid[k] = 0;
    radius[k] = 0;
    centers[2 * k] = centers[2 * k + 1] = 0;
    if (hull_n) hull_n[k] = 0;
    int64_t cnt = (int64_t)li[k].size() / 2;
    if (cnt <= 3) continue;  // :400
    int32_t hn = 0;
    int rc = orc_min_circle(li[k].data(), cnt, centers + 2 * k, radius + k, nullptr, 0, &hn);
    if (rc) return rc;
    valid[k] = 1;
    if (hull_n) hull_n[k] = hn;
  }
  return ORC_OK;
}

// Import conversion + duplicate removal: MainForm.AddFolder, FrmMain.cs:1011-1090, for the
// scan-point types 1 (drop duplicates) and 2 (keep them).  rows = (motor_x, motor_y, Distance).
// state: 0 = filtered (:1011 Distance == 0 or > 1000), 1 = kept, 2 = duplicate (:1065-1067: an earlier KEPT point p
// with p.X == tmpx && p.Y == tmpy && p.Z == tmpz -- the stored, direction-mapped X,Y against the unmapped tmpx,tmpy).
// literal != 0 runs the C#'s O(n^2) FindAll; otherwise a hash on the triple (valid when xdir = 2, ydir = 1, the
// ImportPts defaults, where stored and queried triples coincide).
int orc_import_convert(const double* rows, int64_t n, double x_angle, double y_angle, int xdir, int ydir, int dedupe,
                       int literal, double* xyz, uint8_t* state, int64_t* kept_o, int64_t* dup_o) {
  if (n < 0 || xdir < 1 || xdir > 4 || ydir < 1 || ydir > 4) return ORC_ERR_ARG;
  if (dedupe && !literal && !(xdir == 2 && ydir == 1)) return ORC_ERR_ARG;
  const double PI = 3.14159265358979323846;  // Math.PI
  int64_t kept = 0, dup = 0;
  std::vector<int64_t> keptidx;
  struct K3 {
    double a, b, c;
  };
  auto canon = [](double v) { return v == 0 ? 0.0 : v; };
  std::vector<std::pair<uint64_t, int64_t>> table;  // open addressing: (hash+1, index)
  size_t cap = 1;
  while (cap < (size_t)(2 * n + 16)) cap <<= 1;
  if (dedupe && !literal) table.assign(cap, {0, -1});
  auto hash3 = [&](double a, double b, double c) {
    uint64_t h = 1469598103934665603ull;
    for (double v : {canon(a), canon(b), canon(c)}) {
      uint64_t u;
      std::memcpy(&u, &v, 8);
      h = (h ^ u) * 1099511628211ull;
      h ^= h >> 29;
    }
    return h;
  };
  for (int64_t i = 0; i < n; i++) {
    const double mx = rows[3 * i], my = rows[3 * i + 1], D = rows[3 * i + 2];
    xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = 0;
    if (D == 0 || D > 1000) {
      state[i] = 0;
      continue;
    }
    const double yangjiao = (-2) * (mx - x_angle) / 180 * PI;
    const double fangweijiao = 2 * (my - y_angle) / 180 * PI;
    const double tmpx = D * std::cos(yangjiao) * std::sin(fangweijiao);
    const double tmpy = D * std::sin(yangjiao) * std::cos(fangweijiao);
    const double tmpz = D * std::cos(yangjiao);
    const double pick[5] = {0, tmpy, tmpx, -tmpy, -tmpx};
    const double X = pick[xdir], Y = pick[ydir], Z = tmpz;
    xyz[3 * i] = X;
    xyz[3 * i + 1] = Y;
    xyz[3 * i + 2] = Z;
    bool isdup = false;
    if (dedupe) {
      if (literal) {
        for (int64_t j : keptidx)
          if (xyz[3 * j] == tmpx && xyz[3 * j + 1] == tmpy && xyz[3 * j + 2] == tmpz) {
            isdup = true;
            break;
          }
      } else if (tmpx == tmpx && tmpy == tmpy && tmpz == tmpz) {  // NaN never equals anything
        uint64_t h = hash3(tmpx, tmpy, tmpz);
        size_t sl = h & (cap - 1);
        for (;;) {
          if (table[sl].second < 0) {
            table[sl] = {h, i};
            break;
          }
          int64_t j = table[sl].second;
          if (xyz[3 * j] == tmpx && xyz[3 * j + 1] == tmpy && xyz[3 * j + 2] == tmpz) {
            isdup = true;
            break;
          }
          sl = (sl + 1) & (cap - 1);
        }
      }
    }
    if (isdup) {
      state[i] = 2;
      dup++;
    } else {
      state[i] = 1;
      kept++;
      if (dedupe && literal) keptidx.push_back(i);
    }
  }
  if (kept_o) *kept_o = kept;
  if (dup_o) *dup_o = dup;
  return ORC_OK;
}

// "VTK-like" ICP: the knobs FrmMain.ICP() sets on vtkIterativeClosestPointTransform
// (FrmMain.cs:851-862: RigidBody, MaximumNumberOfIterations 100, StartByMatchingCentroidsOn, mean-distance check
// left off) following the header-documented behaviour (vtk/include/vtk-5.0/vtkIterativeClosestPointTransform.h
// :49-180): landmarks = every step-th source point, step = ns / max_landmarks when ns > max_landmarks; optional
// initial translation target centroid - source centroid; then max_iter rounds of closest target point (vertex)
// + rigid landmark transform (Horn), accumulated.  PARITY UNPINNED: VTK 5.0's sources are not in the tree.
int orc_icp_vtklike(const double* source, int64_t ns, const double* target, int64_t nt, int max_iter,
                    int max_landmarks, int start_by_centroids, double M[16], double* mean_dist, int32_t* iters_o) {
  if (ns <= 0 || nt <= 0) return ORC_ERR_EMPTY;
  if (max_iter < 1 || max_landmarks < 1) return ORC_ERR_ARG;
  int64_t step = 1;
  if (ns > max_landmarks) step = ns / max_landmarks;
  int64_t nb = ns / step;
  std::vector<double> a(3 * nb);
  for (int64_t i = 0, j = 0; i < nb; i++, j += step)
    for (int c = 0; c < 3; c++) a[3 * i + c] = source[3 * j + c];
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[3] = {0, 0, 0};
  if (start_by_centroids) {
    double cs[3], ct[3];
    orc_mean3(source, ns, cs);
    orc_mean3(target, nt, ct);
    for (int c = 0; c < 3; c++) T[c] = ct[c] - cs[c];
  }
  std::vector<double> P(3 * nb);
  int it = 0;
  double md = 0;
  for (;;) {
    orc_trans_point(a.data(), nb, R, T, P.data());
    double s[16], R1[9], T1[3];
    orc_icp_sums(target, nt, P.data(), nb, s);
    int rc = orc_horn_from_sums(s, nb, R1, T1);
    if (rc) return rc;
    double tR[9], tT[3];
    mul33(R1, R, tR);
    mul31(R1, T, tT);
    std::memcpy(R, tR, sizeof(tR));
    for (int c = 0; c < 3; c++) T[c] = tT[c] + T1[c];
    md = std::sqrt(s[15] / (double)nb);  // RMS landmark-to-closest distance before this round's update
    it++;
    if (it >= max_iter) break;
  }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) M[4 * r + c] = R[3 * r + c];
    M[4 * r + 3] = T[r];
  }
  M[12] = M[13] = M[14] = 0;
  M[15] = 1;
  if (mean_dist) *mean_dist = md;
  if (iters_o) *iters_o = it;
  return ORC_OK;
}

// MainForm.refreshClusList (FrmMain.cs:3437-3467): per raw point the nearest truth point
// with sqrt((tmp_X - motor_x)^2 + (tmp_Y - motor_y)^2) < radius; OrderByDescending(DISTANCE).Reverse() makes the
// LAST truth in list order win among equal distances; id = that truth's clusterId, 0 = none ("yedian").
int orc_assign_truths(const double* motor, int64_t n, const double* truths_xy, const int32_t* truth_ids, int32_t T,
                      double radius, int32_t* ids, int64_t* outliers) {
  if (n < 0 || T < 0) return ORC_ERR_ARG;
  int64_t out = 0;
  for (int64_t i = 0; i < n; i++) {
    int32_t id = 0;
    double best = 0;
    bool have = false;
    for (int32_t s = 0; s < T; s++) {
      double ax = truths_xy[2 * s] - motor[2 * i], ay = truths_xy[2 * s + 1] - motor[2 * i + 1];
      double d = std::sqrt(ax * ax + ay * ay);
      if (d < radius && (!have || d <= best)) {  // <= : later entries replace equal ones
        best = d;
        id = truth_ids[s];
        have = true;
      }
    }
    ids[i] = id;  // FirstOrDefault() of an empty sequence is 0; a truth with clusterId 0 also counts as none
    out += id == 0;
  }
  if (outliers) *outliers = out;
  return ORC_OK;
}

int orc_match(const double* centers, int32_t K, const double* truths, int32_t T, const double M[16],
              double max_dist, double* mxyz, uint8_t* is_matched, int32_t* nearest, double* nearest_dist,
              int32_t* count_matched) {
  if (K < 0 || T < 0) return ORC_ERR_ARG;
  if (T == 0 && K > 0) return ORC_ERR_EMPTY;  // truePointCloud.GetPoint(0) FrmMain.cs:3598
  int32_t cnt = 0;
  for (int32_t j = 0; j < K; j++) {
    const double* c = centers + 3 * j;
    // FrmMain.cs:3576-3584
    double m[3];
    for (int r = 0; r < 3; r++)
      m[r] = c[0] * M[4 * r + 0] + c[1] * M[4 * r + 1] + c[2] * M[4 * r + 2] + M[4 * r + 3];
    if (mxyz) {
      mxyz[3 * j] = m[0];
      mxyz[3 * j + 1] = m[1];
      mxyz[3 * j + 2] = m[2];
    }
    // :3594-3614 with getDisP :829-835
    auto dist = [&](int32_t i) {
      double dx = truths[3 * i] - m[0], dy = truths[3 * i + 1] - m[1], dz = truths[3 * i + 2] - m[2];
      return std::sqrt(dx * dx + dy * dy + dz * dz);
    };
    int32_t matchedId = 0;
    double c2t = dist(0);
    for (int32_t i = 0; i < T; i++) {
      double ddd = dist(i);
      if (ddd < c2t) {
        c2t = ddd;
        matchedId = i;
      }
    }
    nearest[j] = matchedId;
    if (nearest_dist) nearest_dist[j] = c2t;
    is_matched[j] = c2t < max_dist;
    cnt += is_matched[j];
  }
  if (count_matched) *count_matched = cnt;
  return ORC_OK;
}

}  // extern "C"
