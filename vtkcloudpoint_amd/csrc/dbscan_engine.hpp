// dbscan_engine.hpp -- internal interface of the DBSCAN engine (dbscan.hip), shared with blocks.hip.
#pragma once
#include "vcp_ctx.hpp"

// Optional grouping: points are neighbours only inside the same group (the reference's per-block
// DBImproved instances, FrmMain.cs:2782-2794).  `ord` replaces the original index as "position in the
// list" for the canonical numbering; it must be a permutation of [0,n) that is group-major, with
// groupstart[g] the first ord of group g.  Cluster ids restart at cf_in+1 in every group.
struct DbscanExt {
  const int32_t* d_group = nullptr;        // [n] by original index; < 0 = point excluded entirely
  const uint32_t* d_ord = nullptr;         // [n] by original index (NULL = the index itself)
  const uint32_t* d_groupstart = nullptr;  // [G+1]
  int32_t G = 0;
  int32_t only_lo = 0, only_hi = -1;       // cluster only groups lo <= g < hi (-1 = G)
  uint32_t* d_group_twice = nullptr;       // [G] zeroed by the engine; border points queried twice
  uint32_t* d_group_nclus = nullptr;       // [G] clusters found per group
  uint64_t* d_group_evals = nullptr;       // [1] sum over clustered groups of n_g*(n_g + K_g + twice_g)
  // staged (slab) call: stop after the component build, leave the grid state in the context and write
  // per point (caller order) the smallest ord of its component (0xFFFFFFFF if it is not expanding)
  bool slab = false;
  uint32_t* d_slab_rep = nullptr;
  // a box the caller already knows to contain every (finite) input point -- {min x, y, z, max x, y, z}: the bounds pass
  // and its host round trip are skipped (any box is correct: cell indices are clamped; a tight one gives the best grid)
  const double* h_bbox = nullptr;
  // grouped calls: groups of at most this many points are somebody else's (their points carry group -1): no statistics
  // are written for them
  uint32_t skip_upto = 0;
  // the caller knows that the box holds no far outliers (points spread over it): when the eps-grid over the box exceeds the
  // cell budget, coarsen at once instead of first trying to trim the range to mean +- 8 sigma (a pass + a host round trip)
  bool no_trim = false;
};

// d_* are device pointers; cf_out / dist_evals host pointers (may be null).  stride = doubles per point.
int vcp_dbscan_engine(vcp_ctx* ctx, const double* d_coords, int64_t n, int stride, int metric, double eps,
                      int min_pts, int32_t cf_in, const uint8_t* d_in_classed, int32_t* d_labels,
                      uint8_t* d_is_core, uint8_t* d_is_classed, int32_t* cf_out, int64_t* dist_evals,
                      const DbscanExt* ext);

// The dead v1.0 class DB (BaseClass/DB.cs:14-115), csrc/dbdead.hip: signed-sum metric on (X, Y), ifShown mask.
int vcp_db_engine(vcp_ctx* ctx, const double* d_coords, int64_t n, int stride, double eps, int min_pts, int32_t cf_in,
                  const uint8_t* d_mask, const uint8_t* d_in_classed, int32_t* d_labels, uint8_t* d_is_core,
                  uint8_t* d_is_classed, int32_t* cf_out, int64_t* dist_evals);

// The same class pair by pair (csrc/dbpairs.hip): exact for every input (non-finite coordinates, e < 0, clouds whose
// signed-sum relation is not provably 1-D), O(n^2), n <= 2^21; vcp_db_engine falls back to it.
int vcp_db_pairs_engine(vcp_ctx* ctx, const double* d_coords, int64_t n, int stride, double eps, int min_pts, int32_t cf_in,
                        const uint8_t* d_mask, const uint8_t* d_in_classed, int32_t* d_labels, uint8_t* d_is_core,
                        uint8_t* d_is_classed, int32_t* cf_out, int64_t* dist_evals);
