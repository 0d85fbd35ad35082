/*
 * vcp_oracle.h -- CPU ORACLE for the vtkCloudPoint DBSCAN + centroid + ICP hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libvcp.so) never links,
 * loads or calls anything in oracle/.
 *
 * It is a C++17 restatement (single thread, strict IEEE-754 binary64, no FMA
 * contraction) of the reference's C# algorithms.  The reference cannot be compiled in
 * this image (no dotnet/mono/csc) and ships no tests, golden vectors or sample data
 * (SURVEY.md section 4, 8c), so PARITY IS UNPINNED BY THE REFERENCE: the oracle is
 * pinned instead by (i) a line-by-line literal transcription checked against an
 * independent order-free formulation on randomised inputs with exact ties,
 * (ii) hand-derived micro cases, (iii) scikit-learn's DBSCAN(metric='manhattan') for
 * the core set / core partition, (iv) closed-form ICP recoveries and numpy SVD.
 *
 * Citations are file:line into /root/reference/vtkPointCloud/ (BC = BaseClass).
 */
#ifndef VCP_ORACLE_H
#define VCP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Distance forms.  L1_2D is the live one (BC/DBImproved.cs:21); the two L2 forms are
 * the commented-out alternatives (BC/DBImproved.cs:20,24); SIGNED_SUM_2D is the dead
 * v1.0 class (BC/DB.cs:21). */
enum { ORC_L1_2D = 0, ORC_L2_2D = 1, ORC_L2_3D = 2, ORC_SIGNED_SUM_2D = 3 };
enum { ORC_STOP_SSE_DELTA = 0, ORC_STOP_RMSE = 1 };

enum {
  ORC_OK = 0,
  ORC_ERR_ARG = -1,          /* bad argument                                        */
  ORC_ERR_EMPTY = -2,        /* the reference would throw on an empty collection    */
  ORC_ERR_DEGENERATE = -3,   /* zero-extent first block => rows/cols undefined      */
  ORC_ERR_INDEX = -4,        /* the reference would throw an index-out-of-range     */
  ORC_ERR_TOO_LARGE = -5
};

/* BC/DBImproved.cs:14-114, line by line (same scans, same growth of the neighbour
 * list with duplicates).  classed/labels/is_key are in/out exactly like the fields
 * of Point3D: dbscan() never resets them (callers do, FrmMain.cs:1219-1223).
 * cf_in is DBImproved.cf before the call, *cf_out after (== clusterAmount).
 * *dist_evals is iritatorNum as a 64-bit counter.  run_dead_dedupe_scan != 0 also
 * executes the never-matching reference-equality scan of :70-83 (timing only; it
 * cannot change results). */
int orc_dbscan_literal(const double* coords, int64_t n, int dim, int metric, double eps,
                       int min_pts, int32_t cf_in, uint8_t* classed, int32_t* labels,
                       uint8_t* is_key, int32_t* cf_out, int64_t* dist_evals,
                       int run_dead_dedupe_scan);

/* Order-free formulation of the same semantics on a CPU grid:
 * O(n k).  Same in/out contract as orc_dbscan_literal.  Not valid for
 * ORC_SIGNED_SUM_2D (asymmetric relation). */
int orc_dbscan_canonical(const double* coords, int64_t n, int dim, int metric, double eps,
                         int min_pts, int32_t cf_in, uint8_t* classed, int32_t* labels,
                         uint8_t* is_key, int32_t* cf_out, int64_t* dist_evals);

/* Staged canonical formulation for a cloud spread over several ranks (SURVEY.md 8e mode 2; mirrors
 * vcp_slab_begin / vcp_slab_finish of include/vcp.h).  begin: core flags, local components of the
 * expanding points (core and not noexpand), rep[i] = smallest ord of the component of i or 0xFFFFFFFF.
 * finish: labels from the caller's table of global clusters (map_rep ascending -> map_k -> tab_gid /
 * tab_seed, tab_gid ascending), border rule BC/DBImproved.cs:87, `twice` over own points. */
int orc_slab_begin(const double* coords, int64_t n, int dim, int metric, double eps, int min_pts,
                   const uint8_t* noexpand, const uint32_t* ord, uint32_t* rep, uint8_t* is_core,
                   int64_t* n_comp);
int orc_slab_finish(const double* coords, int64_t n, int dim, int metric, double eps, const uint8_t* noexpand,
                    const uint32_t* ord, const uint32_t* rep, const uint32_t* map_rep, const uint32_t* map_k,
                    int64_t n_map, const int32_t* tab_gid, const uint32_t* tab_seed, int64_t n_tab, uint32_t own_lo,
                    uint32_t own_count, int32_t* labels, uint8_t* is_classed, int64_t* twice);

/* BC/DB.cs:14-115, line by line (metric dx+dy on X,Y; ifShown filter). */
int orc_db_literal(const double* coords, int64_t n, int dim, double eps, int min_pts,
                   const uint8_t* shown, uint8_t* classed, int32_t* labels, uint8_t* is_key,
                   int32_t* cluster_amount, int32_t* points_amount, int64_t* dist_evals);

/* Block-partitioned pipeline: FrmMain.cs:1214-1291 (partition), :2782-2794 (per-block
 * DBImproved), :1442-1520 (renumber, demotion, global noise pass).
 * motor = [n*2] (motor_x, motor_y).  Outputs are indexed by ORIGINAL point index:
 *   labels[n]     final clusterId (0 = noise or dropped)
 *   block_of[n]   block index, -1 = the point fell in no block (reference drops it)
 *   merge_order   original indices in final clusForMerge order, *m_out entries
 * Declared deviations from the C# (DESIGN.md): List.Sort's unstable tie order is
 * replaced by a stable sort (ties by original index / list position); a block-0 point
 * is never also filed under a rectangle; clusterSum is summed deterministically.
 * brute_partition != 0 uses the literal O(n*blocks) FindAll sweep. */
int orc_block_pipeline(const double* motor, int64_t n, double eps, int min_pts, int pts_in_cell,
                       int small_max, int use_canonical, int brute_partition, int32_t* labels,
                       int32_t* block_of, int64_t* merge_order, int64_t* m_out, int32_t* rows,
                       int32_t* cols, int32_t* kept, int32_t* del_sum, int32_t* cluster_amount,
                       int64_t* dist_evals);

int orc_block_pipeline_keyed(const double* key, const double* motor, int64_t n, double eps, int min_pts, int pts_in_cell,
                             int small_max, int use_canonical, int brute_partition, int32_t* labels,
                             int32_t* block_of, int64_t* merge_order, int64_t* m_out, int32_t* rows,
                             int32_t* cols, int32_t* kept, int32_t* del_sum, int32_t* cluster_amount,
                             int64_t* dist_evals);

/* The same pipeline in stages (used to check the staged multi-GPU path stage by stage):
 * partition  -> block_of [n], raw [n] (list order after the sort), bl [m] (block-major list of original
 *               indices), blockstart [rows*cols+1] (capacity blockstart_cap entries)
 * cluster    -> local [m] block-local ids for block_lo <= b < block_hi
 * finish     -> CompleteWork3 from the full `local` array */
int orc_block_partition(const double* motor, int64_t n, int pts_in_cell, int brute_partition, int32_t* block_of,
                        int64_t* raw, int64_t* bl, int64_t* blockstart, int64_t blockstart_cap, int32_t* rows,
                        int32_t* cols, int64_t* m);
int orc_block_cluster(const double* motor, const int64_t* bl, const int64_t* blockstart, int64_t block_lo,
                      int64_t block_hi, double eps, int min_pts, int use_canonical, int32_t* local,
                      int64_t* evals);
int orc_block_finish(const double* motor, int64_t n, const int64_t* bl, const int64_t* blockstart, int64_t nblocks,
                     const int32_t* local, double eps, int min_pts, int small_max, int use_canonical,
                     int64_t evals_blocks, int32_t* labels, int64_t* merge_order, int64_t* m_out, int32_t* kept,
                     int32_t* del_sum, int32_t* cluster_amount, int64_t* dist_evals);

/* Tools.GetClusList (BC/Tools.cs:162-195): per cluster id 1..K the LINQ Average
 * (sequential binary64 sum, then / count) of X,Y,Z and of motor_x,motor_y over the
 * points visited in `order` (NULL = 0..m-1).  Empty clusters: count 0, NaN rows. */
int orc_centroids(const double* xyz, const double* motor, const int32_t* labels,
                  const int64_t* order, int64_t m, int32_t K, double* c3, double* c2,
                  int64_t* counts);

/* Tools.getFixedPtsCentroid (BC/Tools.cs:78-111): ptsCount-weighted mean per list; see include/vcp.h. */
int orc_fixed_centroids(const double* xyz, const int32_t* group, const int32_t* cluster_id, const int32_t* pts_count,
                        int64_t n, int32_t K, int ignore_dup, double* c3, int64_t* inside_num);

/* Tools.MergeIDByDistance (BC/Tools.cs:580-621): DBImproved(minPts=2, L1 on X,Y) over
 * the K centroids; map_to[k] = id the k-th centroid's cluster is merged into, or 0. */
int orc_merge_ids(const double* cxy, const int32_t* ids, int32_t K, double thr, int32_t* map_to,
                  int32_t* merge_count);

/* Tools.refreshCensAndClusByDictionary (BC/Tools.cs:521-572).  map_by_id[id-1] = target
 * id or 0, for id in 1..K (clusList position == id-1, as the C# assumes).  Relabels
 * `labels` (visited in `order`), returns new cluster count and recomputed centroids
 * (c3 [K*3], c2 [K*2], counts [K], first *new_k rows valid). */
int orc_refresh_by_dictionary(const double* xyz, const double* motor, int32_t* labels,
                              const int64_t* order, int64_t m, int32_t K,
                              const int32_t* map_by_id, int32_t* new_k, double* c3, double* c2,
                              int64_t* counts);

/* ICP sub-functions, each individually correct in the C# (SURVEY.md 8a A10-A13). */
void orc_find_closest(const double* model, int64_t nm, const double* p, int64_t nd,
                      int32_t* idx);                                  /* BC/ICP.cs:224-250 */
void orc_mean3(const double* p, int64_t n, double mean[3]);           /* BC/ICP.cs:255-273 */
void orc_trans_point(const double* src, int64_t n, const double R[9], const double T[3],
                     double* dst);                                    /* BC/ICP.cs:195-219 */
void orc_calc_rotation(const double q[4], double R[9]);               /* BC/ICP.cs:274-285 */
/* One correspondence pass: 16 sums in list order (sum p[3], sum y[3], sum p y^T[9], SSE). */
void orc_icp_sums(const double* model, int64_t nm, const double* p, int64_t nd, double sums[16]);
/* Horn closed form from the 16 sums: the INTENDED arithmetic of BC/ICP.cs:53-124
 * (cov = S/N - muP muY^T; the as-written integer division, '+' sign, delta[2] index and
 * Jacobi indexing bugs are NOT reproduced -- SURVEY.md fact 4). */
int orc_horn_from_sums(const double sums[16], int64_t nd, double R1[9], double T1[3]);
/* Symmetric eigen-decomposition (classical max-pivot Jacobi, the intent of
 * BC/Matrix.cs:571-668), A is n*n row-major and is destroyed; V columns = vectors. */
int orc_jacobi_sym(double* A, int n, double* evals, double* V, int max_sweeps);

/* The ICP loop with the reference's structure, composition order and stop rule
 * (BC/ICP.cs:18-181).  R (3x3 row-major) and T are outputs (overwritten at round 1). */
int orc_icp(const double* model, int64_t nm, const double* data, int64_t nd, double tol,
            int max_iter, int stop_rule, double R[9], double T[3], double* sse, double* rmse,
            int32_t* iters);

/* Geometry.FindMinimalBoundingCircle (BC/Geometry.cs:247-319) on `cnt` points (x,y) in list order: gift-wrap
 * hull, then the smallest enclosing circle through 2 or 3 hull points, first found on ties.  hull_xy may be NULL. */
int orc_min_circle(const double* pts, int64_t cnt, double center[2], double* radius, double* hull_xy,
                   int64_t hull_cap, int32_t* hull_n);
/* Tools.getCircles (BC/Tools.cs:394-409): a circle for every cluster 1..K with more than 3 points (valid[k]). */
int orc_get_circles(const double* xy, const int32_t* labels, const int64_t* order, int64_t m, int32_t K,
                    double* centers, double* radius, uint8_t* valid, int32_t* hull_n);

/* MainForm.AddFolder's conversion + duplicate removal for scan points (FrmMain.cs:1011-1090). */
int orc_import_convert(const double* rows, int64_t n, double x_angle, double y_angle, int xdir, int ydir, int dedupe,
                       int literal, double* xyz, uint8_t* state, int64_t* kept, int64_t* duplicates);

/* "VTK-like" ICP: the configuration of FrmMain.ICP() (FrmMain.cs:851-862) per the VTK 5.0 header; unpinned. */
int orc_icp_vtklike(const double* source, int64_t ns, const double* target, int64_t nt, int max_iter,
                    int max_landmarks, int start_by_centroids, double M[16], double* mean_dist, int32_t* iters);

/* MainForm.refreshClusList (FrmMain.cs:3437-3467): truth-guided assignment of every raw point. */
int orc_assign_truths(const double* motor, int64_t n, const double* truths_xy, const int32_t* truth_ids, int32_t T,
                      double radius, int32_t* ids, int64_t* outliers);

/* calMatchedCoords + RecorrectMatchingPtsByDistance (FrmMain.cs:3572-3618, :829-835). */
int orc_match(const double* centers, int32_t K, const double* truths, int32_t T,
              const double M[16], double max_dist, double* matched_xyz, uint8_t* is_matched,
              int32_t* nearest, double* nearest_dist, int32_t* count_matched);

#ifdef __cplusplus
}
#endif
#endif
