/*
 * vcp.h -- C-ABI of libvcp.so: MI355X (gfx950) DBSCAN + centroid + ICP hot path of
 * ZhiHuangHn/vtkCloudPoint.  Plain pointers and sizes only; no C++/torch types.
 *
 * Each entry point replaces one piece of the reference's C# class surface; the citation
 * (file:line under /root/reference/vtkPointCloud/, BC = BaseClass) names what it stands in
 * for.  The C# side binds these with [DllImport("vcp")] -- see INTEGRATION.md.
 *
 * Conventions
 *  - every call is blocking and re-entrant per context (one vcp_ctx per caller thread, the
 *    way FrmMain.cs:1358 runs one DBImproved per pool thread); the library keeps no pointer
 *    after a call returns;
 *  - return value: 0 = VCP_OK, < 0 = error (vcp_last_error(ctx) gives the text);
 *  - "host" entry points take caller-owned host buffers and do H2D/D2H themselves;
 *    "_dev" entry points take device pointers (inputs already resident in HBM) and run on
 *    the context's stream;
 *  - there is NO CPU fallback: without a HIP device every compute call fails with
 *    VCP_ERR_NO_DEVICE.
 */
#ifndef VCP_H
#define VCP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCP_VERSION_MAJOR 0
#define VCP_VERSION_MINOR 1

typedef struct vcp_ctx vcp_ctx;

/* distance forms: BC/DBImproved.cs:21 (live, |dx|+|dy| on motor_x/motor_y), :24 and :20
 * (commented-out Euclidean 2-D / 3-D), BC/DB.cs:21 (dead v1.0 class, signed dx+dy). */
enum vcp_metric { VCP_L1_2D = 0, VCP_L2_2D = 1, VCP_L2_3D = 2, VCP_SIGNED_SUM_2D = 3 };

/* ICP stop rules: BC/ICP.cs:180 (|SSE - previous SSE| < e) or RMSE = sqrt(SSE/Nd) < e. */
enum vcp_icp_stop { VCP_STOP_SSE_DELTA = 0, VCP_STOP_RMSE = 1 };

enum vcp_status {
  VCP_OK = 0,
  VCP_ERR_ARG = -1,          /* invalid argument                                           */
  VCP_ERR_EMPTY = -2,        /* the C# would throw on an empty collection                  */
  VCP_ERR_DEGENERATE = -3,   /* zero-extent first block: rows/cols undefined               */
  VCP_ERR_INDEX = -4,        /* the C# would throw IndexOutOfRange / ArgumentOutOfRange    */
  VCP_ERR_TOO_LARGE = -5,    /* n or grid beyond 32-bit indexing                           */
  VCP_ERR_NO_DEVICE = -6,    /* no HIP device / device id out of range                     */
  VCP_ERR_HIP = -7,          /* HIP runtime error                                          */
  VCP_ERR_UNSUPPORTED = -8,  /* valid request this build does not run on the GPU           */
  VCP_ERR_NOMEM = -9
};

/* -- context ------------------------------------------------------------------------------ */
/* device_id: HIP ordinal (one context drives one GPU; several GPUs: vcp_create_multi below, or one process per GPU). */
int vcp_create(int device_id, vcp_ctx** out);
void vcp_destroy(vcp_ctx* ctx);
const char* vcp_last_error(const vcp_ctx* ctx); /* ctx may be NULL: last create error      */
int vcp_version(void);                           /* major*1000 + minor                      */
/* Run on a caller-provided hipStream_t (NULL = the context's own stream). */
int vcp_set_stream(vcp_ctx* ctx, void* hip_stream);
/* Pinned-free device allocation helpers for hosts without a HIP binding of their own. */
int vcp_dev_alloc(vcp_ctx* ctx, uint64_t bytes, void** dptr);
int vcp_dev_free(vcp_ctx* ctx, void* dptr);
int vcp_h2d(vcp_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int vcp_d2h(vcp_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);

/* The context keeps its device workspace between calls (no hipMalloc on the steady-state path; ~100 bytes per
 * point of the largest call so far).  This frees it -- and the staged block / slab state -- without destroying the
 * context; the next call allocates again. */
int vcp_release_workspace(vcp_ctx* ctx);

/* Self-test of the library's device prefix scan (every pipeline stage places its output with it): exclusive sum
 * (op 0) or exclusive running maximum (op 1) of d_in [n] u32 into d_out [n] (d_out == d_in allowed), grand total to
 * *total.  Device pointers; returns when the result is in place. */
int vcp_selftest_scan_dev(vcp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n, int op, uint32_t* total);

/* Self-test of the Horn step of vcp_icp (BC/ICP.cs:53-124, intended arithmetic), run on the HOST from the same source
 * the device executes per round: sums[16] as vcp_icp_sums returns them, nd data points.  use_v != 0: V [16] is the
 * eigenvector basis of a previous round (warm start; a basis that is not orthonormal to 1e-9 -- NaN included -- is
 * replaced by the identity) and receives the basis found.  Returns 1 (solved), 0 (degenerate) or VCP_ERR_ARG.
 * Needs no device and no context. */
int vcp_selftest_horn(const double sums[16], int64_t nd, double V[16], int use_v, double R1[9], double T1[3]);

/* -- per-phase device timing (hipEvents on the launch stream) ----------------------------- */
/* When enabled, every compute call records hipEvents around each kernel phase on the stream
 * it launches on.  vcp_timing_get returns the phases of the LAST call. */
int vcp_timing_enable(vcp_ctx* ctx, int on);
int vcp_timing_count(vcp_ctx* ctx);
int vcp_timing_get(vcp_ctx* ctx, int i, const char** name, float* ms);

/* -- DBSCAN -------------------------------------------------------------------------------
 * Replaces DBImproved.dbscan(List<Point3D> lst, double e, int minPts), BC/DBImproved.cs:91-114
 * (with isKeyPoint :33-54 and expandCluster :56-90), call sites FrmMain.cs:1507-1516,
 * :2785-2786, BC/Tools.cs:591-592.  Exact semantics:
 *   core[i]      <=> #{j : d(i,j) <= eps} >= min_pts (the count includes i)
 *   expanding[i] <=> core[i] and not in_classed[i]
 *   clusters      = connected components of the expanding points under d <= eps, numbered
 *                   cf_in+1, cf_in+2, ... by increasing smallest member index
 *   any other point within eps of an expanding point takes the LARGEST such cluster id
 *                   (BC/DBImproved.cs:87 relabels unconditionally); otherwise it is untouched
 * coords     [n*dim] doubles, point-major (x,y[,z]); dim 2 or 3; the 2-D metrics read x,y
 * cf_in      DBImproved.cf before the call (FrmMain.cs:1509 presets it)
 * in_mask    ifShown filter of BC/DB.cs:40,63,98 -- only with VCP_SIGNED_SUM_2D, the dead v1.0 class DB
 *            (BC/DB.cs:14-115, FrmMain.cs:38): signed distance dx + dy on (x, y), clusters numbered from cf_in + 1,
 *            dist_evals = DB.iritatorNum, DB.pointsAmount = the number of shown points.  Sorts and scans where the
 *            class's floating-point relation is provably the 1-D relation on x + y (eps >= 0, finite coordinates, a
 *            common binary grid or no pair within rounding of the threshold: csrc/dbdead.hip); every other input --
 *            eps < 0 or NaN, non-finite coordinates included -- pair by pair, O(n^2), up to 2^21 points
 *            (csrc/dbpairs.hip; VCP_ERR_UNSUPPORTED beyond)
 * in_classed NULL = nobody classed and labels start at 0 (what every caller sets up,
 *            FrmMain.cs:1219-1223, :1512-1515); else Point3D.isClassed on entry and `labels`
 *            is read as Point3D.clusterId on entry
 * labels     [n] out (in/out with in_classed): Point3D.clusterId
 * is_core    [n] out, may be NULL: isKeyPoint flags SET by this call (core and queried)
 * is_classed [n] out, may be NULL: Point3D.isClassed after the call
 * cf_out     DBImproved.cf / clusterAmount after the call
 * dist_evals DBImproved.iritatorNum increment as a 64-bit count (BC/DBImproved.cs:12,19):
 *            what the O(n^2) C# would have evaluated, not what the GPU evaluates
 */
int vcp_dbscan(vcp_ctx* ctx, const double* coords, int64_t n, int dim, int metric, double eps,
               int min_pts, int32_t cf_in, const uint8_t* in_mask, const uint8_t* in_classed,
               int32_t* labels, uint8_t* is_core, uint8_t* is_classed, int32_t* cf_out,
               int64_t* dist_evals);
/* Same with device pointers; cf_out / dist_evals stay host pointers (16 bytes read back). */
int vcp_dbscan_dev(vcp_ctx* ctx, const double* d_coords, int64_t n, int dim, int metric, double eps,
                   int min_pts, int32_t cf_in, const uint8_t* d_in_classed, int32_t* d_labels,
                   uint8_t* d_is_core, uint8_t* d_is_classed, int32_t* cf_out, int64_t* dist_evals);

/* -- block-partitioned DBSCAN ("v2.0 multithread") ----------------------------------------
 * Replaces MainForm.getClusterFromMotor (FrmMain.cs:1214-1291: sort, first-block size, (lo,hi]
 * rectangle blocks via Tools.getListByScale2 BC/Tools.cs:510-513), StartCode (:2782-2794: one
 * DBImproved per block) and CompleteWork3 (:1442-1520: renumber, demote clusters of
 * <= small_max points, one global DBImproved over all noise with cf preset).
 * motor [n*2]; labels [n] by original index (0 = noise or dropped); block_of [n] may be NULL
 * (-1 = in no block); merge_order [n] may be NULL: original indices in final clusForMerge
 * order, *m_out entries.  kept = clusters surviving the demotion, del_sum = demoted clusters,
 * cluster_amount = DBImproved.clusterAmount after the noise pass (FrmMain.cs:1521-1522).
 * Limits: rows * cols <= 2^26 - 4 blocks (VCP_ERR_TOO_LARGE beyond: the C#'s cells[,] would be a 4 GB array of list
 * references there); a first block of zero extent = VCP_ERR_DEGENERATE (the C# divides by it, :1256-1259). */
int vcp_dbscan_blocks(vcp_ctx* ctx, const double* motor, int64_t n, double eps, int min_pts,
                      int pts_in_cell, int small_max, int32_t* labels, int32_t* block_of,
                      int64_t* merge_order, int64_t* m_out, int32_t* rows, int32_t* cols,
                      int32_t* kept, int32_t* del_sum, int32_t* cluster_amount, int64_t* dist_evals);

/* The 3-D twin MainForm.getClusterFromList (FrmMain.cs:1136-1213 with Tools.getListByScale, BC/Tools.cs:507-509): the
 * same pipeline, but the PARTITION (bounds, sort key, first-block size, (lo,hi] rectangles) reads key_xy [n*2] = (X, Y)
 * while every DBImproved -- per block and the noise pass -- still clusters on motor [n*2] (StartCode :2785-2786).
 * key_xy == NULL is vcp_dbscan_blocks. */
int vcp_dbscan_blocks_keyed(vcp_ctx* ctx, const double* key_xy, const double* motor, int64_t n, double eps,
                            int min_pts, int pts_in_cell, int small_max, int32_t* labels, int32_t* block_of,
                            int64_t* merge_order, int64_t* m_out, int32_t* rows, int32_t* cols, int32_t* kept,
                            int32_t* del_sum, int32_t* cluster_amount, int64_t* dist_evals);

/* The same pipeline in three stages, for sharding the per-block step over several GPUs (one
 * process and one context per GPU, every rank holds the whole cloud; SURVEY.md 8e mode 1):
 *   begin    partition (deterministic, identical on every rank); *m = points that fell in a block
 *   share    contiguous block range of `rank`, balanced on point count, and the matching
 *            [pos_lo, pos_hi) slice of the block-major label array
 *   cluster  DBImproved per block for block_lo <= b < block_hi; writes the block-local cluster ids
 *            (cf starts at 0 in every block, FrmMain.cs:2785) into d_local[pos_lo..pos_hi)
 *   -- the caller all-gathers the slices of d_local (RCCL) and sums the per-rank evals --
 *   finish   CompleteWork3 on the full d_local [m]; d_* outputs are device pointers
 * The state lives in the context until the next begin.  The _dev forms of begin read d_motor (and d_key_xy) in
 * place: the caller keeps those arrays valid and unchanged until the finish stage has returned. */
int vcp_blocks_begin(vcp_ctx* ctx, const double* motor, int64_t n, double eps, int min_pts,
                     int pts_in_cell, int small_max, int32_t* rows, int32_t* cols, int64_t* nblocks,
                     int64_t* m);
int vcp_blocks_begin_dev(vcp_ctx* ctx, const double* d_motor, int64_t n, double eps, int min_pts,
                         int pts_in_cell, int small_max, int32_t* rows, int32_t* cols,
                         int64_t* nblocks, int64_t* m);
/* begin with separate partition keys (getClusterFromList); share / cluster / finish are unchanged */
int vcp_blocks_begin_keyed(vcp_ctx* ctx, const double* key_xy, const double* motor, int64_t n, double eps,
                           int min_pts, int pts_in_cell, int small_max, int32_t* rows, int32_t* cols,
                           int64_t* nblocks, int64_t* m);
int vcp_blocks_begin_keyed_dev(vcp_ctx* ctx, const double* d_key_xy, const double* d_motor, int64_t n, double eps,
                               int min_pts, int pts_in_cell, int small_max, int32_t* rows, int32_t* cols,
                               int64_t* nblocks, int64_t* m);
int vcp_blocks_share(vcp_ctx* ctx, int rank, int world, int32_t* block_lo, int32_t* block_hi,
                     int64_t* pos_lo, int64_t* pos_hi);
int vcp_blocks_cluster_dev(vcp_ctx* ctx, int32_t block_lo, int32_t block_hi, int32_t* d_local,
                           int64_t* evals);
int vcp_blocks_finish_dev(vcp_ctx* ctx, const int32_t* d_local, int64_t evals_blocks,
                          int32_t* d_labels, int32_t* d_block_of, int64_t* d_merge_order,
                          int64_t* m_out, int32_t* kept, int32_t* del_sum, int32_t* cluster_amount,
                          int64_t* dist_evals);

/* The same pipeline with EVERY stage sharded (one process and one context per GPU; every rank holds the cloud, or at
 * least can read it): the ranks repeat only the streaming passes that decide the partition, build, cluster and merge
 * their own share of the blocks, and exchange a few words plus the final (index, label) pairs -- driver:
 * vtkcloudpoint_amd/distributed.py: sharded_pipeline; result = vcp_dbscan_blocks, bit for bit.
 *   plan      bounds, first block (FrmMain.cs:1224-1258), block of every point and the population of every SUPER-BUCKET
 *             (2^k consecutive block ids, *nsuper of them): identical on every rank
 *   cuts      cuts [world + 1]: first super-bucket of every rank, balanced on the point count (host arithmetic)
 *   build     the block-major list of the super-buckets [super_lo, super_hi) only (:1259-1285): blocks
 *             [*block_lo, *block_hi), *m_loc points in them, *n_loc points in all (the last share also holds the
 *             points in no block)
 *   cluster   vcp_blocks_cluster_dev on [*block_lo, *block_hi), d_local [m_loc]
 *   local     CompleteWork3 inside the share (:1442-1504): info = {clusters, kept ones, error (the C# would throw),
 *             request (the demotion quirk of :1485-1488 reaches the last entry of an EARLIER share), the share has a
 *             non-empty block, its last entry still carries a label, m_loc, n_loc}
 *   -- exchange: kept offsets; whose last entry is zeroed (the nearest earlier share with a non-empty block) --
 *   zero      zero_last != 0: zero that entry; the zero list of the share (:1510-1515): *z_count points, of which
 *             *active_count can be reached by the noise pass at all -- those within 2 eps of their block's boundary and
 *             those that lost a label (csrc/blocks.hip: k_zero_flag); everybody else provably keeps 0
 *   zcoords   the active points' coordinates [active_count * 2] in zero-list order (swap_xy: as (y, x); shares are bands
 *             in y)
 *   -- the global noise pass over all shares' active points: vcp_slab_* (exact DBSCAN over several GPUs), cf preset;
 *      DBImproved.iritatorNum of the pass = Z x (Z + clusters + border points queried twice), Z = sum of z_count --
 *   pairs     d_pairs [n_loc] = (original index << 32 | final label): kept clusters + kept_offset, active points with
 *             d_zlab [active_count] (their labels from the noise pass), everybody else 0
 *   scatter   d_labels[index] = label for count pairs (any rank's), indices < n */
int vcp_blocks_plan_dev(vcp_ctx* ctx, const double* d_key_xy, const double* d_motor, int64_t n, double eps, int min_pts,
                        int pts_in_cell, int small_max, int32_t* rows, int32_t* cols, int64_t* nblocks, int64_t* nsuper);
int vcp_blocks_plan_cuts(vcp_ctx* ctx, int world, int64_t* cuts);
int vcp_blocks_build_dev(vcp_ctx* ctx, int64_t super_lo, int64_t super_hi, int32_t* block_lo, int32_t* block_hi,
                         int64_t* m_loc, int64_t* n_loc);
int vcp_blocks_finish_local_dev(vcp_ctx* ctx, const int32_t* d_local, int64_t info[8]);
int vcp_blocks_finish_zero_dev(vcp_ctx* ctx, int zero_last, int64_t* z_count, int64_t* active_count);
int vcp_blocks_finish_zcoords_dev(vcp_ctx* ctx, int swap_xy, double* d_zcoords);
int vcp_blocks_finish_pairs_dev(vcp_ctx* ctx, int32_t kept_offset, const int32_t* d_zlab, int64_t* d_pairs);
int vcp_scatter_pairs_dev(vcp_ctx* ctx, const int64_t* d_pairs, int64_t count, int64_t n, int32_t* d_labels);

/* -- several GPUs from ONE process -----------------------------------------------------------
 * The reference fans its blocks out from one process (ThreadPool.QueueUserWorkItem(StartCode, cells[i]) per block,
 * FrmMain.cs:1356-1359) and merges on the UI thread (CompleteWork3, :1442-1520).  vcp_multi is the drop-in form of
 * that for a host that owns several GPUs: one vcp_ctx per listed HIP device (a device id may repeat: several contexts
 * on one GPU), driven by one host thread each.  vcp_dbscan_blocks_multi = vcp_dbscan_blocks_keyed (key_xy may be NULL)
 * with every stage sharded (round 3): each device repeats the streaming passes that decide the partition, then builds,
 * clusters and merges its own share of the blocks (the vcp_blocks_plan_dev ... stages below); the shares' counters meet
 * in this process's memory, the active points of the zero lists travel to device 0 as peer copies over xGMI, device 0
 * runs the global noise pass over them and assembles the label array from the devices' (index, label) pairs.  Results
 * are identical to the one-device call, bit for bit.  (The multi-PROCESS form -- one rank per GPU, RCCL all-gathers -- is
 * vtkcloudpoint_amd/distributed.py: sharded_pipeline, over the same stages.) */
typedef struct vcp_multi vcp_multi;
int vcp_create_multi(const int* device_ids, int n, vcp_multi** out);
void vcp_destroy_multi(vcp_multi* m);
const char* vcp_multi_last_error(const vcp_multi* m); /* m may be NULL: last create error */
int vcp_multi_count(const vcp_multi* m);
vcp_ctx* vcp_multi_ctx(vcp_multi* m, int i);          /* borrowed: context of device i for single-GPU calls */
int vcp_dbscan_blocks_multi(vcp_multi* m, const double* key_xy, const double* motor, int64_t n, double eps, int min_pts,
                            int pts_in_cell, int small_max, int32_t* labels, int32_t* block_of, int64_t* merge_order,
                            int64_t* m_out, int32_t* rows, int32_t* cols, int32_t* kept, int32_t* del_sum,
                            int32_t* cluster_amount, int64_t* dist_evals);
/* The range arithmetic of vcp_blocks_share and vcp_dbscan_blocks_multi as a pure host function (no device, no
 * context): blockstart [nblocks + 1] = first block-major position of every block (ascending, blockstart[nblocks] = m);
 * cuts [world + 1] <- first block of every rank (cuts[0] = 0, cuts[world] = nblocks): rank r starts at the first block
 * whose first position is >= m * r / world. */
int vcp_blocks_share_plan(const uint32_t* blockstart, int64_t nblocks, int world, int64_t* cuts);

/* -- exact DBSCAN over several GPUs (SURVEY.md 8e mode 2) -----------------------------------
 * One monolithic DBImproved.dbscan (BC/DBImproved.cs:91-114) over a cloud that is spread over several
 * ranks (one process and one context per GPU): every rank clusters its own points plus a 2*eps halo of
 * the other ranks' points, the ranks agree on the components that cross a boundary, and each labels its
 * own points with the ids the single-GPU call would have produced (vtkcloudpoint_amd/distributed.py:
 * exact_slabs drives the exchange over RCCL).
 *   begin    d_coords [n*dim] = own points followed by halo points; d_ord [n] = position of each point in
 *            the GLOBAL list (what "index in lst" means for the seed / numbering rules); d_noexpand [n]
 *            (may be NULL) = 1 for points whose neighbourhood is incomplete here: they count as
 *            neighbours but never seed or extend a cluster.  Builds the grid, the core flags and the
 *            LOCAL components; writes d_rep [n] = smallest ord of the point's local component
 *            (0xFFFFFFFF for points that are not expanding) and d_is_core [n] (may be NULL);
 *            *n_comp = number of local components.
 *   comps    copies the n_comp component seeds (their smallest ord, in no particular order) to the host.
 *   finish   map_rep [n_comp] strictly ascending = the component seeds; map_k [n_comp] = index of the
 *            component's GLOBAL cluster in the table tab_gid / tab_seed [n_tab] (cluster id, strictly
 *            ascending, and the global seed position of that cluster); all four are host arrays.  Applies
 *            the border rule (:87, largest adjacent id) and writes d_labels [n], d_is_classed [n] (may be
 *            NULL); *twice = own points (own_lo <= ord < own_lo + own_count) that the C# loop queries a
 *            second time (the iritatorNum term).
 * The state lives in the context's workspace: no other call on this context between begin and finish, and d_coords
 * (read in place by the exact re-tests of the finish stage too) stays valid and unchanged until finish has returned. */
int vcp_slab_begin(vcp_ctx* ctx, const double* d_coords, int64_t n, int dim, int metric, double eps,
                   int min_pts, const uint8_t* d_noexpand, const uint32_t* d_ord, uint32_t* d_rep,
                   uint8_t* d_is_core, int64_t* n_comp);
int vcp_slab_comps(vcp_ctx* ctx, uint32_t* comp_rep);
int vcp_slab_finish(vcp_ctx* ctx, const uint32_t* map_rep, const uint32_t* map_k, int64_t n_tab,
                    const int32_t* tab_gid, const uint32_t* tab_seed, uint32_t own_lo, uint32_t own_count,
                    int32_t* d_labels, uint8_t* d_is_classed, int64_t* twice);

/* -- centroids ----------------------------------------------------------------------------
 * Replaces Tools.GetClusList (BC/Tools.cs:162-195; also getClusterCenter :118-155): per cluster
 * id 1..K the mean of (X,Y,Z) -> c3 [K*3] and of (motor_x,motor_y) -> c2 [K*2]; counts [K].
 * xyz or motor may be NULL (then c3 / c2 is not written).  Empty clusters: count 0, NaN rows.
 * Sums are fixed-order binary64 tree reductions (run-to-run deterministic; they differ from the
 * C#'s sequential sum in the last bits -- tolerance 1e-12 relative). */
int vcp_centroids(vcp_ctx* ctx, const double* xyz, const double* motor, const int32_t* labels,
                  int64_t n, int32_t K, double* c3, double* c2, int64_t* counts);
int vcp_centroids_dev(vcp_ctx* ctx, const double* d_xyz, const double* d_motor, const int32_t* d_labels,
                      int64_t n, int32_t K, double* d_c3, double* d_c2, int64_t* d_counts);

/* Replaces Tools.getFixedPtsCentroid(clusList, isIgnoreDuplication) (BC/Tools.cs:78-111; caller
 * SureDistanceFilter.cs:74): per list 1..K the ptsCount-weighted mean of (X,Y,Z).  group [n] = which ClusObj.li the
 * point sits in (1..K, 0 = none); cluster_id [n] = Point3D.clusterId (NULL = group); pts_count [n] = Point3D.ptsCount
 * (FrmMain.cs:1074,1081: multiplicity of a deduplicated fixed point).  A member with clusterId != 0 counts ONCE when
 * ignore_duplication is set, otherwise ptsCount times (:88-101).  c3 [K*3]; inside_num [K] (may be NULL) = the C#'s
 * insideNum (0 gives NaN rows, 0/0).  An empty list makes the C# throw at li[0] (:106): VCP_ERR_INDEX.
 * Sums are the fixed tree of vcp_centroids (1e-12 relative to the C#'s sequential sum). */
int vcp_centroids_weighted(vcp_ctx* ctx, const double* xyz, const int32_t* group, const int32_t* cluster_id,
                           const int32_t* pts_count, int64_t n, int32_t K, int ignore_duplication, double* c3,
                           int64_t* inside_num);

/* Replaces Tools.MergeIDByDistance (BC/Tools.cs:580-621): DBImproved(minPts 2, L1 on X,Y) over the
 * K centroids; map_to[k] = cluster id the k-th centroid's cluster is merged into, 0 = none. */
int vcp_merge_centroids(vcp_ctx* ctx, const double* cxy, const int32_t* ids, int32_t K, double thr,
                        int32_t* map_to, int32_t* merge_count);
/* Replaces Tools.refreshCensAndClusByDictionary (BC/Tools.cs:521-572): relabel points through
 * map_by_id (index id-1, 0 = keep), renumber surviving ids 1..K' ascending, recompute centroids. */
int vcp_refresh_by_dictionary(vcp_ctx* ctx, const double* xyz, const double* motor, int32_t* labels,
                              int64_t n, int32_t K, const int32_t* map_by_id, int32_t* new_k,
                              double* c3, double* c2, int64_t* counts);

/* -- ICP ------------------------------------------------------------------------------------
 * Replaces ICP.go_hell_ICP(model, data, R, T, e) (BC/ICP.cs:18-181): per round nearest model
 * point per data point (FindClosestPointSet :224-250, lowest index on ties), the 16 sums
 * (CalculateMeanPoint3D :255-273, sum p y^T :38-52, SSE :126-133), Horn's closed form on the
 * host (the INTENDED arithmetic of :53-124; the as-written code is non-functional, SURVEY.md
 * fact 4), composition R <- R1 R, T <- R1 T + T1 (:149-177), P <- R data + T (TransPoint
 * :195-219).  model [nm*3], data [nd*3]; R [9] row-major and T [3] are outputs. */
int vcp_icp(vcp_ctx* ctx, const double* model, int64_t nm, const double* data, int64_t nd, double tol,
            int max_iter, int stop_rule, double R[9], double T[3], double* sse, double* rmse,
            int32_t* iters);
int vcp_icp_dev(vcp_ctx* ctx, const double* d_model, int64_t nm, const double* d_data, int64_t nd,
                double tol, int max_iter, int stop_rule, double R[9], double T[3], double* sse,
                double* rmse, int32_t* iters);
/* One correspondence pass (A10+A11): sums[16] = sum p[3], sum y[3], sum p y^T[9], SSE for
 * p = R data + T; nn [nd] may be NULL.  R,T NULL = identity. */
int vcp_icp_sums(vcp_ctx* ctx, const double* model, int64_t nm, const double* data, int64_t nd,
                 const double R[9], const double T[3], double sums[16], int32_t* nn);

/* "VTK-like" ICP: the configuration MainForm.ICP() gives VTK's closed
 * vtkIterativeClosestPointTransform (FrmMain.cs:851-862), behaviour per the VTK 5.0 header
 * (vtkIterativeClosestPointTransform.h:49-180): landmarks = every step-th source point (step = ns /
 * max_landmarks when ns > max_landmarks; VTK's default cap is 200), optional start by matching centroids,
 * exactly max_iter rounds (the reference sets 100 and leaves the mean-distance check off), rigid body.
 * source [ns*3] (the reference feeds (tmp_X, tmp_Y, 0), Tools.cs:696-703), target [nt*3]; M = the accumulated
 * 4x4 row-major matrix (what icp.GetMatrix() returns, FrmMain.cs:862); mean_dist = RMS landmark-to-closest
 * distance seen by the last round.  PARITY UNPINNED against VTK itself (sources not in the reference tree). */
int vcp_icp_vtklike(vcp_ctx* ctx, const double* source, int64_t ns, const double* target, int64_t nt,
                    int max_iter, int max_landmarks, int start_by_matching_centroids, double M[16],
                    double* mean_dist, int32_t* iters);

/* -- minimal bounding circles ---------------------------------------------
 * Replaces Tools.getCircles (BC/Tools.cs:394-409) / Geometry.FindMinimalBoundingCircle
 * (BC/Geometry.cs:247-319; gift-wrap hull :122-208, circle through 2 or 3 hull points :260-312): for
 * every cluster 1..K with more than 3 points the smallest enclosing circle of its (x,y).
 * xy [n*2] = (X,Y) for the 3-D view or (motor_x,motor_y) for the 2-D one; labels [n]; order [m] = the
 * list order the C# iterates (clusForMerge; NULL = 0..n-1, then m must equal n): "first in the list"
 * decides every tie.  centers [K*2], radius [K], valid [K] (0 = cluster skipped), hull_n [K] may be NULL.
 * Bit-identical to the C# arithmetic (binary64, no FMA contraction). */
int vcp_mcc(vcp_ctx* ctx, const double* xy, const int32_t* labels, const int64_t* order, int64_t m,
            int64_t n, int32_t K, double* centers, double* radius, uint8_t* valid, int32_t* hull_n);

/* -- matching ------------------------------------------------------------------------------
 * Replaces MainForm.calMatchedCoords (FrmMain.cs:3572-3587) + RecorrectMatchingPtsByDistance
 * (:3588-3618, getDisP :829-835): matched = M * (c,1); nearest truth by Euclidean distance
 * (strict <, lowest index on ties); is_matched iff distance < max_dist. */
int vcp_match(vcp_ctx* ctx, const double* centers, int32_t K, const double* truths, int32_t T,
              const double M[16], double max_dist, double* matched_xyz, uint8_t* is_matched,
              int32_t* nearest, double* nearest_dist, int32_t* count_matched);

/* -- import conversion + duplicate removal ------------------------------------
 * Replaces the per-row work of MainForm.AddFolder for scan points (FrmMain.cs:1011-1090, typpe 1 / 2):
 * rows [n*3] = (motor_x, motor_y, Distance) as parsed from the tab-separated text (BC/FileMap.cs:16-33);
 * rows with Distance == 0 or > 1000 are filtered (:1011); X,Y,Z by the spherical conversion of :1025-1062 with
 * the zero angles x_angle, y_angle and the axis choices xdir, ydir (1 up, 2 right, 3 down, 4 left); with
 * dedupe != 0 a row whose (tmpx,tmpy,tmpz) equals that of an earlier kept row is a duplicate (:1063-1068; the
 * C#'s O(n^2) FindAll becomes a device hash table).  xyz [n*3] and state [n] (0 filtered, 1 kept, 2 duplicate)
 * are written for every row in input order.  dedupe needs the ImportPts defaults xdir = 2, ydir = 1. */
int vcp_import_convert(vcp_ctx* ctx, const double* rows, int64_t n, double x_angle, double y_angle, int xdir,
                       int ydir, int dedupe, double* xyz, uint8_t* state, int64_t* kept, int64_t* duplicates);

/* -- truth-guided assignment -------------------------------------------------
 * Replaces the query of MainForm.refreshClusList (FrmMain.cs:3437-3467): per raw point (motor_x, motor_y)
 * the nearest truth (tmp_X, tmp_Y) with Euclidean distance < radius; among equal distances the LAST truth
 * in list order wins (OrderByDescending(DISTANCE).Reverse()); ids[i] = that truth's clusterId, 0 = none;
 * *outliers = number of points with id 0 ("yedian"). */
int vcp_assign_truths(vcp_ctx* ctx, const double* motor, int64_t n, const double* truths_xy,
                      const int32_t* truth_ids, int32_t T, double radius, int32_t* ids, int64_t* outliers);

#ifdef __cplusplus
}
#endif
#endif
