// blocks_state.hpp -- state of the staged block-partitioned pipeline, shared by blockpart.hip (the partition:
// MainForm.getClusterFromMotor, FrmMain.cs:1214-1291) and blocks.hip (StartCode :2782-2794, CompleteWork3 :1442-1520).
#pragma once
#include <vector>

#include "dbscan_engine.hpp"

struct BlocksState {
  int64_t n = 0, m = 0;
  int32_t rows = 0, cols = 0;
  int64_t nblocks = 0;
  double eps = 0;
  int min_pts = 0, small_max = 3, take = 0;
  double x_Min = 0, x_Max = 0, y_Min = 0, y_Max = 0, cell_x = 0, cell_y = 0;
  double mbox[4] = {0, 0, 0, 0};      // bounding box of the motor coordinates (x min, x max, y min, y max): the engine's grid
  const double* motor_ptr = nullptr;  // the cloud on the device: our upload (host entry points) or the caller's array
  DevBuf motor, pkey, blockof, bl, motor_bm, blockstart, gtwice, gnclus, tmp0, tmp1, tmp2, tmp3, sorttmp, blk_t, csize,
      cstart, kb, zb, keep, order, newlab, zflag, zlist, zcoords, zlab, misc;
  // partition workspace (blockpart.hip)
  DevBuf sel, cand, counts, counts_t, rec, rec2, stage, rank, binfo, slicelist, vlist, fall, gcnt;
  // blocks of at most brute_thr points are clustered by the all-pairs kernel (blocks.hip: k_block_brute), the others by the
  // grid engine, which reads grp_big [m]: the block id of a point of a larger block, -1 for the others (0 = engine only)
  uint32_t brute_thr = 0;
  DevBuf grp_big, brutecnt;
  hipStream_t side = nullptr;  // the all-pairs kernels run here, beside the engine's launches on the context's stream
  hipEvent_t ev_fork = nullptr;
  bool virt_clean = false;  // gcnt is all zero
  // the plan (identical on every rank) ...
  bool planned = false, built = false;
  const double* d_key = nullptr;
  const double* d_motor = nullptr;
  unsigned long long key_T = 0;
  uint32_t idx_T = 0, fsh = 0, NS = 0, chunk = 0, nchunk = 0;
  DevBuf sbstart;
  std::vector<uint32_t> h_sbstart;  // [NS + 1] first point of every super-bucket in the list of all n (vcp_blocks_plan)
  // ... and the share this context built: super-buckets [S_lo, S_hi) = blocks [b_lo, b_hi), n_loc points of which m fell in
  // a block; positions in the block-major arrays are relative to the share
  uint32_t S_lo = 0, S_hi = 0;
  int64_t b_lo = 0, b_hi = 0, n_loc = 0;
  std::vector<uint32_t> h_blockstart;
  DevBuf biglist;  // blocks of more than BIG_BLOCK points, listed by the partition (device), in no particular order
  uint32_t nbig = 0;
  // blocks [0, cov_hi) have been clustered by this context, in adjoining ranges, and hold totalC_acc clusters
  bool cov_ok = true;
  int64_t cov_lo = 0, cov_hi = 0, totalC_acc = 0;
  // the finish stage's counters between its parts (blocks.hip: finish_local / finish_zero)
  bool f_by_sort = false;
  const int32_t* f_local = nullptr;
  uint32_t f_totalC = 0, f_kept = 0, f_err = 0, f_req = 0, f_nonempty = 0, f_last_nonzero = 0, f_Z = 0, f_A = 0;
  bool ready = false;
};

constexpr uint32_t VCP_BIG_BLOCK = 1024;  // blocks beyond this take the workgroup-per-block kernels
constexpr uint32_t VCP_BRUTE_MAX = 1024;  // largest block the all-pairs kernel takes (its LDS copy of the coordinates)

// ensure capacity of a state-owned buffer (contents are NOT preserved)
int vcp_blocks_ens(vcp_ctx* ctx, DevBuf& b, size_t bytes);

// The partition (blockpart.hip).  key = the coordinates the partition reads, motor = the coordinates every DBImproved
// clusters on (the same array unless the caller came through getClusterFromList); both on the device, finite.  Fills
// rows / cols / nblocks / cell sizes, blockof [n], blockstart [nblocks + 2], the block-major arrays bl [n], blk_t [n],
// motor_bm [n * 2], the list of large blocks, m and the host copy of blockstart; one host synchronisation per select
// round (normally one) and one at the end.
int vcp_blocks_partition(vcp_ctx* ctx, BlocksState* s, const double* d_key, const double* d_motor, int64_t n,
                         int pts_in_cell);
// the same in two stages, for ranks that each build a share of the blocks (blockpart.hip)
int vcp_blocks_plan(vcp_ctx* ctx, BlocksState* s, const double* d_key, const double* d_motor, int64_t n, int pts_in_cell,
                    bool want_cuts);
int vcp_blocks_build(vcp_ctx* ctx, BlocksState* s, uint32_t S_lo, uint32_t S_hi, uint32_t off0, int64_t n_loc);

// internal (multi.hip): vcp_blocks_finish_zcoords_dev that also writes the share's part of the merge order
extern "C" int vcp_blocks_finish_zcoords_order(vcp_ctx* ctx, double* d_zcoords, int64_t* d_merge_order);
