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
  DevBuf sel, cand, counts, rec, rec2, stage, rank, binfo, slicelist, vlist, fall, gcnt;
  bool virt_clean = false;  // gcnt is all zero
  std::vector<uint32_t> h_blockstart;
  DevBuf biglist;  // blocks of more than BIG_BLOCK points, listed by the partition (device), in no particular order
  uint32_t nbig = 0;
  // blocks [0, cov_hi) have been clustered by this context, in adjoining ranges, and hold totalC_acc clusters
  bool cov_ok = true;
  int64_t cov_hi = 0, totalC_acc = 0;
  bool ready = false;
};

constexpr uint32_t VCP_BIG_BLOCK = 1024;  // blocks beyond this take the workgroup-per-block kernels

// ensure capacity of a state-owned buffer (contents are NOT preserved)
int vcp_blocks_ens(vcp_ctx* ctx, DevBuf& b, size_t bytes);

// The partition (blockpart.hip).  key = the coordinates the partition reads, motor = the coordinates every DBImproved
// clusters on (the same array unless the caller came through getClusterFromList); both on the device, finite.  Fills
// rows / cols / nblocks / cell sizes, blockof [n], blockstart [nblocks + 2], the block-major arrays bl [n], blk_t [n],
// motor_bm [n * 2], the list of large blocks, m and the host copy of blockstart; one host synchronisation per select
// round (normally one) and one at the end.
int vcp_blocks_partition(vcp_ctx* ctx, BlocksState* s, const double* d_key, const double* d_motor, int64_t n,
                         int pts_in_cell);
