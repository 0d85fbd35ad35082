// multi.hip -- several GPUs behind the C-ABI, driven from ONE process.
//
// The reference fans its blocks out from one process: ThreadPool.QueueUserWorkItem(StartCode, cells[i]) per block
// (FrmMain.cs:1356-1359), merge on the UI thread (CompleteWork3, :1442-1520).  The drop-in equivalent for a C# host is
// one call that drives N devices: vcp_multi holds one vcp_ctx per listed device, a host thread per device runs the
// staged block pipeline (identical partition on every device, contiguous block ranges balanced on point count,
// vcp_blocks_share_plan), the block-major label slices travel to device 0 with hipMemcpyPeerAsync over xGMI (what the
// multi-process form does with ONE RCCL all-gather, distributed.py: sharded_blocks), device 0 runs CompleteWork3.
// A device id may be listed several times (several contexts on one GPU): how the path is exercised on a one-GPU box.
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "vcp_ctx.hpp"

struct vcp_multi {
  std::vector<vcp_ctx*> ctx;
  std::vector<int> dev;
  std::string err;
  // per-device cloud, label slice and (device 0) outputs, kept between calls
  std::vector<DevBuf> motor, key, local;
  DevBuf labels, blockof, order;
};

static thread_local std::string g_multi_err;

static int mfail(vcp_multi* m, int code, const std::string& msg) {
  if (m) m->err = msg; else g_multi_err = msg;
  return code;
}

// grow-only device buffer on the device of context c (the buffer is NOT registered with the context: vcp_multi owns it)
static int mensure(vcp_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.cap >= bytes) return VCP_OK;
  VCP_TRY(vcp_bind(c));
  if (b.p) {
    VCP_HIP(c, hipStreamSynchronize(c->stream));
    VCP_HIP(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    return vcp_fail(c, VCP_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  b.cap = want;
  return VCP_OK;
}

extern "C" {

// Contiguous block ranges balanced on the point count: cuts[r] = first block of rank r (cuts[0] = 0, cuts[world] =
// nblocks); rank r starts at the first block whose first block-major position is >= m * r / world.  blockstart
// [nblocks + 1] ascending, blockstart[nblocks] = m.  Pure host arithmetic: every rank (process or device thread)
// derives ALL ranges from the identical partition, so no size exchange precedes the data.
int vcp_blocks_share_plan(const uint32_t* blockstart, int64_t nblocks, int world, int64_t* cuts) {
  if (!blockstart || !cuts || nblocks < 0 || world < 1) return VCP_ERR_ARG;
  const int64_t m = blockstart[nblocks];
  cuts[0] = 0;
  for (int r = 1; r < world; r++) {
    const uint32_t target = (uint32_t)((m * (int64_t)r) / world);
    const uint32_t* it = std::lower_bound(blockstart, blockstart + nblocks, target);
    cuts[r] = std::max<int64_t>(cuts[r - 1], (int64_t)(it - blockstart));
  }
  cuts[world] = nblocks;
  return VCP_OK;
}

const char* vcp_multi_last_error(const vcp_multi* m) { return m ? m->err.c_str() : g_multi_err.c_str(); }

int vcp_create_multi(const int* device_ids, int n, vcp_multi** out) {
  if (!out) return VCP_ERR_ARG;
  *out = nullptr;
  if (!device_ids || n < 1 || n > 64) return mfail(nullptr, VCP_ERR_ARG, "vcp_create_multi needs 1..64 device ids");
  vcp_multi* m = new vcp_multi();
  for (int i = 0; i < n; i++) {
    vcp_ctx* c = nullptr;
    const int rc = vcp_create(device_ids[i], &c);
    if (rc != VCP_OK) {
      const std::string why = vcp_last_error(nullptr);
      for (vcp_ctx* p : m->ctx) vcp_destroy(p);
      delete m;
      return mfail(nullptr, rc, "device " + std::to_string(device_ids[i]) + ": " + why);
    }
    m->ctx.push_back(c);
    m->dev.push_back(device_ids[i]);
  }
  m->motor.resize(n);
  m->key.resize(n);
  m->local.resize(n);
  // peer access between distinct devices (speed only: hipMemcpyPeerAsync stages through the host without it)
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      if (m->dev[i] != m->dev[j]) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->dev[i], m->dev[j]) == hipSuccess && can && hipSetDevice(m->dev[i]) == hipSuccess) {
          const hipError_t e = hipDeviceEnablePeerAccess(m->dev[j], 0);
          if (e != hipSuccess) (void)hipGetLastError();  // already enabled, or not possible: not an error here
        }
      }
  *out = m;
  return VCP_OK;
}

void vcp_destroy_multi(vcp_multi* m) {
  if (!m) return;
  auto drop = [&](vcp_ctx* c, DevBuf& b) {
    if (b.p && vcp_bind(c) == VCP_OK) (void)hipFree(b.p);
    b.p = nullptr;
  };
  for (size_t i = 0; i < m->ctx.size(); i++) {
    (void)hipSetDevice(m->dev[i]);
    (void)hipStreamSynchronize(m->ctx[i]->stream);
    drop(m->ctx[i], m->motor[i]);
    drop(m->ctx[i], m->key[i]);
    drop(m->ctx[i], m->local[i]);
  }
  if (!m->ctx.empty()) {
    drop(m->ctx[0], m->labels);
    drop(m->ctx[0], m->blockof);
    drop(m->ctx[0], m->order);
  }
  for (vcp_ctx* c : m->ctx) vcp_destroy(c);
  delete m;
}

int vcp_multi_count(const vcp_multi* m) { return m ? (int)m->ctx.size() : 0; }

vcp_ctx* vcp_multi_ctx(vcp_multi* m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

int vcp_dbscan_blocks_multi(vcp_multi* mg, const double* key_xy, const double* motor, int64_t n, double eps, int min_pts,
                            int pts_in_cell, int small_max, int32_t* labels, int32_t* block_of, int64_t* merge_order,
                            int64_t* m_out, int32_t* rows, int32_t* cols, int32_t* kept, int32_t* del_sum,
                            int32_t* cluster_amount, int64_t* dist_evals) {
  if (!mg) return VCP_ERR_ARG;
  if (n < 0 || (n > 0 && (!motor || !labels))) return mfail(mg, VCP_ERR_ARG, "null buffer");
  const int W = (int)mg->ctx.size();
  // every device: upload, partition (identical everywhere), cluster its block range
  std::vector<int> rc(W, VCP_OK);
  std::vector<int64_t> ev(W, 0), plo(W, 0), phi(W, 0), mm(W, 0), nbk(W, 0);
  std::vector<int32_t> rws(W, 0), cls(W, 0);
  auto work = [&](int i) {
    vcp_ctx* c = mg->ctx[i];
    auto run = [&]() -> int {
      VCP_TRY(vcp_bind(c));
      const size_t bytes = (size_t)std::max<int64_t>(n, 1) * 16;
      VCP_TRY(mensure(c, mg->motor[i], bytes));
      if (n > 0) VCP_HIP(c, hipMemcpyAsync(mg->motor[i].p, motor, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
      if (key_xy) {
        VCP_TRY(mensure(c, mg->key[i], bytes));
        if (n > 0) VCP_HIP(c, hipMemcpyAsync(mg->key[i].p, key_xy, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
        VCP_TRY(vcp_blocks_begin_keyed_dev(c, mg->key[i].as<double>(), mg->motor[i].as<double>(), n, eps, min_pts, pts_in_cell,
                                           small_max, &rws[i], &cls[i], &nbk[i], &mm[i]));
      } else {
        VCP_TRY(vcp_blocks_begin_dev(c, mg->motor[i].as<double>(), n, eps, min_pts, pts_in_cell, small_max, &rws[i], &cls[i],
                                     &nbk[i], &mm[i]));
      }
      int32_t lo = 0, hi = 0;
      VCP_TRY(vcp_blocks_share(c, i, W, &lo, &hi, &plo[i], &phi[i]));
      VCP_TRY(mensure(c, mg->local[i], (size_t)(mm[i] + 1) * 4));
      VCP_TRY(vcp_blocks_cluster_dev(c, lo, hi, mg->local[i].as<int32_t>(), &ev[i]));
      return VCP_OK;
    };
    rc[i] = run();
  };
  if (W == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int i = 0; i < W; i++) th.emplace_back(work, i);
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < W; i++)
    if (rc[i] != VCP_OK) return mfail(mg, rc[i], "device " + std::to_string(mg->dev[i]) + ": " + vcp_last_error(mg->ctx[i]));
  for (int i = 1; i < W; i++)
    if (mm[i] != mm[0] || nbk[i] != nbk[0]) return mfail(mg, VCP_ERR_HIP, "the devices disagree on the partition");
  // the label slices travel to device 0 (xGMI peer copies; what ONE RCCL all-gather does in the multi-process form)
  vcp_ctx* c0 = mg->ctx[0];
  auto gather = [&]() -> int {
    VCP_TRY(vcp_bind(c0));
    int64_t evsum = ev[0];
    for (int i = 1; i < W; i++) {
      evsum += ev[i];
      if (phi[i] > plo[i])
        VCP_HIP(c0, hipMemcpyPeerAsync(mg->local[0].as<int32_t>() + plo[i], mg->dev[0], mg->local[i].as<int32_t>() + plo[i],
                                       mg->dev[i], (size_t)(phi[i] - plo[i]) * 4, c0->stream));
    }
    VCP_TRY(mensure(c0, mg->labels, (size_t)std::max<int64_t>(n, 1) * 4));
    VCP_TRY(mensure(c0, mg->blockof, (size_t)std::max<int64_t>(n, 1) * 4));
    if (merge_order) VCP_TRY(mensure(c0, mg->order, (size_t)(mm[0] + 1) * 8));
    int64_t mo = 0;
    VCP_TRY(vcp_blocks_finish_dev(c0, mg->local[0].as<int32_t>(), evsum, mg->labels.as<int32_t>(),
                                  block_of ? mg->blockof.as<int32_t>() : nullptr,
                                  merge_order ? mg->order.as<int64_t>() : nullptr, &mo, kept, del_sum, cluster_amount,
                                  dist_evals));
    if (n > 0) VCP_HIP(c0, hipMemcpyAsync(labels, mg->labels.p, (size_t)n * 4, hipMemcpyDeviceToHost, c0->stream));
    if (block_of && n > 0) VCP_HIP(c0, hipMemcpyAsync(block_of, mg->blockof.p, (size_t)n * 4, hipMemcpyDeviceToHost, c0->stream));
    if (merge_order && mo > 0)
      VCP_HIP(c0, hipMemcpyAsync(merge_order, mg->order.p, (size_t)mo * 8, hipMemcpyDeviceToHost, c0->stream));
    VCP_HIP(c0, hipStreamSynchronize(c0->stream));
    if (m_out) *m_out = mo;
    return VCP_OK;
  };
  const int g = gather();
  if (g != VCP_OK) return mfail(mg, g, std::string("device ") + std::to_string(mg->dev[0]) + ": " + vcp_last_error(c0));
  if (rows) *rows = rws[0];
  if (cols) *cols = cls[0];
  return VCP_OK;
}

}  // extern "C"
