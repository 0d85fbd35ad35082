// multi.hip -- several GPUs behind the C-ABI, driven from ONE process.
//
// The reference fans its blocks out from one process: ThreadPool.QueueUserWorkItem(StartCode, cells[i]) per block
// (FrmMain.cs:1356-1359), merge on the UI thread (CompleteWork3, :1442-1520).  The drop-in equivalent for a C# host is
// one call that drives N devices: vcp_multi holds one vcp_ctx per listed device, a host thread per device runs the
// share-wise block pipeline (round 3: the streaming passes that decide the partition on every device, then build, cluster
// and merge of the device's own share of the blocks -- the stages distributed.py: sharded_pipeline drives from several
// processes); the exchanges between the stages are this process's memory and hipMemcpyPeerAsync over xGMI; device 0 runs
// the global noise pass over the shares' active points and assembles the label array.
// A device id may be listed several times (several contexts on one GPU): how the path is exercised on a one-GPU box.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "blocks_state.hpp"

struct vcp_multi {
  std::vector<vcp_ctx*> ctx;
  std::vector<int> dev;
  std::string err;
  // per-device cloud, label slice and (device 0) outputs, kept between calls
  std::vector<DevBuf> motor, key, local, zc, zl, pairs, mo;
  DevBuf labels, order, zall, zlab, pall;
};

namespace {
// the device threads of one call meet here between the stages
struct Barrier {
  explicit Barrier(int n) : n(n) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    const int g = gen;
    if (++cnt == n) {
      cnt = 0;
      gen++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
  std::mutex mu;
  std::condition_variable cv;
  int n, cnt = 0, gen = 0;
};
}  // namespace

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
  m->zc.resize(n);
  m->zl.resize(n);
  m->pairs.resize(n);
  m->mo.resize(n);
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
    drop(m->ctx[i], m->zc[i]);
    drop(m->ctx[i], m->zl[i]);
    drop(m->ctx[i], m->pairs[i]);
    drop(m->ctx[i], m->mo[i]);
  }
  if (!m->ctx.empty()) {
    drop(m->ctx[0], m->labels);
    drop(m->ctx[0], m->order);
    drop(m->ctx[0], m->zall);
    drop(m->ctx[0], m->zlab);
    drop(m->ctx[0], m->pall);
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
  // Every device repeats the streaming passes that decide the partition (vcp_blocks_plan_dev), then builds, clusters
  // and merges its own share of the blocks; the stages are the ones distributed.py: sharded_pipeline drives from several
  // processes, the exchanges are this process's memory and peer copies:
  //   1  per-share counters -> renumbering offsets, who zeroes whose last entry (the clusLen quirk across a boundary)
  //   2  the active points of the shares' zero lists -> device 0, which runs the global noise pass (an eighth of the
  //      noise at the reference defaults: 0.3 ms) and hands the labels back
  //   3  (index, label) pairs -> device 0 -> the label array
  struct Share {
    int rc = VCP_OK;
    int64_t info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t ev = 0, Z = 0, A = 0, nsuper = 0, nblocks = 0;
    int32_t rows = 0, cols = 0, blo = 0, bhi = 0;
    bool zero_me = false;
    int64_t kept_off = 0;
  };
  std::vector<Share> sh(W);
  Barrier bar(W);
  std::atomic<int> failed{0};
  int64_t kept_total = 0, clusters_total = 0, m_total = 0, z_total = 0, a_total = 0, ev_blocks = 0, noise_ev = 0;
  int32_t cf_final = 0;
  std::string quirk_err;
  auto fail_here = [&](int i, int rc) {
    sh[i].rc = rc;
    failed.store(1);
  };
  auto work = [&](int i) {
    vcp_ctx* c = mg->ctx[i];
    Share& S = sh[i];
    // ---- plan, build, cluster, local merge
    auto stage1 = [&]() -> int {
      VCP_TRY(vcp_bind(c));
      const size_t bytes = (size_t)std::max<int64_t>(n, 1) * 16;
      VCP_TRY(mensure(c, mg->motor[i], bytes));
      if (n > 0) VCP_HIP(c, hipMemcpyAsync(mg->motor[i].p, motor, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
      if (key_xy) {
        VCP_TRY(mensure(c, mg->key[i], bytes));
        if (n > 0) VCP_HIP(c, hipMemcpyAsync(mg->key[i].p, key_xy, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
      }
      VCP_TRY(vcp_blocks_plan_dev(c, key_xy ? mg->key[i].as<double>() : nullptr, mg->motor[i].as<double>(), n, eps, min_pts,
                                  pts_in_cell, small_max, &S.rows, &S.cols, &S.nblocks, &S.nsuper));
      std::vector<int64_t> cuts((size_t)W + 1);
      VCP_TRY(vcp_blocks_plan_cuts(c, W, cuts.data()));
      int64_t mloc = 0, nloc = 0;
      VCP_TRY(vcp_blocks_build_dev(c, cuts[(size_t)i], cuts[(size_t)i + 1], &S.blo, &S.bhi, &mloc, &nloc));
      VCP_TRY(mensure(c, mg->local[i], (size_t)(mloc + 1) * 4));
      if (mloc > 0) VCP_TRY(vcp_blocks_cluster_dev(c, S.blo, S.bhi, mg->local[i].as<int32_t>(), &S.ev));
      VCP_TRY(vcp_blocks_finish_local_dev(c, mg->local[i].as<int32_t>(), S.info));
      return VCP_OK;
    };
    int rc = stage1();
    if (rc != VCP_OK) fail_here(i, rc);
    bar.wait();
    // ---- exchange 1 (every thread derives the same numbers)
    if (!failed.load() && i == 0) {
      int64_t ko = 0;
      for (int q = 0; q < W; q++) {
        if (sh[q].info[2] != 0) quirk_err = "clusForMerge index -1 while demoting the first cluster (FrmMain.cs:1487)";
        if (sh[q].info[3]) {  // asks the nearest earlier share with a non-empty block to zero its last entry
          int t = -1;
          for (int pq = 0; pq < q; pq++)
            if (sh[pq].info[4]) t = pq;
          if (t < 0) quirk_err = "clusForMerge index -1 while demoting the first cluster (FrmMain.cs:1487)";
          else sh[t].zero_me = true;
        }
        sh[q].kept_off = ko;
        ko += sh[q].info[1];
        clusters_total += sh[q].info[0];
        m_total += sh[q].info[6];
        ev_blocks += sh[q].ev;
      }
      kept_total = ko;
      if (!quirk_err.empty()) failed.store(2);
    }
    bar.wait();
    // ---- zero list of the share, its active points
    auto stage2 = [&]() -> int {
      VCP_TRY(vcp_blocks_finish_zero_dev(c, S.zero_me ? 1 : 0, &S.Z, &S.A));
      VCP_TRY(mensure(c, mg->zc[i], (size_t)(S.A + 1) * 16));
      if (merge_order) VCP_TRY(mensure(c, mg->mo[i], (size_t)(S.info[6] + 1) * 8));
      VCP_TRY(vcp_blocks_finish_zcoords_order(c, mg->zc[i].as<double>(), merge_order ? mg->mo[i].as<int64_t>() : nullptr));
      return VCP_OK;
    };
    if (!failed.load()) {
      rc = stage2();
      if (rc != VCP_OK) fail_here(i, rc);
    }
    bar.wait();
    // ---- exchange 2: the global noise pass (FrmMain.cs:1507-1516) on device 0 over everybody's active points
    if (!failed.load() && i == 0) {
      auto noise = [&]() -> int {
        for (int q = 0; q < W; q++) {
          z_total += sh[q].Z;
          a_total += sh[q].A;
        }
        VCP_TRY(mensure(c, mg->zall, (size_t)(a_total + 1) * 16));
        VCP_TRY(mensure(c, mg->zlab, (size_t)(a_total + 1) * 4));
        int64_t off = 0;
        for (int q = 0; q < W; q++) {
          if (sh[q].A > 0)
            VCP_HIP(c, hipMemcpyPeerAsync(mg->zall.as<double>() + 2 * off, mg->dev[0], mg->zc[q].p, mg->dev[q],
                                          (size_t)sh[q].A * 16, c->stream));
          off += sh[q].A;
        }
        cf_final = (int32_t)kept_total;
        int64_t ev = 0;
        if (a_total > 0)
          VCP_TRY(vcp_dbscan_dev(c, mg->zall.as<double>(), a_total, 2, VCP_L1_2D, eps, min_pts, (int32_t)kept_total, nullptr,
                                 mg->zlab.as<int32_t>(), nullptr, nullptr, &cf_final, &ev));
        // iritatorNum of the pass over the whole zero list, from the counters of the pass over its active points
        const int64_t K = (int64_t)cf_final - kept_total;
        const int64_t twice = a_total > 0 ? ev / a_total - a_total - K : 0;
        noise_ev = z_total * (z_total + K + twice);
        VCP_HIP(c, hipStreamSynchronize(c->stream));
        return VCP_OK;
      };
      rc = noise();
      if (rc != VCP_OK) fail_here(0, rc);
    }
    bar.wait();
    // ---- the labels of the share's active points come back; (index, label) pairs
    auto stage3 = [&]() -> int {
      int64_t aoff = 0;
      for (int q = 0; q < i; q++) aoff += sh[q].A;
      VCP_TRY(mensure(c, mg->zl[i], (size_t)(S.A + 1) * 4));
      if (S.A > 0) {
        VCP_HIP(c, hipMemcpyPeerAsync(mg->zl[i].p, mg->dev[i], mg->zlab.as<int32_t>() + aoff, mg->dev[0], (size_t)S.A * 4,
                                      c->stream));
        VCP_HIP(c, hipStreamSynchronize(c->stream));
      }
      VCP_TRY(mensure(c, mg->pairs[i], (size_t)(S.info[7] + 1) * 8));
      VCP_TRY(vcp_blocks_finish_pairs_dev(c, (int32_t)S.kept_off, S.A > 0 ? mg->zl[i].as<int32_t>() : nullptr,
                                          mg->pairs[i].as<int64_t>()));
      return VCP_OK;
    };
    if (!failed.load()) {
      rc = stage3();
      if (rc != VCP_OK) fail_here(i, rc);
    }
    bar.wait();
    // ---- exchange 3: device 0 assembles the outputs
    if (!failed.load() && i == 0) {
      auto finish = [&]() -> int {
        VCP_TRY(mensure(c, mg->labels, (size_t)std::max<int64_t>(n, 1) * 4));
        VCP_TRY(mensure(c, mg->pall, (size_t)std::max<int64_t>(n, 1) * 8));
        int64_t off = 0;
        for (int q = 0; q < W; q++) {
          const int64_t cnt = sh[q].info[7];
          if (cnt > 0)
            VCP_HIP(c, hipMemcpyPeerAsync(mg->pall.as<int64_t>() + off, mg->dev[0], mg->pairs[q].p, mg->dev[q],
                                          (size_t)cnt * 8, c->stream));
          off += cnt;
        }
        if (off != n) return vcp_fail(c, VCP_ERR_HIP, "the shares hold %lld of %lld points", (long long)off, (long long)n);
        VCP_HIP(c, hipStreamSynchronize(c->stream));
        VCP_TRY(vcp_scatter_pairs_dev(c, mg->pall.as<int64_t>(), n, n, mg->labels.as<int32_t>()));
        if (n > 0) VCP_HIP(c, hipMemcpyAsync(labels, mg->labels.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (block_of && n > 0)  // the plan has filed every point on every device
          VCP_HIP(c, hipMemcpyAsync(block_of, c->blocks->blockof.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (merge_order && m_total > 0) {
          // clusForMerge: the shares' non-zero entries in final order, then the shares' zero lists (FrmMain.cs:1510-1520)
          VCP_TRY(mensure(c, mg->order, (size_t)(m_total + 1) * 8));
          int64_t nz = 0, zz = m_total - z_total;
          for (int q = 0; q < W; q++) {
            const int64_t mq = sh[q].info[6], zq = sh[q].Z;
            if (mq - zq > 0)
              VCP_HIP(c, hipMemcpyPeerAsync(mg->order.as<int64_t>() + nz, mg->dev[0], mg->mo[q].p, mg->dev[q],
                                            (size_t)(mq - zq) * 8, c->stream));
            if (zq > 0)
              VCP_HIP(c, hipMemcpyPeerAsync(mg->order.as<int64_t>() + zz, mg->dev[0], mg->mo[q].as<int64_t>() + (mq - zq),
                                            mg->dev[q], (size_t)zq * 8, c->stream));
            nz += mq - zq;
            zz += zq;
          }
          VCP_HIP(c, hipMemcpyAsync(merge_order, mg->order.p, (size_t)m_total * 8, hipMemcpyDeviceToHost, c->stream));
        }
        VCP_HIP(c, hipStreamSynchronize(c->stream));
        return VCP_OK;
      };
      rc = finish();
      if (rc != VCP_OK) fail_here(0, rc);
    }
  };
  if (W == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int i = 0; i < W; i++) th.emplace_back(work, i);
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < W; i++)
    if (sh[i].rc != VCP_OK)
      return mfail(mg, sh[i].rc, "device " + std::to_string(mg->dev[i]) + ": " + vcp_last_error(mg->ctx[i]));
  if (!quirk_err.empty()) return mfail(mg, VCP_ERR_INDEX, quirk_err);
  for (int i = 1; i < W; i++)
    if (sh[i].nblocks != sh[0].nblocks || sh[i].nsuper != sh[0].nsuper)
      return mfail(mg, VCP_ERR_HIP, "the devices disagree on the partition");
  if (m_out) *m_out = m_total;
  if (rows) *rows = sh[0].rows;
  if (cols) *cols = sh[0].cols;
  if (kept) *kept = (int32_t)kept_total;
  if (del_sum) *del_sum = (int32_t)(clusters_total - kept_total);
  if (cluster_amount) *cluster_amount = cf_final;
  if (dist_evals) *dist_evals = ev_blocks + noise_ev;
  return VCP_OK;
}

}  // extern "C"
