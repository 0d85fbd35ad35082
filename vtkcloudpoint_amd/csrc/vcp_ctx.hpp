// vcp_ctx.hpp -- context, workspace and error plumbing shared by the libvcp.so translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "vcp.h"

// A growable device buffer owned by the context (no hipMalloc on the steady-state path).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool registered = false;  // listed in vcp_ctx::bufs (exactly once, whatever happens to p afterwards)
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

struct Phase {
  const char* name;
  hipEvent_t ev;  // recorded BEFORE the phase starts
};

struct vcp_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  hipDeviceProp_t prop;
  // pinned scratch for tiny readbacks (64 KB).  Who reads back where (byte offsets; a context runs one call at a time on
  // one stream, and every user has consumed its words before the call that wrote them returns or goes on):
  //   [0, 1024)     the DBSCAN engine: bounds, counters, work sizes (dbscan.hip); the block partition's bounds (blockpart.hip);
  //                 the finish stage's counters (blocks.hip)
  //   [1024, 2048)  the partition's SelState (blockpart.hip); the all-pairs kernel's counters (blocks.hip: blocks_cluster);
  //                 DB's counters (dbdead.hip, dbpairs.hip)
  //   [2048, 2064)  DB pair by pair: next seed / frontier size (dbpairs.hip)
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // pinned staging area of the host-buffer entry points with several small arrays (vcp_stage; grown on demand, <= 64 MiB)
  void* stage = nullptr;
  size_t stage_bytes = 0;
  uint32_t scan_gen = 0;  // generation number of the scan descriptors in b_scan_tmp (vcp_ctx.hip: k_scan)
  // workspace
  std::vector<DevBuf*> bufs;
  DevBuf b_cellcnt, b_cellof, b_rank, b_sorted, b_sidx, b_flags, b_parent, b_minord, b_seedflag,
      b_rootcl, b_clseed, b_scan_tmp, b_misc, b_in0, b_in1, b_in2, b_in3, b_out0, b_out1, b_out2,
      b_out3, b_icp_part, b_aux0, b_aux1, b_aux2, b_aux3, b_aux4, b_aux5, b_pos, b_labk, b_sgroup, b_wl, b_skey, b_sorttmp, b_hist, b_rec, b_nn_misc, b_nn_cells, b_nn_cid, b_nn_rec, b_nn_cur, b_nbr, b_nboff, b_sorted32, b_self, b_outcur, b_fineq, b_rec2, b_bstart, b_ctw, b_ctd, b_bstate;
  struct BlocksState* blocks = nullptr;  // staged block-partitioned pipeline (blocks.hip)
  struct SlabState* slab = nullptr;      // staged exact multi-GPU DBSCAN (dbscan.hip: vcp_slab_*)
  // timing
  bool timing = false;
  std::vector<Phase> phases;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<std::pair<const char*, float>> last_timing;
};

int vcp_fail(vcp_ctx* ctx, int code, const char* fmt, ...);
// pinned host memory of at least `bytes` (nullptr when bytes > 64 MiB or the allocation fails: callers then copy
// array by array from the caller's pageable memory)
void* vcp_stage(vcp_ctx* ctx, size_t bytes);

#define VCP_HIP(ctx, call)                                                                    \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return vcp_fail((ctx), VCP_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,        \
                      hipGetErrorString(e__));                                                \
  } while (0)

#define VCP_TRY(expr)        \
  do {                       \
    int rc__ = (expr);       \
    if (rc__ != VCP_OK) return rc__; \
  } while (0)

// ensure capacity (contents are NOT preserved)
int vcp_ensure(vcp_ctx* ctx, DevBuf& b, size_t bytes);
// bind the calling thread to the context's device
int vcp_bind(vcp_ctx* ctx);
// timing
void vcp_phase_reset(vcp_ctx* ctx);
void vcp_phase(vcp_ctx* ctx, const char* name);  // marks the start of a phase
int vcp_phase_finish(vcp_ctx* ctx);              // closes the last phase, syncs, fills last_timing

static inline unsigned vcp_blocks(int64_t n, int per_block, int cap = 1 << 30) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// exclusive scan of n uint32 (in place allowed: out may equal in); writes the grand total to
// d_total (device uint32) if non-null.  Defined in scan.hip.
int vcp_exclusive_scan_u32(vcp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n,
                           uint32_t* d_total);
int vcp_exclusive_max_scan_u32(vcp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n,
                           uint32_t* d_total);
