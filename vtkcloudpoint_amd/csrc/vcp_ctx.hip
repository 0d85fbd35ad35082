// vcp_ctx.hip -- context lifetime, workspace, per-phase hipEvent timing and the u32 scan used
// by the grid build and the canonical cluster numbering.
#include <cstring>

#include "vcp_ctx.hpp"

static thread_local std::string g_create_err;

int vcp_fail(vcp_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf; else g_create_err = buf;
  return code;
}

int vcp_bind(vcp_ctx* ctx) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_HIP(ctx, hipSetDevice(ctx->device));
  return VCP_OK;
}

int vcp_ensure(vcp_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.cap >= bytes) return VCP_OK;
  if (b.p) {
    VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VCP_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  if (!b.registered) {  // a buffer whose regrow failed (p == nullptr again) must not be listed twice
    ctx->bufs.push_back(&b);
    b.registered = true;
  }
  size_t want = bytes + bytes / 8 + 256;  // headroom so that slowly growing inputs do not realloc
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    return vcp_fail(ctx, VCP_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  b.cap = want;
  return VCP_OK;
}

void vcp_phase_reset(vcp_ctx* ctx) {
  ctx->phases.clear();
  ctx->ev_used = 0;
}

void vcp_phase(vcp_ctx* ctx, const char* name) {
  if (!ctx->timing) return;
  if (ctx->ev_used == ctx->ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    ctx->ev_pool.push_back(e);
  }
  hipEvent_t e = ctx->ev_pool[ctx->ev_used++];
  (void)hipEventRecord(e, ctx->stream);
  ctx->phases.push_back(Phase{name, e});
}

int vcp_phase_finish(vcp_ctx* ctx) {
  if (!ctx->timing) return VCP_OK;
  vcp_phase(ctx, nullptr);
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->last_timing.clear();
  for (size_t i = 0; i + 1 < ctx->phases.size(); i++) {
    float ms = 0.f;
    VCP_HIP(ctx, hipEventElapsedTime(&ms, ctx->phases[i].ev, ctx->phases[i + 1].ev));
    ctx->last_timing.emplace_back(ctx->phases[i].name, ms);
  }
  return VCP_OK;
}

// ------------------------------------------------------------------------------------------
// exclusive scan (u32), reduce-then-scan over tiles of 8192 elements:
//   k_scan_tile_sums  one workgroup per tile, 16-B loads, tile sum
//   k_scan_offsets    one workgroup scans the tile sums (and emits the grand total)
//   k_scan_tiles      one workgroup per tile re-reads it, scans it in 8 chunks of 1024 with a running carry
// 2 reads + 1 write of the array; in-place allowed (a tile is read before it is written, tiles are disjoint).
// ------------------------------------------------------------------------------------------
namespace {
constexpr int ST = 256;            // threads
constexpr int SCH = 8;             // chunks per tile
constexpr int STILE = ST * 4 * SCH;  // 8192 elements per tile

// the scan operator: sum, or max (identity 0 for both)
template <bool MX>
__device__ __forceinline__ uint32_t sop(uint32_t a, uint32_t b) {
  return MX ? max(a, b) : a + b;
}

template <bool MX>
__device__ __forceinline__ uint32_t wave_incl(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v = sop<MX>(v, t);
  }
  return v;
}

// block-wide exclusive scan of one value per thread; *total = block sum (max).  Two barriers.
template <bool MX>
__device__ __forceinline__ uint32_t block_excl(uint32_t v, uint32_t* total, uint32_t* sm /*[ST/64 + 1]*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t inc = wave_incl<MX>(v, lane);
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < ST / 64; k++) {
    const uint32_t x = sm[k];
    if (k < w) base = sop<MX>(base, x);
    tot = sop<MX>(tot, x);
  }
  __syncthreads();
  *total = tot;
  // exclusive value of this thread: everything before it in its wave, plus the earlier waves
  uint32_t prev = __shfl_up(inc, 1, 64);
  if (lane == 0) prev = 0;
  return sop<MX>(base, prev);
}

template <bool VEC>
__device__ __forceinline__ uint4 ld4(const uint32_t* __restrict__ in, int64_t i, int64_t n) {
  if (VEC && i + 3 < n) return *reinterpret_cast<const uint4*>(in + i);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (i < n) v.x = in[i];
  if (i + 1 < n) v.y = in[i + 1];
  if (i + 2 < n) v.z = in[i + 2];
  if (i + 3 < n) v.w = in[i + 3];
  return v;
}

template <bool VEC, bool MX>
__global__ __launch_bounds__(ST) void k_scan_tile_sums(const uint32_t* __restrict__ in, int64_t n,
                                                      uint32_t* __restrict__ tsum) {
  const int64_t base = (int64_t)blockIdx.x * STILE;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCH; k++) {
    const uint4 v = ld4<VEC>(in, base + ((int64_t)k * ST + threadIdx.x) * 4, n);
    s = sop<MX>(s, sop<MX>(sop<MX>(v.x, v.y), sop<MX>(v.z, v.w)));
  }
  __shared__ uint32_t sm[ST / 64 + 1];
  uint32_t tot;
  block_excl<MX>(s, &tot, sm);
  if (threadIdx.x == 0) tsum[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of nt tile sums in place; thread t owns a contiguous run
template <bool MX>
__global__ __launch_bounds__(ST) void k_scan_offsets(uint32_t* __restrict__ tsum, int nt, uint32_t* __restrict__ total) {
  const int per = (nt + ST - 1) / ST;
  const int lo = min((int)threadIdx.x * per, nt), hi = min(lo + per, nt);
  uint32_t s = 0;
  for (int i = lo; i < hi; i++) s = sop<MX>(s, tsum[i]);
  __shared__ uint32_t sm[ST / 64 + 1];
  uint32_t tot;
  uint32_t pre = block_excl<MX>(s, &tot, sm);
  for (int i = lo; i < hi; i++) {
    const uint32_t v = tsum[i];
    tsum[i] = pre;
    pre = sop<MX>(pre, v);
  }
  if (threadIdx.x == 0 && total) *total = tot;
}

template <bool VEC, bool MX>
__global__ __launch_bounds__(ST) void k_scan_tiles(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n,
                                                  const uint32_t* __restrict__ toff) {
  const int64_t base = (int64_t)blockIdx.x * STILE;
  __shared__ uint32_t sm[ST / 64 + 1];
  uint32_t carry = toff[blockIdx.x];
#pragma unroll 1
  for (int k = 0; k < SCH; k++) {
    const int64_t i = base + ((int64_t)k * ST + threadIdx.x) * 4;
    const uint4 v = ld4<VEC>(in, i, n);
    uint32_t tot;
    const uint32_t pre =
        sop<MX>(carry, block_excl<MX>(sop<MX>(sop<MX>(v.x, v.y), sop<MX>(v.z, v.w)), &tot, sm));
    const uint32_t o1 = sop<MX>(pre, v.x), o2 = sop<MX>(o1, v.y), o3 = sop<MX>(o2, v.z);
    const uint4 o = make_uint4(pre, o1, o2, o3);
    if (VEC && i + 3 < n) {
      *reinterpret_cast<uint4*>(out + i) = o;
    } else {
      if (i < n) out[i] = o.x;
      if (i + 1 < n) out[i + 1] = o.y;
      if (i + 2 < n) out[i + 2] = o.z;
      if (i + 3 < n) out[i + 3] = o.w;
    }
    carry = sop<MX>(carry, tot);
  }
}
}  // namespace

template <bool MX>
static int scan_u32(vcp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n, uint32_t* d_total) {
  if (n <= 0) {
    if (d_total) VCP_HIP(ctx, hipMemsetAsync(d_total, 0, 4, ctx->stream));
    return VCP_OK;
  }
  const int64_t nt = (n + STILE - 1) / STILE;
  VCP_TRY(vcp_ensure(ctx, ctx->b_scan_tmp, (size_t)(nt + 16) * sizeof(uint32_t)));
  uint32_t* tsum = ctx->b_scan_tmp.as<uint32_t>();
  const bool vec = (((uintptr_t)d_in | (uintptr_t)d_out) & 15u) == 0;
  hipStream_t st = ctx->stream;
  if (vec) hipLaunchKernelGGL((k_scan_tile_sums<true, MX>), dim3((unsigned)nt), dim3(ST), 0, st, d_in, n, tsum);
  else hipLaunchKernelGGL((k_scan_tile_sums<false, MX>), dim3((unsigned)nt), dim3(ST), 0, st, d_in, n, tsum);
  hipLaunchKernelGGL(k_scan_offsets<MX>, dim3(1), dim3(ST), 0, st, tsum, (int)nt, d_total);
  if (vec) hipLaunchKernelGGL((k_scan_tiles<true, MX>), dim3((unsigned)nt), dim3(ST), 0, st, d_in, d_out, n, tsum);
  else hipLaunchKernelGGL((k_scan_tiles<false, MX>), dim3((unsigned)nt), dim3(ST), 0, st, d_in, d_out, n, tsum);
  VCP_HIP(ctx, hipGetLastError());
  return VCP_OK;
}

int vcp_exclusive_scan_u32(vcp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n, uint32_t* d_total) {
  return scan_u32<false>(ctx, d_in, d_out, n, d_total);
}

// out[i] = max(in[0..i-1]), 0 for i = 0
int vcp_exclusive_max_scan_u32(vcp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n, uint32_t* d_total) {
  return scan_u32<true>(ctx, d_in, d_out, n, d_total);
}

void* vcp_stage(vcp_ctx* ctx, size_t bytes) {
  if (bytes > ((size_t)64 << 20)) return nullptr;
  if (bytes <= ctx->stage_bytes) return ctx->stage;
  if (ctx->stage) (void)hipHostFree(ctx->stage);
  ctx->stage = nullptr;
  ctx->stage_bytes = 0;
  const size_t want = std::max(bytes + bytes / 2, (size_t)1 << 20);
  if (hipHostMalloc(&ctx->stage, want, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    ctx->stage = nullptr;
    return nullptr;
  }
  ctx->stage_bytes = want;
  return ctx->stage;
}

// ------------------------------------------------------------------------------------------
extern "C" {

int vcp_version(void) { return VCP_VERSION_MAJOR * 1000 + VCP_VERSION_MINOR; }

const char* vcp_last_error(const vcp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int vcp_create(int device_id, vcp_ctx** out) {
  if (!out) return VCP_ERR_ARG;
  *out = nullptr;
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0)
    return vcp_fail(nullptr, VCP_ERR_NO_DEVICE, "no HIP device (%s); libvcp has no CPU fallback",
                    e == hipSuccess ? "count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= cnt)
    return vcp_fail(nullptr, VCP_ERR_NO_DEVICE, "device %d out of range (0..%d)", device_id, cnt - 1);
  vcp_ctx* c = new vcp_ctx();
  c->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&c->prop, device_id) != hipSuccess ||
      hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return vcp_fail(nullptr, VCP_ERR_HIP, "device %d initialisation failed", device_id);
  }
  c->stream = c->own_stream;
  c->pinned_bytes = 1 << 16;
  if (hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return vcp_fail(nullptr, VCP_ERR_NOMEM, "pinned scratch allocation failed");
  }
  *out = c;
  return VCP_OK;
}

void vcp_blocks_state_free(vcp_ctx* ctx);  // blocks.hip
void vcp_slab_state_free(vcp_ctx* ctx);    // dbscan.hip

int vcp_release_workspace(vcp_ctx* ctx) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  vcp_blocks_state_free(ctx);
  vcp_slab_state_free(ctx);
  for (DevBuf* b : ctx->bufs) {
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
    b->registered = false;
  }
  ctx->bufs.clear();  // vcp_ensure registers a buffer again when it allocates it
  if (ctx->stage) (void)hipHostFree(ctx->stage);
  ctx->stage = nullptr;
  ctx->stage_bytes = 0;
  return VCP_OK;
}

void vcp_destroy(vcp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  vcp_blocks_state_free(ctx);
  vcp_slab_state_free(ctx);
  for (DevBuf* b : ctx->bufs) {
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr;
  }
  for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->stage) (void)hipHostFree(ctx->stage);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int vcp_set_stream(vcp_ctx* ctx, void* s) {
  if (!ctx) return VCP_ERR_ARG;
  ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
  return VCP_OK;
}

int vcp_dev_alloc(vcp_ctx* ctx, uint64_t bytes, void** dptr) {
  if (!ctx || !dptr) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
  if (e != hipSuccess) return vcp_fail(ctx, VCP_ERR_NOMEM, "hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
  return VCP_OK;
}

int vcp_dev_free(vcp_ctx* ctx, void* dptr) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  VCP_HIP(ctx, hipFree(dptr));
  return VCP_OK;
}

int vcp_h2d(vcp_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  VCP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VCP_OK;
}

int vcp_d2h(vcp_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  if (!ctx) return VCP_ERR_ARG;
  VCP_TRY(vcp_bind(ctx));
  VCP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VCP_OK;
}

int vcp_timing_enable(vcp_ctx* ctx, int on) {
  if (!ctx) return VCP_ERR_ARG;
  ctx->timing = on != 0;
  return VCP_OK;
}

int vcp_timing_count(vcp_ctx* ctx) { return ctx ? (int)ctx->last_timing.size() : 0; }

int vcp_timing_get(vcp_ctx* ctx, int i, const char** name, float* ms) {
  if (!ctx || i < 0 || i >= (int)ctx->last_timing.size()) return VCP_ERR_ARG;
  if (name) *name = ctx->last_timing[i].first;
  if (ms) *ms = ctx->last_timing[i].second;
  return VCP_OK;
}

}  // extern "C"
