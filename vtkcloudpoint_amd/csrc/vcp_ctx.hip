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
// exclusive scan (u32), ONE kernel: tiles of 8192 elements chained by decoupled look-back.
//   * a workgroup takes its tile number from a ticket counter (tiles therefore start in order: whatever a tile waits
//     for belongs to a workgroup that is already running), loads its 8 x 1024 elements into registers (16-B loads),
//     reduces them and publishes the tile aggregate;
//   * wave 0 walks back over the descriptors of the preceding tiles, 64 at a time, adding aggregates until it meets
//     a tile whose inclusive prefix is known, publishes its own inclusive prefix, and the tile is scanned out of the
//     registers: 1 read + 1 write of the array (the three-kernel reduce-then-scan it replaces read it twice and cost
//     16 us + two launch gaps per call; the DBSCAN step makes four such calls).
// A descriptor is one 64-bit word {generation << 2 | state, value}, written and read with single relaxed atomics: no
// fences, and no clearing between calls -- the generation number (per context, one per call) tells a fresh word from a
// stale one.  The descriptor array is zero-filled when it is (re)allocated; the last ticket holder resets the ticket.
// In-place allowed (a tile is in registers before it is written; tiles are disjoint).
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

constexpr uint32_t SD_AGG = 1u, SD_INC = 2u;  // descriptor states: tile aggregate / inclusive prefix available

__device__ __forceinline__ void sd_put(unsigned long long* d, uint32_t gen, uint32_t state, uint32_t v) {
  __hip_atomic_store(d, ((unsigned long long)((gen << 2) | state) << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool VEC, bool MX>
__global__ __launch_bounds__(ST) void k_scan(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n,
                                            uint32_t nt, uint32_t gen, uint32_t* __restrict__ ticket,
                                            unsigned long long* __restrict__ desc, uint32_t* __restrict__ total) {
  __shared__ uint32_t s_tile, s_carry;
  __shared__ uint32_t sm[SCH][ST / 64];
  if (threadIdx.x == 0) {
    const uint32_t t = atomicAdd(ticket, 1u);
    if (t == nt - 1) atomicExch(ticket, 0u);  // every ticket of this call has been handed out
    s_tile = t;
    s_carry = 0u;
  }
  __syncthreads();
  const uint32_t tile = s_tile;
  const int64_t base = (int64_t)tile * STILE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint4 v[SCH];
  uint32_t inc[SCH];
#pragma unroll
  for (int k = 0; k < SCH; k++) v[k] = ld4<VEC>(in, base + ((int64_t)k * ST + threadIdx.x) * 4, n);
#pragma unroll
  for (int k = 0; k < SCH; k++) {
    inc[k] = wave_incl<MX>(sop<MX>(sop<MX>(v[k].x, v[k].y), sop<MX>(v[k].z, v[k].w)), lane);
    if (lane == 63) sm[k][w] = inc[k];
  }
  __syncthreads();
  // element order is chunk-major: everything in earlier chunks, then the earlier waves of the own chunk
  uint32_t agg = 0, pre[SCH];
#pragma unroll
  for (int k = 0; k < SCH; k++) {
    uint32_t b = agg;
#pragma unroll
    for (int ww = 0; ww < ST / 64; ww++) {
      const uint32_t x = sm[k][ww];
      if (ww < w) b = sop<MX>(b, x);
      agg = sop<MX>(agg, x);
    }
    pre[k] = b;
  }
  if (w == 0) {
    if (tile == 0) {
      if (lane == 0) sd_put(&desc[0], gen, SD_INC, agg);
    } else {
      if (lane == 0) sd_put(&desc[tile], gen, SD_AGG, agg);
      uint32_t carry = 0;
      int64_t j0 = (int64_t)tile - 1;  // lane l looks at tile j0 - l
      for (;;) {
        const int64_t j = j0 - lane;
        uint32_t stt = SD_INC, val = 0;  // before tile 0: an inclusive prefix of nothing
        if (j >= 0) {
          unsigned long long d;
          do {
            d = __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((uint32_t)(d >> 34) != gen || ((uint32_t)(d >> 32) & 3u) == 0u);
          stt = (uint32_t)(d >> 32) & 3u;
          val = (uint32_t)d;
        }
        const unsigned long long incm = __ballot(stt == SD_INC);
        const int first = __ffsll((long long)incm) - 1;  // nearest tile with a known prefix; -1: 64 aggregates, go on
        uint32_t x = (first < 0 || lane <= first) ? val : 0u;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) x = sop<MX>(x, (uint32_t)__shfl_xor((int)x, d, 64));
        carry = sop<MX>(carry, x);
        if (first >= 0) break;
        j0 -= 64;
      }
      if (lane == 0) {
        sd_put(&desc[tile], gen, SD_INC, sop<MX>(carry, agg));
        s_carry = carry;
      }
    }
  }
  __syncthreads();
  const uint32_t carry = s_carry;
  if (total && tile == nt - 1 && threadIdx.x == 0) *total = sop<MX>(carry, agg);
#pragma unroll
  for (int k = 0; k < SCH; k++) {
    const int64_t i = base + ((int64_t)k * ST + threadIdx.x) * 4;
    uint32_t prev = __shfl_up(inc[k], 1, 64);
    if (lane == 0) prev = 0;
    const uint32_t p0 = sop<MX>(sop<MX>(carry, pre[k]), prev);
    const uint32_t o1 = sop<MX>(p0, v[k].x), o2 = sop<MX>(o1, v[k].y), o3 = sop<MX>(o2, v[k].z);
    if (VEC && i + 3 < n) {
      *reinterpret_cast<uint4*>(out + i) = make_uint4(p0, o1, o2, o3);
    } else {
      if (i < n) out[i] = p0;
      if (i + 1 < n) out[i + 1] = o1;
      if (i + 2 < n) out[i + 2] = o2;
      if (i + 3 < n) out[i + 3] = o3;
    }
  }
}
}  // namespace

template <bool MX>
static int scan_u32(vcp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n, uint32_t* d_total) {
  hipStream_t st = ctx->stream;
  if (n <= 0) {
    if (d_total) VCP_HIP(ctx, hipMemsetAsync(d_total, 0, 4, st));
    return VCP_OK;
  }
  const int64_t nt = (n + STILE - 1) / STILE;
  if (nt >= ((int64_t)1 << 31)) return vcp_fail(ctx, VCP_ERR_TOO_LARGE, "scan of %lld elements", (long long)n);
  // [0, 64) bytes: the ticket; then one descriptor per tile
  const void* before = ctx->b_scan_tmp.p;
  const size_t before_cap = ctx->b_scan_tmp.cap;
  VCP_TRY(vcp_ensure(ctx, ctx->b_scan_tmp, 64 + (size_t)nt * 8));
  ctx->scan_gen = (ctx->scan_gen + 1u) & 0x3FFFFFFFu;
  if (ctx->b_scan_tmp.p != before || ctx->b_scan_tmp.cap != before_cap || ctx->scan_gen == 0u) {  // fresh memory, or the generation numbers wrapped round
    VCP_HIP(ctx, hipMemsetAsync(ctx->b_scan_tmp.p, 0, ctx->b_scan_tmp.cap, st));
    if (ctx->scan_gen == 0u) ctx->scan_gen = 1u;
  }
  uint32_t* ticket = ctx->b_scan_tmp.as<uint32_t>();
  unsigned long long* desc = reinterpret_cast<unsigned long long*>(ctx->b_scan_tmp.as<char>() + 64);
  const bool vec = (((uintptr_t)d_in | (uintptr_t)d_out) & 15u) == 0;
  if (vec)
    hipLaunchKernelGGL((k_scan<true, MX>), dim3((unsigned)nt), dim3(ST), 0, st, d_in, d_out, n, (uint32_t)nt, ctx->scan_gen,
                       ticket, desc, d_total);
  else
    hipLaunchKernelGGL((k_scan<false, MX>), dim3((unsigned)nt), dim3(ST), 0, st, d_in, d_out, n, (uint32_t)nt, ctx->scan_gen,
                       ticket, desc, d_total);
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

int vcp_selftest_scan_dev(vcp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n, int op, uint32_t* total) {
  if (!ctx) return VCP_ERR_ARG;
  if (n < 0 || (n > 0 && (!d_in || !d_out)) || (op != 0 && op != 1)) return vcp_fail(ctx, VCP_ERR_ARG, "bad argument");
  VCP_TRY(vcp_bind(ctx));
  VCP_TRY(vcp_ensure(ctx, ctx->b_self, 64));
  uint32_t* d_tot = ctx->b_self.as<uint32_t>();
  VCP_TRY(op ? vcp_exclusive_max_scan_u32(ctx, d_in, d_out, n, d_tot) : vcp_exclusive_scan_u32(ctx, d_in, d_out, n, d_tot));
  uint32_t* h = reinterpret_cast<uint32_t*>(ctx->pinned);
  VCP_HIP(ctx, hipMemcpyAsync(h, d_tot, 4, hipMemcpyDeviceToHost, ctx->stream));
  VCP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (total) *total = h[0];
  return VCP_OK;
}

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
