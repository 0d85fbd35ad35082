// vcp_ctx.hip -- context lifetime, workspace, per-phase hipEvent timing and the u32 scan used
// by the grid build and the canonical cluster numbering.
#include <string.h>  // rocprim's texture_cache_iterator.hpp calls ::memset without including it

#include <rocprim/rocprim.hpp>

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
  } else {
    ctx->bufs.push_back(&b);
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
// exclusive scan (u32): rocPRIM's single-pass decoupled look-back scan (a plain library primitive); the
// grand total, when asked for, is out[n-1] + in[n-1] (the last input is saved first: in-place is allowed)
// ------------------------------------------------------------------------------------------
namespace {
__global__ void k_scan_total(const uint32_t* __restrict__ out_last, const uint32_t* __restrict__ in_last,
                             uint32_t* __restrict__ total) {
  *total = *out_last + *in_last;
}
}  // namespace

int vcp_exclusive_scan_u32(vcp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n,
                           uint32_t* d_total) {
  if (n <= 0) {
    if (d_total) VCP_HIP(ctx, hipMemsetAsync(d_total, 0, 4, ctx->stream));
    return VCP_OK;
  }
  size_t tb = 0;
  VCP_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, d_in, d_out, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
  VCP_TRY(vcp_ensure(ctx, ctx->b_scan_tmp, tb + 64));
  uint32_t* saved = reinterpret_cast<uint32_t*>(ctx->b_scan_tmp.as<char>() + ((tb + 15) & ~(size_t)15));
  if (d_total)
    VCP_HIP(ctx, hipMemcpyAsync(saved, d_in + (n - 1), 4, hipMemcpyDeviceToDevice, ctx->stream));
  VCP_HIP(ctx, rocprim::exclusive_scan(ctx->b_scan_tmp.p, tb, d_in, d_out, 0u, (size_t)n, rocprim::plus<uint32_t>(),
                                       ctx->stream));
  if (d_total) hipLaunchKernelGGL(k_scan_total, dim3(1), dim3(1), 0, ctx->stream, d_out + (n - 1), saved, d_total);
  VCP_HIP(ctx, hipGetLastError());
  return VCP_OK;
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

void vcp_destroy(vcp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  vcp_blocks_state_free(ctx);
  for (DevBuf* b : ctx->bufs)
    if (b->p) (void)hipFree(b->p);
  for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
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
