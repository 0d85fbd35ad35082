// micro-benchmark: what does moving 10 M (x, y, index) records into B buckets cost on MI355X, by bucket count,
// record layout and write shape?  Decides the grid-build design (csrc/gridbuild.hip).
//   copy        plain 16 B + 4 B streaming copy (ceiling)
//   gather      random 16-B gather by index (the round-1 build)
//   soa/aos     direct scatter through LDS cursors: every lane stores its record at its own slot
//   staged      the chunk is first ordered by bucket in LDS, then consecutive lanes store consecutive slots
//   window*     the output direction: scattered 4+1+1-byte stores inside 2^15-entry windows, direct or LDS-staged
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int PT = 512;

__global__ void k_copy(const double2* __restrict__ xy, const uint32_t* __restrict__ idx, double2* oxy, uint32_t* oidx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    oxy[i] = xy[i];
    oidx[i] = idx[i];
  }
}
__global__ void k_gather(const double2* __restrict__ xy, const uint32_t* __restrict__ perm, double2* oxy, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) oxy[i] = xy[perm[i]];
}
__global__ void k_hist(const uint32_t* __restrict__ key, size_t n, uint32_t B, uint32_t chunk, uint32_t nchunk, uint32_t* counts) {
  extern __shared__ uint32_t h[];
  for (uint32_t k = threadIdx.x; k < B; k += PT) h[k] = 0;
  __syncthreads();
  size_t first = (size_t)blockIdx.x * chunk, last = first + chunk < n ? first + chunk : n;
  for (size_t i = first + threadIdx.x; i < last; i += PT) atomicAdd(&h[key[i]], 1u);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < B; k += PT) counts[(size_t)k * nchunk + blockIdx.x] = h[k];
}
template <int MODE>  // 0 SoA 16+4, 1 AoS 32 B, 2 AoS 16 B
__global__ __launch_bounds__(PT) void k_scatter(const double2* __restrict__ xy, const uint32_t* __restrict__ key, size_t n, uint32_t B,
                                               uint32_t chunk, uint32_t nchunk, const uint32_t* __restrict__ base, double2* oxy,
                                               uint32_t* oidx, double4* oaos) {
  extern __shared__ uint32_t h[];
  for (uint32_t k = threadIdx.x; k < B; k += PT) h[k] = base[(size_t)k * nchunk + blockIdx.x];
  __syncthreads();
  size_t first = (size_t)blockIdx.x * chunk, last = first + chunk < n ? first + chunk : n;
#pragma unroll 4
  for (size_t i = first + threadIdx.x; i < last; i += PT) {
    const double2 v = xy[i];
    const uint32_t dst = atomicAdd(&h[key[i]], 1u);
    if (MODE == 0) {
      oxy[dst] = v;
      oidx[dst] = (uint32_t)i;
    } else if (MODE == 1) {
      oaos[dst] = make_double4(v.x, v.y, __hiloint2double(0, (int)i), 0.0);
    } else {  // 16-byte record: binary32 coordinates, index, key
      reinterpret_cast<float4*>(oaos)[dst] =
          make_float4((float)v.x, (float)v.y, __uint_as_float((uint32_t)i), __uint_as_float(key[i]));
    }
  }
}
// staged: sub-chunks of SUB records are ordered by bucket in LDS (count, scan, place), then written so that consecutive
// lanes store consecutive slots of a run
constexpr int SUB = 4096;
template <int MODE>
__global__ __launch_bounds__(PT) void k_scatter_staged(const double2* __restrict__ xy, const uint32_t* __restrict__ key, size_t n,
                                                      uint32_t B, uint32_t chunk, uint32_t nchunk, const uint32_t* __restrict__ base,
                                                      double2* oxy, uint32_t* oidx, double4* oaos) {
  extern __shared__ uint32_t sm[];
  uint32_t* cur = sm;            // [B] global cursor of this chunk per bucket
  uint32_t* lcnt = sm + B;       // [B] local count / start within the sub-chunk
  uint32_t* skey = lcnt + B;     // [SUB] bucket of staged record
  uint32_t* sidx = skey + SUB;   // [SUB]
  uint32_t* sdst = sidx + SUB;   // [SUB] global slot of staged record
  double2* sxy = reinterpret_cast<double2*>(sdst + SUB);  // [SUB]
  __shared__ uint32_t wsum[PT / 64];
  for (uint32_t k = threadIdx.x; k < B; k += PT) cur[k] = base[(size_t)k * nchunk + blockIdx.x];
  size_t first = (size_t)blockIdx.x * chunk, last = first + chunk < n ? first + chunk : n;
  for (size_t s0 = first; s0 < last; s0 += SUB) {
    for (uint32_t k = threadIdx.x; k < B; k += PT) lcnt[k] = 0;
    __syncthreads();
    constexpr int PER = SUB / PT;
    uint32_t kk[PER], rr[PER];
    double2 vv[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
      size_t i = s0 + (size_t)u * PT + threadIdx.x;
      kk[u] = 0xFFFFFFFFu;
      if (i < last) {
        kk[u] = key[i];
        vv[u] = xy[i];
        rr[u] = atomicAdd(&lcnt[kk[u]], 1u);
      }
    }
    __syncthreads();
    // exclusive scan of lcnt over B (B <= 8192): thread t owns ceil(B/PT) consecutive entries
    const uint32_t per = (B + PT - 1) / PT;
    uint32_t loc = 0;
    for (uint32_t k = 0; k < per; k++) { uint32_t b = threadIdx.x * per + k; if (b < B) loc += lcnt[b]; }
    uint32_t inc = loc;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t pre = inc - loc;
    for (int k = 0; k < w; k++) pre += wsum[k];
    for (uint32_t k = 0; k < per; k++) {
      uint32_t b = threadIdx.x * per + k;
      if (b < B) { uint32_t c = lcnt[b]; lcnt[b] = pre; pre += c; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; u++) {
      if (kk[u] == 0xFFFFFFFFu) continue;
      const uint32_t slot = lcnt[kk[u]] + rr[u];
      size_t i = s0 + (size_t)u * PT + threadIdx.x;
      sxy[slot] = vv[u];
      sidx[slot] = (uint32_t)i;
      sdst[slot] = cur[kk[u]] + rr[u];
      skey[slot] = kk[u];
    }
    __syncthreads();
    const uint32_t m = (uint32_t)((last - s0) < SUB ? (last - s0) : SUB);
    for (uint32_t j = threadIdx.x; j < m; j += PT) {
      const uint32_t dst = sdst[j];
      if (MODE == 0) {
        oxy[dst] = sxy[j];
        oidx[dst] = sidx[j];
      } else {
        oaos[dst] = make_double4(sxy[j].x, sxy[j].y, __hiloint2double(0, (int)sidx[j]), 0.0);
      }
    }
    __syncthreads();
    // advance the chunk's cursors by this sub-chunk's counts: lcnt holds starts; count_b = start_{b+1} - start_b
    for (uint32_t b = threadIdx.x; b < B; b += PT) {
      const uint32_t nxt = (b + 1 < B) ? lcnt[b + 1] : m;
      cur[b] += nxt - lcnt[b];
    }
    __syncthreads();
  }
}

// output direction: records (ord, word) grouped by windows of 2^15 ords; write labels[ord] (4 B), a[ord], b[ord] (1 B each)
constexpr int OWSH = 15;
__global__ __launch_bounds__(256) void k_window_direct(const uint2* __restrict__ rec, size_t n, int32_t* lab, uint8_t* a, uint8_t* b) {
  const uint32_t wgs = (1 << OWSH) / 4096;
  const uint32_t bk = (blockIdx.x / (8u * wgs)) * 8u + (blockIdx.x & 7u);
  const uint32_t wg = (blockIdx.x >> 3) % wgs;
  const size_t lo = (size_t)bk << OWSH;
  if (lo >= n) return;
  const size_t cnt = (n - lo) < (1u << OWSH) ? (n - lo) : (1u << OWSH);
#pragma unroll 4
  for (int k = 0; k < 16; k++) {
    size_t j = (size_t)wg * 4096 + (size_t)k * 256 + threadIdx.x;
    if (j >= cnt) break;
    uint2 v = rec[lo + j];
    lab[v.x] = (int32_t)(v.y >> 2);
    a[v.x] = v.y & 1;
    b[v.x] = (v.y >> 1) & 1;
  }
}
// LDS-staged: one workgroup per window of 2^13 ords (8192 x 6 B = 48 KB), records grouped by those windows
constexpr int SWSH = 13;
__global__ __launch_bounds__(512) void k_window_staged(const uint2* __restrict__ rec, size_t n, int32_t* lab, uint8_t* a, uint8_t* b) {
  __shared__ int32_t sl[1 << SWSH];
  __shared__ uint8_t sa[1 << SWSH], sb[1 << SWSH];
  const size_t lo = (size_t)blockIdx.x << SWSH;
  if (lo >= n) return;
  const uint32_t cnt = (uint32_t)((n - lo) < (1u << SWSH) ? (n - lo) : (1u << SWSH));
  for (uint32_t j = threadIdx.x; j < cnt; j += 512) {
    uint2 v = rec[lo + j];
    uint32_t o = v.x - (uint32_t)lo;
    sl[o] = (int32_t)(v.y >> 2);
    sa[o] = v.y & 1;
    sb[o] = (v.y >> 1) & 1;
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < cnt; j += 512) lab[lo + j] = sl[j];
  for (uint32_t j = threadIdx.x; j < cnt / 4; j += 512) {
    reinterpret_cast<uint32_t*>(a + lo)[j] = reinterpret_cast<uint32_t*>(sa)[j];
    reinterpret_cast<uint32_t*>(b + lo)[j] = reinterpret_cast<uint32_t*>(sb)[j];
  }
}

int main() {
  const size_t n = 10000000;
  std::mt19937 rng(7);
  std::vector<double2> hxy(n);
  std::vector<uint32_t> hperm(n);
  for (size_t i = 0; i < n; i++) { hxy[i] = make_double2((double)i, 0.5); hperm[i] = (uint32_t)i; }
  for (size_t i = n - 1; i > 0; i--) std::swap(hperm[i], hperm[rng() % (i + 1)]);
  double2 *xy, *oxy; double4* oaos; uint32_t *key, *idx, *oidx, *counts, *perm;
  CK(hipMalloc(&xy, n * 16)); CK(hipMalloc(&oxy, n * 16)); CK(hipMalloc(&oaos, n * 32));
  CK(hipMalloc(&key, n * 4)); CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&oidx, n * 4)); CK(hipMalloc(&perm, n * 4));
  CK(hipMalloc(&counts, (size_t)8192 * 2048 * 4));
  CK(hipMemcpy(xy, hxy.data(), n * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(perm, hperm.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& f) {
    for (int it = 0; it < 3; it++) f();
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; it++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    printf("%-40s %.3f ms\n", name, ms / 10); fflush(stdout);
  };
  timeit("copy 16+4 B", [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, xy, perm, oxy, oidx, n); });
  timeit("gather 16 B by random index", [&] { hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, 0, xy, perm, oxy, n); });
  std::vector<uint32_t> hkey(n), hc;
  for (uint32_t B : {1024u, 4096u, 8192u}) {
    for (size_t i = 0; i < n; i++) hkey[i] = rng() % B;
    CK(hipMemcpy(key, hkey.data(), n * 4, hipMemcpyHostToDevice));
    for (uint32_t chunk : {10240u, 40960u}) {
      const uint32_t nchunk = (uint32_t)((n + chunk - 1) / chunk);
      hipLaunchKernelGGL(k_hist, dim3(nchunk), dim3(PT), B * 4, 0, key, n, B, chunk, nchunk, counts);
      hc.resize((size_t)B * nchunk);
      CK(hipMemcpy(hc.data(), counts, hc.size() * 4, hipMemcpyDeviceToHost));
      uint32_t run = 0;
      for (auto& c : hc) { uint32_t t = c; c = run; run += t; }
      CK(hipMemcpy(counts, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
      char nm[128];
      snprintf(nm, sizeof nm, "hist   B=%u chunk=%u", B, chunk);
      timeit(nm, [&] { hipLaunchKernelGGL(k_hist, dim3(nchunk), dim3(PT), B * 4, 0, key, n, B, chunk, nchunk, counts + (size_t)B * nchunk); });
      snprintf(nm, sizeof nm, "soa    B=%u chunk=%u", B, chunk);
      timeit(nm, [&] { hipLaunchKernelGGL(k_scatter<0>, dim3(nchunk), dim3(PT), B * 4, 0, xy, key, n, B, chunk, nchunk, counts, oxy, oidx, oaos); });
      snprintf(nm, sizeof nm, "aos32  B=%u chunk=%u", B, chunk);
      timeit(nm, [&] { hipLaunchKernelGGL(k_scatter<1>, dim3(nchunk), dim3(PT), B * 4, 0, xy, key, n, B, chunk, nchunk, counts, oxy, oidx, oaos); });
      snprintf(nm, sizeof nm, "aos16  B=%u chunk=%u", B, chunk);
      timeit(nm, [&] { hipLaunchKernelGGL(k_scatter<2>, dim3(nchunk), dim3(PT), B * 4, 0, xy, key, n, B, chunk, nchunk, counts, oxy, oidx, oaos); });
      const size_t lds = (size_t)B * 8 + (size_t)SUB * 12 + (size_t)SUB * 16 + 64;
      if (lds > 64 * 1024) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter_staged<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter_staged<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      }
      snprintf(nm, sizeof nm, "staged soa   B=%u chunk=%u", B, chunk);
      timeit(nm, [&] { hipLaunchKernelGGL(k_scatter_staged<0>, dim3(nchunk), dim3(PT), lds, 0, xy, key, n, B, chunk, nchunk, counts, oxy, oidx, oaos); });
      snprintf(nm, sizeof nm, "staged aos32 B=%u chunk=%u", B, chunk);
      timeit(nm, [&] { hipLaunchKernelGGL(k_scatter_staged<1>, dim3(nchunk), dim3(PT), lds, 0, xy, key, n, B, chunk, nchunk, counts, oxy, oidx, oaos); });
    }
  }
  // output direction: records grouped by windows, random order inside a window
  {
    std::vector<uint2> hrec(n);
    for (int sh : {OWSH, SWSH}) {
      for (size_t lo = 0; lo < n; lo += (size_t)1 << sh) {
        size_t cnt = std::min(n - lo, (size_t)1 << sh);
        for (size_t j = 0; j < cnt; j++) hrec[lo + j] = make_uint2((uint32_t)(lo + j), (uint32_t)(rng() & 0xFFFF));
        for (size_t j = cnt - 1; j > 0; j--) std::swap(hrec[lo + j], hrec[lo + rng() % (j + 1)]);
      }
      uint2* rec = reinterpret_cast<uint2*>(oaos);
      CK(hipMemcpy(rec, hrec.data(), n * 8, hipMemcpyHostToDevice));
      int32_t* lab = reinterpret_cast<int32_t*>(oidx);
      uint8_t* a = reinterpret_cast<uint8_t*>(oxy);
      uint8_t* b = a + n;
      if (sh == OWSH) {
        const unsigned OB = (unsigned)((n + (1 << OWSH) - 1) >> OWSH);
        const unsigned nb = ((OB + 7) / 8) * 8 * ((1 << OWSH) / 4096);
        timeit("window direct 2^15 (4+1+1 B)", [&] { hipLaunchKernelGGL(k_window_direct, dim3(nb), dim3(256), 0, 0, rec, n, lab, a, b); });
      } else {
        const unsigned nb = (unsigned)((n + (1 << SWSH) - 1) >> SWSH);
        timeit("window LDS-staged 2^13 (4+1+1 B)", [&] { hipLaunchKernelGGL(k_window_staged, dim3(nb), dim3(512), 0, 0, rec, n, lab, a, b); });
      }
    }
  }
  return 0;
}
