// micro-benchmark: rocPRIM radix_sort_pairs / radix_sort_keys on 10M u32 keys (23 significant bits)
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
int main() {
  const size_t n = 10000000;
  std::vector<uint32_t> hk(n), hv(n);
  std::mt19937 rng(1);
  for (size_t i = 0; i < n; i++) { hk[i] = rng() % 4600000; hv[i] = (uint32_t)i; }
  uint32_t *k, *ko, *v, *vo; void* tmp; size_t tb = 0;
  hipMalloc(&k, n * 4); hipMalloc(&ko, n * 4); hipMalloc(&v, n * 4); hipMalloc(&vo, n * 4);
  hipMemcpy(k, hk.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), n * 4, hipMemcpyHostToDevice);
  for (int bits : {23, 32}) {
    rocprim::radix_sort_pairs(nullptr, tb, k, ko, v, vo, n, 0, bits, 0);
    hipMalloc(&tmp, tb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; it++) rocprim::radix_sort_pairs(tmp, tb, k, ko, v, vo, n, 0, bits, 0);
    hipEventRecord(a); for (int it = 0; it < 10; it++) rocprim::radix_sort_pairs(tmp, tb, k, ko, v, vo, n, 0, bits, 0); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("radix_sort_pairs u32/u32 n=%zu bits=%d: %.3f ms (tmp %zu MB)\n", n, bits, ms / 10, tb >> 20);
    hipFree(tmp);
  }
  return 0;
}
