// micro-benchmark: three 8.5 GB columns filled side by side (k_mat_mid2's shape) — does the time depend on where the
// columns lie relative to each other?  One 30 GB allocation, columns at A, A + L + s, A + 2 (L + s) for several skews s.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_fill_skew scripts/ubench_fill_skew.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef long long ll2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_fill3(ll2 *__restrict__ a, ll2 *__restrict__ b, ll2 *__restrict__ c, uint64_t pairs,
                                               uint64_t per_wg) {
  const uint64_t lo = (uint64_t)blockIdx.x * per_wg, hi = lo + per_wg < pairs ? lo + per_wg : pairs;
  ll2 v;
  v.x = (long long)blockIdx.x;
  v.y = (long long)threadIdx.x;
  for (uint64_t q = lo + threadIdx.x; q < hi; q += 256) {
    __builtin_nontemporal_store(v, a + q);
    __builtin_nontemporal_store(v, b + q);
    __builtin_nontemporal_store(v, c + q);
  }
}

int main() {
  const uint64_t rows = 1063072142ull, pairs = rows / 2, L = ((pairs * 16 + (2u << 20) - 1) >> 21) << 21;
  char *base;
  if (hipMalloc(&base, 3 * L + (64u << 20)) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const uint64_t skews[] = {0, 256, 4096, 4096 + 256, 65536 + 4096, (1u << 20) + 65536 + 4096 + 256, (2u << 20), (6u << 20) + 4096};
  for (int pass = 0; pass < 2; pass++)
    for (uint64_t s : skews) {
      ll2 *a = (ll2 *)base, *b = (ll2 *)(base + L + s), *c = (ll2 *)(base + 2 * (L + s));
      float best = 1e30f;
      for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
        k_fill3<<<(unsigned)((pairs + 16383) / 16384), 256>>>(a, b, c, pairs, 16384);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("skew %9llu B: %.3f ms  %.2f TB/s\n", (unsigned long long)s, best, (double)pairs * 48.0 / best / 1e9);
    }
  (void)hipFree(base);
  return 0;
}
