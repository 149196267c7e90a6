// micro-benchmark: does the speed of a fill depend on WHERE in device memory the arrays lie?  Per trial: hold a dummy
// allocation of trial x 32 GB, allocate three 8.5 GB columns behind it, time the three-column fill, free everything.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_fill_place scripts/ubench_fill_place.hip
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
  const uint64_t rows = 1063072142ull, pairs = rows / 2;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int round = 0; round < 2; round++)
    for (int trial = 0; trial < 7; trial++) {
      void *dummy = nullptr;
      if (trial && hipMalloc(&dummy, (uint64_t)trial * (32ull << 30)) != hipSuccess) {
        printf("dummy alloc of %d x 32 GB failed\n", trial);
        continue;
      }
      ll2 *col[3];
      bool ok = true;
      for (auto &p : col) ok = ok && hipMalloc(&p, pairs * 16) == hipSuccess;
      if (ok) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
          (void)hipEventRecord(e0, 0);
          k_fill3<<<(unsigned)((pairs + 16383) / 16384), 256>>>(col[0], col[1], col[2], pairs, 16384);
          (void)hipEventRecord(e1, 0);
          (void)hipEventSynchronize(e1);
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (rep && ms < best) best = ms;
        }
        printf("behind %3d GB: %.3f ms  %.2f TB/s   (columns at %p %p %p)\n", trial * 32, best,
               (double)pairs * 48.0 / best / 1e9, (void *)col[0], (void *)col[1], (void *)col[2]);
      }
      for (auto &p : col) (void)hipFree(p);
      if (dummy) (void)hipFree(dummy);
    }
  return 0;
}
