// Map of the fill rate by WHERE two streams lie: one large allocation, streams of 4 GiB at multiples of 8 GiB; the rate
// of every pair (x, y) written in lockstep, and of single streams.  Looking for the rule behind the fast (7.0 TB/s)
// and slow (5.5-5.9 TB/s) placements of lockstep-written columns (profiles/r04_ubench_fill_candidates.txt).
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_fill_map scripts/ubench_fill_map.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef long long ll2 __attribute__((ext_vector_type(2)));

template <int ARRAYS>
__global__ __launch_bounds__(256) void k_fill(ll2 *__restrict__ a, ll2 *__restrict__ b, ll2 *__restrict__ c, uint64_t pairs,
                                              uint64_t per_wg) {
  const uint64_t lo = (uint64_t)blockIdx.x * per_wg, hi = lo + per_wg < pairs ? lo + per_wg : pairs;
  ll2 v;
  v.x = (long long)blockIdx.x;
  v.y = (long long)threadIdx.x;
  for (uint64_t q = lo + threadIdx.x; q < hi; q += 256) {
    __builtin_nontemporal_store(v, a + q);
    if (ARRAYS > 1) __builtin_nontemporal_store(v, b + q);
    if (ARRAYS > 2) __builtin_nontemporal_store(v, c + q);
  }
}

static hipEvent_t e0, e1;
template <int ARRAYS> static double rate(char *a, char *b, char *c, uint64_t len) {
  const uint64_t pairs = len / 16, per_wg = 16384;
  const unsigned grid = (unsigned)((pairs + per_wg - 1) / per_wg);
  float best = 1e30f;
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0, 0);
    k_fill<ARRAYS><<<grid, 256>>>((ll2 *)a, (ll2 *)b, (ll2 *)c, pairs, per_wg);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return (double)ARRAYS * len / best / 1e9;
}

int main(int argc, char **argv) {
  const uint64_t GiB = 1ull << 30;
  uint64_t total = (argc > 1 ? atoll(argv[1]) : 208) * GiB;
  const uint64_t L = 4 * GiB, step = 8 * GiB;
  char *base = nullptr;
  while (hipMalloc(&base, total) != hipSuccess) {
    (void)hipGetLastError();
    total -= 16 * GiB;
    if (total < 64 * GiB) return 1;
  }
  const int n = (int)((total - L) / step) + 1;
  printf("one allocation of %llu GiB at %p; streams of 4 GiB at multiples of 8 GiB (%d positions)\n",
         (unsigned long long)(total / GiB), (void *)base, n);
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  printf("single stream by position (TB/s):\n");
  for (int i = 0; i < n; i++) printf(" %4.2f", rate<1>(base + i * step, nullptr, nullptr, L));
  printf("\ntwo streams in lockstep, row = position of the first, column = of the second (GiB / 8; TB/s):\n     ");
  for (int j = 0; j < n; j++) printf(" %4d", j * 8);
  printf("\n");
  for (int i = 0; i < n; i++) {
    printf("%4d ", i * 8);
    for (int j = 0; j < n; j++) {
      if (j <= i)
        printf("     ");
      else
        printf(" %4.2f", rate<2>(base + i * step, base + j * step, nullptr, L));
    }
    printf("\n");
  }
  printf("three streams in lockstep at (x, x + d, x + 2d), d = 8 / 16 / 24 / 32 GiB, by x (TB/s):\n");
  for (int d = 1; d <= 4; d++) {
    printf("d = %2d GiB:", d * 8);
    for (int i = 0; i + 2 * d < n; i++) printf(" %4.2f", rate<3>(base + i * step, base + (i + d) * step, base + (i + 2 * d) * step, L));
    printf("\n");
  }
  printf("three adjacent streams (x, x + 4, x + 8 GiB) by x in steps of 8 GiB:\n");
  for (int i = 0; i + 1 < n; i++) printf(" %4.2f", rate<3>(base + i * step, base + i * step + L, base + i * step + 2 * L, L));
  printf("\n");
  return 0;
}
