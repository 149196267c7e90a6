// What decides the fill rate of three columns written in lockstep?  scripts/ubench_fill_candidates.hip showed it is a
// stable property of the TRIPLE of blocks (back-to-back allocations mostly 5.9 TB/s, the same blocks taken crosswise
// mostly 7.0).  Here: one 60 GiB allocation, three streams of 5.5 GiB at base + {0, D, 2D} for a list of distances D.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_fill_offsets scripts/ubench_fill_offsets.hip
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

int main() {
  const uint64_t GiB = 1ull << 30, MiB = 1ull << 20, KiB = 1ull << 10;
  const uint64_t len = 11 * (GiB / 2);  // 5.5 GiB per stream: the pool's block for one SF100 part column
  const uint64_t total = 60 * GiB;
  char *base;
  if (hipMalloc(&base, total) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  const uint64_t pairs = len / 16, per_wg = 16384;
  const unsigned grid = (unsigned)((pairs + per_wg - 1) / per_wg);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const uint64_t Ds[] = {len,           len + 4 * KiB,   len + 64 * KiB,  len + 1 * MiB,   len + 2 * MiB,  len + 6 * MiB,
                         len + 16 * MiB, len + 32 * MiB,  len + 64 * MiB,  len + 128 * MiB, len + 192 * MiB, len + 256 * MiB,
                         len + 384 * MiB, 6 * GiB,        6 * GiB + 256 * MiB, 6 * GiB + 512 * MiB, 7 * GiB, 7 * GiB + 512 * MiB,
                         8 * GiB,       9 * GiB,         10 * GiB,        11 * GiB,        12 * GiB,       16 * GiB,
                         16 * GiB + 512 * MiB, 17 * GiB, 22 * GiB,        27 * GiB};
  for (uint64_t D : Ds) {
    if (2 * D + len > total) continue;
    float ms3 = 1e30f, ms2 = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0, 0);
      k_fill<3><<<grid, 256>>>((ll2 *)base, (ll2 *)(base + D), (ll2 *)(base + 2 * D), pairs, per_wg);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      ms3 = ms < ms3 ? ms : ms3;
      (void)hipEventRecord(e0, 0);
      k_fill<2><<<grid, 256>>>((ll2 *)base, (ll2 *)(base + D), (ll2 *)nullptr, pairs, per_wg);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      ms2 = ms < ms2 ? ms : ms2;
    }
    printf("D = %8.3f GiB (%6llu MiB past 5.5 GiB): three streams %5.2f TB/s, two streams %5.2f TB/s\n", (double)D / GiB,
           (unsigned long long)((D - len) / MiB), 3.0 * len / ms3 / 1e9, 2.0 * len / ms2 / 1e9);
  }
  return 0;
}
