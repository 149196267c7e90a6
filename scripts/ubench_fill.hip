// micro-benchmark: what do plain fills reach on this box?  The ceiling of k_mat_mid2, which writes three int64
// columns of 1.06 G rows (25.5 GB at SF10) and nothing else of size.  Variants: one array or three arrays at the same
// offsets (the materialised result's shape); plain or nontemporal 16-byte stores; a KB per wave instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_fill scripts/ubench_fill.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef long long ll2 __attribute__((ext_vector_type(2)));

// each workgroup of 256 lanes writes `per_wg` pairs of every array, contiguous, 4 KB per step
// SHIFT: every store instruction starts `SHIFT` pairs (16 bytes each) past a 1 KB boundary, so its first and last
// 128-byte lines are shared with the neighbouring instructions (the shape of rows that start anywhere)
template <int ARRAYS, bool NT, int SHIFT = 0>
__global__ __launch_bounds__(256) void k_fill(ll2 *__restrict__ a, ll2 *__restrict__ b, ll2 *__restrict__ c, uint64_t pairs,
                                              uint64_t per_wg) {
  a += SHIFT;
  b += SHIFT;
  c += SHIFT;
  pairs -= 64;
  const uint64_t lo = (uint64_t)blockIdx.x * per_wg, hi = lo + per_wg < pairs ? lo + per_wg : pairs;
  ll2 v;
  v.x = (long long)blockIdx.x;
  v.y = (long long)threadIdx.x;
  for (uint64_t q = lo + threadIdx.x; q < hi; q += 256) {
    if (NT) {
      __builtin_nontemporal_store(v, a + q);
      if (ARRAYS > 1) __builtin_nontemporal_store(v, b + q);
      if (ARRAYS > 2) __builtin_nontemporal_store(v, c + q);
    } else {
      a[q] = v;
      if (ARRAYS > 1) b[q] = v;
      if (ARRAYS > 2) c[q] = v;
    }
  }
}

template <int ARRAYS, bool NT, int SHIFT = 0>
static void run(const char *name, ll2 *a, ll2 *b, ll2 *c, uint64_t pairs, uint64_t per_wg) {
  const unsigned grid = (unsigned)((pairs + per_wg - 1) / per_wg);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f, sum = 0;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(e0, 0);
    k_fill<ARRAYS, NT, SHIFT><<<grid, 256>>>(a, b, c, pairs, per_wg);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep) {
      best = ms < best ? ms : best;
      sum += ms;
    }
  }
  const double bytes = (double)pairs * 16.0 * ARRAYS;
  printf("%-34s per-wg %8llu pairs  best %7.3f ms  mean %7.3f ms  %5.2f TB/s (best)\n", name, (unsigned long long)per_wg, best,
         sum / 4, bytes / best / 1e9);
}

int main() {
  const uint64_t rows = 1063072142ull, pairs = rows / 2;  // SF10's 2-hop rows
  ll2 *a, *b, *c;
  if (hipMalloc(&a, pairs * 16) != hipSuccess || hipMalloc(&b, pairs * 16) != hipSuccess || hipMalloc(&c, pairs * 16) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  for (uint64_t per_wg : {1024ull, 8192ull, 65536ull}) {
    run<1, false>("1 array, plain", a, b, c, pairs, per_wg);
    run<1, true>("1 array, nt", a, b, c, pairs, per_wg);
    run<3, false>("3 arrays, plain", a, b, c, pairs, per_wg);
    run<3, true>("3 arrays, nt", a, b, c, pairs, per_wg);
    run<3, true, 1>("3 arrays, nt, +16 B", a, b, c, pairs, per_wg);
    run<3, true, 3>("3 arrays, nt, +48 B", a, b, c, pairs, per_wg);
    run<3, false, 1>("3 arrays, plain, +16 B", a, b, c, pairs, per_wg);
  }
  (void)hipFree(a);
  (void)hipFree(b);
  (void)hipFree(c);
  return 0;
}
