// Is the fill rate a stable property of WHERE a result's columns lie?  Eight candidate placements of three 5.7 GB
// columns (one materialised SF100 part) are held at the same time and filled round-robin, four rounds: if a
// placement's rate repeats from round to round and differs between placements, a library can pick its result blocks
// by a probe fill; if all move together, the mode belongs to the process and no choice of blocks helps.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_fill_candidates scripts/ubench_fill_candidates.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef long long ll2 __attribute__((ext_vector_type(2)));

// groups > 1: workgroup i writes chunk (i % groups) * ceil(n / groups) + i / groups — the workgroups resident at any
// moment are then spread over `groups` places of every column instead of moving through it as one window
__global__ __launch_bounds__(256) void k_fill3(ll2 *__restrict__ a, ll2 *__restrict__ b, ll2 *__restrict__ c, uint64_t pairs,
                                               uint64_t per_wg, uint32_t groups) {
  uint64_t chunk = blockIdx.x;
  if (groups > 1) {
    const uint64_t per_group = (gridDim.x + groups - 1) / groups;
    chunk = (uint64_t)(blockIdx.x % groups) * per_group + blockIdx.x / groups;
    if (chunk >= gridDim.x || blockIdx.x / groups >= per_group) return;
  }
  const uint64_t lo = chunk * per_wg, hi = lo + per_wg < pairs ? lo + per_wg : pairs;
  if (lo >= pairs) return;
  ll2 v;
  v.x = (long long)blockIdx.x;
  v.y = (long long)threadIdx.x;
  for (uint64_t q = lo + threadIdx.x; q < hi; q += 256) {
    __builtin_nontemporal_store(v, a + q);
    __builtin_nontemporal_store(v, b + q);
    __builtin_nontemporal_store(v, c + q);
  }
}

int main(int argc, char **argv) {
  const int n_cand = argc > 1 ? atoi(argv[1]) : 8;
  const uint64_t rows = 709188913ull, pairs = rows / 2, bytes = pairs * 16;  // one SF100 part of 18
  const uint64_t per_wg = 16384;
  const unsigned grid = (unsigned)((pairs + per_wg - 1) / per_wg);
  ll2 *col[16][3];
  for (int i = 0; i < n_cand; i++)
    for (int c = 0; c < 3; c++)
      if (hipMalloc(&col[i][c], bytes) != hipSuccess) {
        printf("alloc failed at candidate %d\n", i);
        return 1;
      }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const uint32_t group_list[] = {1, 1, 8, 64, 512, 4096};
  for (int round = 0; round < 6; round++) {
    const uint32_t groups = group_list[round];
    // (a grid padded to a multiple of `groups`, so that the mapping is a bijection onto the chunks)
    const unsigned g = groups > 1 ? (unsigned)(((grid + groups - 1) / groups) * groups) : grid;
    printf("round %d (groups %4u):", round, groups);
    for (int i = 0; i < n_cand; i++) {
      (void)hipEventRecord(e0, 0);
      k_fill3<<<g, 256>>>(col[i][0], col[i][1], col[i][2], pairs, per_wg, groups);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("  %5.2f", 3.0 * bytes / ms / 1e9);
    }
    printf("  TB/s\n");
  }
  // the same columns taken crosswise (column c of candidate i + c): is it the triple or the single block that decides?
  printf("crosswise:");
  for (int i = 0; i < n_cand; i++) {
    (void)hipEventRecord(e0, 0);
    k_fill3<<<grid, 256>>>(col[i][0], col[(i + 1) % n_cand][1], col[(i + 2) % n_cand][2], pairs, per_wg, 1);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %5.2f", 3.0 * bytes / ms / 1e9);
  }
  printf("  TB/s\n");
  for (int i = 0; i < n_cand; i++) printf("cand %d: %p %p %p\n", i, (void *)col[i][0], (void *)col[i][1], (void *)col[i][2]);
  return 0;
}
