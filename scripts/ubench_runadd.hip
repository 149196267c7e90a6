// Checks gg::run_add / gg::wave_has_runs (duckdb_pgq_amd/csrc/gg_runs.h) lane by lane against a sequential count,
// on key sequences from fully sorted to random.  Memory-safe by construction: keys are masked to the 64 LDS
// counters, every thread writes only its own output slot.
//   hipcc --offload-arch=gfx950 -O3 -I duckdb_pgq_amd/csrc scripts/ubench_runadd.hip -o build/ubench_runadd && build/ubench_runadd
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gg_runs.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

constexpr int STEPS = 16;  // 64-entry steps per wave

// one wave per block: ranks of its STEPS * 64 entries (out_pos), the counters afterwards (out_cnt), its decision
__global__ __launch_bounds__(64) void k_check(const uint32_t *__restrict__ keys, const uint8_t *__restrict__ valid,
                                              uint32_t *__restrict__ out_pos, uint32_t *__restrict__ out_cnt,
                                              uint32_t *__restrict__ out_cnt2, uint32_t *__restrict__ out_runs) {
  __shared__ uint32_t cur[64], cnt[64];
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * STEPS * 64;
  cur[lane] = 0;
  cnt[lane] = 0;
  __builtin_amdgcn_wave_barrier();
  const bool runs = gg::wave_has_runs(keys[base + lane] & 63u, valid[base + lane] != 0, lane);
  for (int s = 0; s < STEPS; s++) {
    const size_t i = base + (size_t)s * 64 + lane;
    const uint32_t k = keys[i] & 63u;
    const bool v = valid[i] != 0;
    out_pos[i] = gg::run_add<true>(cur, k, v, lane);
    gg::run_add<false>(cnt, k, v, lane);
  }
  __builtin_amdgcn_wave_barrier();
  out_cnt[blockIdx.x * 64 + lane] = cur[lane];
  out_cnt2[blockIdx.x * 64 + lane] = cnt[lane];
  if (lane == 0) out_runs[blockIdx.x] = runs;
}

int main() {
  const int waves = 4096;
  const size_t n = (size_t)waves * STEPS * 64;
  std::vector<uint32_t> keys(n);
  std::vector<uint8_t> valid(n);
  uint64_t x = 88172645463325252ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int w = 0; w < waves; w++) {
    // wave w: runs of mean length 1 (random) .. 200 (one key per several steps), 0..30 % invalid lanes
    const uint32_t mean = (w % 8 == 0) ? 1 : 1 + (uint32_t)(rnd() % 200);
    const uint32_t inval = (uint32_t)(rnd() % 4) * 10;
    uint32_t key = (uint32_t)rnd() & 63u, left = 1 + (uint32_t)(rnd() % (2 * mean));
    for (int i = 0; i < STEPS * 64; i++) {
      if (left == 0) {
        key = (w % 3 == 0) ? (key + 1) & 63u : (uint32_t)rnd() & 63u;  // sorted or not
        left = 1 + (uint32_t)(rnd() % (2 * mean));
      }
      left--;
      keys[(size_t)w * STEPS * 64 + i] = key | ((uint32_t)rnd() << 6);  // junk above the masked bits
      valid[(size_t)w * STEPS * 64 + i] = (rnd() % 100) >= inval;
    }
    if (w % 16 == 5)  // a ragged tail: everything past a point is invalid
      for (int i = (int)(rnd() % (STEPS * 64)); i < STEPS * 64; i++) valid[(size_t)w * STEPS * 64 + i] = 0;
  }
  uint32_t *d_keys, *d_pos, *d_cnt, *d_cnt2, *d_runs;
  uint8_t *d_valid;
  CHECK(hipMalloc(&d_keys, n * 4));
  CHECK(hipMalloc(&d_valid, n));
  CHECK(hipMalloc(&d_pos, n * 4));
  CHECK(hipMalloc(&d_cnt, waves * 64 * 4));
  CHECK(hipMalloc(&d_cnt2, waves * 64 * 4));
  CHECK(hipMalloc(&d_runs, waves * 4));
  CHECK(hipMemcpy(d_keys, keys.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_valid, valid.data(), n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_check, dim3(waves), dim3(64), 0, 0, d_keys, d_valid, d_pos, d_cnt, d_cnt2, d_runs);
  CHECK(hipDeviceSynchronize());
  std::vector<uint32_t> pos(n), cnt(waves * 64), cnt2(waves * 64), runs(waves);
  CHECK(hipMemcpy(pos.data(), d_pos, n * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(cnt.data(), d_cnt, waves * 64 * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(cnt2.data(), d_cnt2, waves * 64 * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(runs.data(), d_runs, waves * 4, hipMemcpyDeviceToHost));
  size_t bad_pos = 0, bad_cnt = 0, nruns = 0;
  for (int w = 0; w < waves; w++) {
    uint32_t c[64] = {0};
    for (int i = 0; i < STEPS * 64; i++) {
      const size_t at = (size_t)w * STEPS * 64 + i;
      if (!valid[at]) continue;
      const uint32_t k = keys[at] & 63u;
      if (pos[at] != c[k]) bad_pos++;
      c[k]++;
    }
    for (int k = 0; k < 64; k++) bad_cnt += (cnt[w * 64 + k] != c[k]) + (cnt2[w * 64 + k] != c[k]);
    nruns += runs[w];
  }
  printf("{\"entries\": %zu, \"bad_ranks\": %zu, \"bad_counts\": %zu, \"waves_with_runs\": %zu, \"waves\": %d}\n", n,
         bad_pos, bad_cnt, nruns, waves);
  return bad_pos || bad_cnt ? 1 : 0;
}
