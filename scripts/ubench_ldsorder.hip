// micro-test: does one ds_add_rtn_u32 hand its return values to the lanes that hit the same LDS word in increasing
// lane order?  (The stable ranking kernels want that: position = atomicAdd(&cursor[bucket], 1).)  Every wave owns a
// row of K counters; lanes pick pseudo-random counters under a pseudo-random exec mask; the expected return value is
// counter-before + number of lower active lanes with the same counter (computed with ballots).  Also times the
// atomic against the ballot match loop it would replace.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>  // 0: verify order; 1: time atomic rank; 2: time ballot rank
__global__ __launch_bounds__(512) void k(uint32_t K, uint32_t bits, int rounds, uint32_t seed, unsigned long long *bad,
                                         uint32_t *sink) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *row = lds + wave * K;
  for (uint32_t i = lane; i < K; i += 64) row[i] = 0;
  __builtin_amdgcn_wave_barrier();
  const uint64_t lane_lt = (1ULL << lane) - 1ULL;
  uint32_t acc = 0;
  unsigned long long nbad = 0;
  for (int r = 0; r < rounds; r++) {
    const uint32_t h = mix(seed + (blockIdx.x * 8 + wave) * 0x9E3779B9u + r * 64 + lane);
    const uint32_t a = (h >> 8) % K;
    const bool valid = MODE != 0 || (h & 7u) != 0;  // verify: an eighth of the lanes sit out
    if (MODE == 0) {
      volatile uint32_t *vr = row;
      uint32_t before = 0;
      if (valid) before = vr[a];
      uint64_t m = __ballot(valid);
      for (uint32_t b = 0; b < bits; b++) {
        const uint64_t bb = __ballot((a >> b) & 1u);
        m &= ((a >> b) & 1u) ? bb : ~bb;
      }
      __builtin_amdgcn_wave_barrier();
      uint32_t got = 0;
      if (valid) got = atomicAdd(&row[a], 1u);
      __builtin_amdgcn_wave_barrier();
      if (valid && got != before + (uint32_t)__popcll(m & lane_lt)) nbad++;
    } else if (MODE == 1) {
      acc += atomicAdd(&row[a], 1u);
    } else {
      volatile uint32_t *vr = row;
      uint64_t m = ~0ULL;
      for (uint32_t b = 0; b < bits; b++) {
        const uint64_t bb = __ballot((a >> b) & 1u);
        m &= ((a >> b) & 1u) ? bb : ~bb;
      }
      acc += vr[a] + __popcll(m & lane_lt);
      __builtin_amdgcn_wave_barrier();
      if ((m & lane_lt) == 0) atomicAdd(&row[a], (uint32_t)__popcll(m));
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (MODE == 0) {
    if (nbad) atomicAdd(bad, nbad);
  } else {
    sink[blockIdx.x * 512 + threadIdx.x] = acc;
  }
}

int main() {
  unsigned long long *bad;
  uint32_t *sink;
  hipMalloc(&bad, 8);
  hipMalloc(&sink, 4096 * 512 * 4);
  const uint32_t Ks[] = {1, 2, 3, 5, 8, 16, 64, 512, 1000};
  for (uint32_t K : Ks) {
    uint32_t bits = 0;
    while ((1u << bits) < K) bits++;
    hipMemset(bad, 0, 8);
    const int rounds = 4096, blocks = 2048;
    for (uint32_t seed = 1; seed <= 4; seed++)
      hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 8 * K * 4, 0, K, bits, rounds, seed * 7919u, bad, sink);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("K=%4u  lane-ops checked %.3g  out-of-order returns %llu\n", K, 4.0 * blocks * 512 * rounds * 7 / 8, h);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms[3] = {0, 0, 0};
    for (int mode = 1; mode <= 2; mode++) {
      hipEventRecord(e0);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 8 * K * 4, 0, K, bits, rounds, 3u, bad, sink);
      else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 8 * K * 4, 0, K, bits, rounds, 3u, bad, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
    }
    const double ops = (double)blocks * 512 * rounds;
    printf("        rank by atomic %.1f G lane-ops/s   rank by ballots %.1f G lane-ops/s\n", ops / ms[1] * 1e-6,
           ops / ms[2] * 1e-6);
  }
  return 0;
}
