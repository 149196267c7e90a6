// micro-benchmark: chip-wide rate of random (one line per lane) loads from a table that sits in an XCD's L2, in the
// Infinity Cache, or in HBM — the access pattern of the edge densification's dictionary probes (k_densify_pairs).
// Sweeps load width (4 / 16 bytes), cache policy bits of the buffer load (sc0, sc1, nt) and table size.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_gather scripts/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t h) {
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}

template <int WIDTH, int AUX, int K>
__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ tab, uint32_t slots_mask, int iters,
                                                uint32_t *__restrict__ out) {
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(tab), 0, (int)((slots_mask + 1u) * 16u), 0x00020000);
  uint32_t acc = 0, h = (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B9u + 12345u;
  for (int i = 0; i < iters; i++) {
    uint32_t off[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
      h = mix(h + 0x632BE5ABu);
      off[j] = (h & slots_mask) * 16u;  // 16-byte slots
    }
    if (WIDTH == 16) {
      u32x4 r[K];
#pragma unroll
      for (int j = 0; j < K; j++) r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], 0, AUX);
#pragma unroll
      for (int j = 0; j < K; j++) acc += r[j].x ^ r[j].w;
    } else {
      uint32_t r[K];
#pragma unroll
      for (int j = 0; j < K; j++) r[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, off[j], 0, AUX);
#pragma unroll
      for (int j = 0; j < K; j++) acc += r[j];
    }
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int WIDTH, int AUX, int K>
static void run(const char *policy, size_t table_bytes, int wgs_per_cu) {
  uint32_t *tab, *out;
  const int grid = 256 * wgs_per_cu;
  hipMalloc(&tab, table_bytes);
  hipMemset(tab, 1, table_bytes);
  hipMalloc(&out, (size_t)grid * 256 * 4);
  const uint32_t mask = (uint32_t)(table_bytes / 16) - 1u;
  const int iters = 256 / K;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_gather<WIDTH, AUX, K><<<grid, 256>>>(tab, mask, iters, out);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    k_gather<WIDTH, AUX, K><<<grid, 256>>>(tab, mask, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double loads = (double)grid * 256 * iters * K;
  printf("%2d B  %-8s K=%d  table %6.1f MB  %2d waves/CU  %8.3f ms  %7.1f G loads/s  (80 M probes: %6.1f us)\n", WIDTH,
         policy, K, table_bytes / 1048576.0, wgs_per_cu * 4, best, loads / best / 1e6, 80e6 / (loads / best / 1e3));
  hipFree(tab);
  hipFree(out);
}

int main() {
  const size_t sizes[] = {1u << 21, 1u << 22, 1u << 23, 1u << 26, 1u << 30};
  for (size_t sz : sizes) {
    run<16, 0, 4>("plain", sz, 8);
    run<16, 1, 4>("sc0", sz, 8);
    run<16, 2, 4>("nt", sz, 8);
    run<16, 16, 4>("sc1", sz, 8);
    run<16, 17, 4>("sc0sc1", sz, 8);
    run<4, 0, 4>("plain", sz, 8);
    run<4, 16, 4>("sc1", sz, 8);
    run<4, 17, 4>("sc0sc1", sz, 8);
    run<16, 0, 8>("plain", sz, 8);
    run<16, 0, 4>("plain", sz, 4);
    run<16, 0, 2>("plain", sz, 8);
  }
  return 0;
}
