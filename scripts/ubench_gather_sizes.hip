// micro-benchmark: the edge densification's shape (two id columns streamed in, dense pairs streamed out, two random
// 16-byte dictionary probes per row) against the SIZE of the dictionary, in steps finer than powers of two:
// where between 4 MB (an XCD's L2) and 8 MB does the probe rate fall off?
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_gather_sizes scripts/ubench_gather_sizes.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t h) {
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(512) void k(const uint32_t *__restrict__ tab, uint32_t granules, const u32x4 *__restrict__ in,
                                         u32x2 *__restrict__ out, uint64_t rows) {
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(tab), 0, (int)(granules * 16u), 0x00020000);
  const uint64_t tile = (uint64_t)blockIdx.x * 8192;
  const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(in + tile), 0, 8192 * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + tile, 0, 8192 * 8, 0x00020000);
  if (tile >= rows) return;
#pragma unroll 1
  for (int it = 0; it < 16; it += 2) {
    u32x4 x[2], g[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++) x[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, ((it + j) * 512 + threadIdx.x) * 16, 0, 2);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const uint32_t h0 = mix(x[j].x + (uint32_t)tile + threadIdx.x * 977u + it + j), h1 = mix(h0 ^ x[j].z ^ 0x5bd1e995u);
      g[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rt, __umulhi(h0, granules) * 16u, 0, 0);
      g[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rt, __umulhi(h1, granules) * 16u, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      u32x2 r;
      r.x = g[j][0].x ^ g[j][0].w;
      r.y = g[j][1].y ^ g[j][1].z;
      __builtin_amdgcn_raw_buffer_store_b64(r, ro, ((it + j) * 512 + threadIdx.x) * 8, 0, 2);
    }
  }
}

int main() {
  const uint64_t rows = 40000000ull / 8192 * 8192;
  uint32_t *tab;
  u32x4 *in;
  u32x2 *out;
  (void)hipMalloc(&tab, 64u << 20);
  (void)hipMemset(tab, 1, 64u << 20);
  (void)hipMalloc(&in, rows * 16);
  (void)hipMemset(in, 3, rows * 16);
  (void)hipMalloc(&out, rows * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const double mbs[] = {2, 3, 3.5, 4, 4.57, 5, 5.33, 6, 6.4, 7, 8, 12, 16};
  for (int pass = 0; pass < 2; pass++)
    for (double mb : mbs) {
      const uint32_t granules = (uint32_t)(mb * 1048576.0 / 16.0);
      float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        k<<<(unsigned)(rows / 8192), 512>>>(tab, granules, in, out, rows);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("table %6.2f MB  %8.1f us  %6.1f G probes/s\n", mb, best * 1e3, rows * 2.0 / best / 1e6);
    }
  return 0;
}
