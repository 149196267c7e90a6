// micro-benchmark: random 16-byte gathers from a small table WHILE the same kernel streams a large array through
// the memory system — the edge densification's shape (two id columns in, dense pairs out, two dictionary probes per
// row).  Question: which cache policy of the streaming loads / stores leaves the table in the XCD's L2?
// Per row: one 16-byte stream load, two gathers, one 8-byte stream store.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_gather_stream scripts/ubench_gather_stream.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t h) {
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}

template <int LD_AUX, int ST_AUX, bool GATHER, int LOADS = 1>
__global__ __launch_bounds__(512) void k(const uint32_t *__restrict__ tab, uint32_t slots_mask, const u32x4 *__restrict__ in,
                                         u32x2 *__restrict__ out, uint64_t rows) {
  const __amdgpu_buffer_rsrc_t rt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(tab), 0, (int)((slots_mask + 1u) * 16u), 0x00020000);
  const uint64_t tile = (uint64_t)blockIdx.x * 8192;
  // (a tile of 8192 rows: 128 KB in, 64 KB out: 32-bit offsets inside a per-tile descriptor)
  const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(in + tile), 0, 8192 * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + tile, 0, 8192 * 8, 0x00020000);
  if (tile >= rows) return;
#pragma unroll 1
  for (int it = 0; it < 16; it += 2) {
    u32x4 x[2];
    u32x4 g[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++) x[j] = __builtin_amdgcn_raw_buffer_load_b128(ri, ((it + j) * 512 + threadIdx.x) * 16, 0, LD_AUX);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const uint32_t h0 = mix(x[j].x + (uint32_t)tile + threadIdx.x * 977u + it + j), h1 = mix(h0 ^ x[j].z ^ 0x5bd1e995u);
      if (GATHER) {
        g[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rt, (h0 & slots_mask) * 16u, 0, 0);
        g[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rt, (h1 & slots_mask) * 16u, 0, 0);
        if (LOADS >= 2) {  // a second 16-byte load from the SAME 128-byte line, issued back to back (pair ^ 1)
          const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rt, ((h0 & slots_mask) ^ 1u) * 16u, 0, 0);
          const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rt, ((h1 & slots_mask) ^ 1u) * 16u, 0, 0);
          g[j][0].w ^= a.y;
          g[j][1].z ^= b.x;
        }
        if (LOADS >= 3) {  // and, for a fifth of the probes, a third one from the same line after the first two came back
          if ((g[j][0].x ^ h0) % 5u == 0) {
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rt, ((h0 & slots_mask) ^ 2u) * 16u, 0, 0);
            g[j][0].w ^= c.x;
          }
          if ((g[j][1].y ^ h1) % 5u == 0) {
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rt, ((h1 & slots_mask) ^ 2u) * 16u, 0, 0);
            g[j][1].z ^= c.y;
          }
        }
      } else {
        g[j][0] = x[j];
        g[j][1] = x[j];
        g[j][0].x = h0;
        g[j][1].y = h1;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      u32x2 r;
      r.x = g[j][0].x ^ g[j][0].w;
      r.y = g[j][1].y ^ g[j][1].z;
      __builtin_amdgcn_raw_buffer_store_b64(r, ro, ((it + j) * 512 + threadIdx.x) * 8, 0, ST_AUX);
    }
  }
}

template <int LD_AUX, int ST_AUX, bool GATHER, int LOADS = 1>
static void run(const char *name, size_t table_bytes, uint32_t *tab, u32x4 *in, u32x2 *out, uint64_t rows) {
  const uint32_t mask = (uint32_t)(table_bytes / 16) - 1u;
  const unsigned grid = (unsigned)((rows + 8191) / 8192);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<LD_AUX, ST_AUX, GATHER, LOADS><<<grid, 512>>>(tab, mask, in, out, rows);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    k<LD_AUX, ST_AUX, GATHER, LOADS><<<grid, 512>>>(tab, mask, in, out, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-34s table %5.1f MB  %8.1f us  (%5.2f TB/s of streams%s)\n", name, table_bytes / 1048576.0, best * 1e3,
         rows * 24.0 / best / 1e9, GATHER ? "" : ", no gathers");
}

int main() {
  const uint64_t rows = 40000000ull / 8192 * 8192;  // ~SF100's edge rows: 0.64 GB in, 0.32 GB out
  uint32_t *tab;
  u32x4 *in;
  u32x2 *out;
  hipMalloc(&tab, 64u << 20);
  hipMemset(tab, 1, 64u << 20);
  hipMalloc(&in, rows * 16);
  hipMemset(in, 3, rows * 16);
  hipMalloc(&out, rows * 8);
  // aux bits (gfx940+): 1 = sc0, 2 = nt, 16 = sc1
  run<0, 0, false>("streams only: plain / plain", 2u << 20, tab, in, out, rows);
  run<2, 2, false>("streams only: nt / nt", 2u << 20, tab, in, out, rows);
  run<17, 17, false>("streams only: sc0sc1 / sc0sc1", 2u << 20, tab, in, out, rows);
  for (size_t sz : {2u << 20, 4u << 20, 8u << 20}) {
    run<0, 0, true>("ld plain   / st plain", sz, tab, in, out, rows);
    run<2, 2, true>("ld nt      / st nt", sz, tab, in, out, rows);
    run<16, 16, true>("ld sc1     / st sc1", sz, tab, in, out, rows);
    run<17, 17, true>("ld sc0sc1  / st sc0sc1", sz, tab, in, out, rows);
    run<3, 3, true>("ld sc0nt   / st sc0nt", sz, tab, in, out, rows);
    run<18, 18, true>("ld sc1nt   / st sc1nt", sz, tab, in, out, rows);
    run<19, 19, true>("ld sc0sc1nt/ st sc0sc1nt", sz, tab, in, out, rows);
    run<2, 2, true, 2>("ld nt / st nt, 2 loads per probe", sz, tab, in, out, rows);
    run<2, 2, true, 3>("ld nt / st nt, 2 + 0.2 loads", sz, tab, in, out, rows);
    run<2, 17, true>("ld nt      / st sc0sc1", sz, tab, in, out, rows);
    run<17, 2, true>("ld sc0sc1  / st nt", sz, tab, in, out, rows);
  }
  return 0;
}
