// micro-benchmark: issue rate of v_xad_u32 vs v_xor_b32+v_add_u32 vs v_add3_u32 vs v_lshl_add_u64 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a[8], t[8];
#pragma unroll
  for (int r = 0; r < 8; r++) { a[r] = threadIdx.x + r; t[r] = seed * (r + 1) + threadIdx.x; }
  uint32_t q = seed;
  for (int i = 0; i < iters; i++) {
    q = q * 1664525u + 1013904223u;  // scalar-ish
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (MODE == 0) asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(a[r]) : "v"(q), "v"(t[r]));
      if (MODE == 1) { uint32_t x; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(q), "v"(t[r])); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[r]) : "v"(x)); }
      if (MODE == 2) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[r]) : "v"(q), "v"(t[r]));
      if (MODE == 3) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[r]) : "v"(t[r]));
      if (MODE == 4) { uint32_t qs = __builtin_amdgcn_readfirstlane(q); asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(a[r]) : "s"(qs), "v"(t[r])); }
      if (MODE == 5) { uint32_t qs = __builtin_amdgcn_readfirstlane(q); uint32_t x; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(qs), "v"(t[r])); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[r]) : "v"(x)); }
      if (MODE == 6) { asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(*(uint64_t*)&a[r & 6]) : "v"(*(uint64_t*)&t[r & 6])); }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) s += a[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name) {
  uint32_t *d; hipMalloc(&d, 256 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  k<MODE><<<2048, 256>>>(d, 100, 1);
  hipEventRecord(e0); k<MODE><<<2048, 256>>>(d, iters, 7); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winstr = 2048.0 * 4 * iters * 8 * ((MODE == 1 || MODE == 5) ? 2 : 1);
  printf("%-10s %.3f ms  %.1f G wave-instr/s  (%.2f cycles/instr/SIMD at 2.4GHz x 1024 SIMDs)\n", name, ms, winstr / ms / 1e6,
         1024.0 * 2.4e9 / (winstr / (ms * 1e-3)));
  hipFree(d);
}
int main() { run<0>("xad"); run<1>("xor+add"); run<2>("add3"); run<3>("xor"); run<4>("xad(sgpr)"); run<5>("xor(s)+add"); run<6>("lshl_add_u64"); return 0; }
