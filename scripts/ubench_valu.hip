// micro-benchmark: VALU issue rate on gfx950 of the product kernels' per-walk instruction (v_xad_u32: xor + add) next
// to reference instructions (v_add_u32, v_fma_f32, v_xor_b32, v_add3_u32, v_mul_lo_u32), at 1, 2, 4 and 8 waves
// per SIMD.  The counted loop holds ONLY the measured instruction (8 independent accumulator chains per lane, the
// second operand a loop-invariant register), plus the loop's scalar counter and branch, which issue on the scalar
// pipe.  bench.py's expansion roofline takes its peak from the "xad" row at 8 waves per SIMD of the output committed
// under profiles/ (lane-ops/s = wave-instr/s x 64).
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu scripts/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a[8], t[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    a[r] = threadIdx.x + r;
    t[r] = seed * (r + 1) + threadIdx.x;
  }
  const uint32_t q = seed * 2654435761u + threadIdx.x;
  float fa[8], ft[8], fq = (float)seed * 1.0001f;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    fa[r] = (float)a[r];
    ft[r] = 1.0f + (float)r * 1e-7f;
  }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        if (MODE == 0) asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(a[r]) : "v"(q), "v"(t[r]));
        if (MODE == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[r]) : "v"(t[r]));
        if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fa[r]) : "v"(fq), "v"(ft[r]));
        if (MODE == 3) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[r]) : "v"(t[r]));
        if (MODE == 4) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[r]) : "v"(q), "v"(t[r]));
        if (MODE == 5) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[r]) : "v"(t[r]));
        if (MODE == 7) {  // the state in a scalar register: two vector-register sources instead of three
          const uint32_t qs = __builtin_amdgcn_readfirstlane(q);
          asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(a[r]) : "s"(qs), "v"(t[r]));
        }
        if (MODE == 8) {  // xor with a scalar operand, then add
          const uint32_t qs = __builtin_amdgcn_readfirstlane(q);
          uint32_t x;
          asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(qs), "v"(t[r]));
          asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[r]) : "v"(x));
        }
        if (MODE == 9) {  // both xad inputs loop-invariant registers of other lanes' values (acc += (t ^ t2))
          asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(a[r]) : "v"(t[(r + 1) & 7]), "v"(t[r]));
        }
        if (MODE == 6) {
          uint32_t x;
          asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(q), "v"(t[r]));
          asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[r]) : "v"(x));
        }
      }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) s += a[r] + (uint32_t)fa[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int instr_per_slot) {
  uint32_t *d;
  hipMalloc(&d, 256 * 2048 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD: a 256-thread workgroup puts one wave on each SIMD of a CU
    const int grid = 256 * wps;
    k<MODE><<<grid, 256>>>(d, 10, 1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      k<MODE><<<grid, 256>>>(d, iters, 7);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double winstr = (double)grid * 4 * iters * 32 * instr_per_slot;
    const double per_simd_cycles = 1024.0 * 2.4e9 / (winstr / (best * 1e-3));
    printf("%-12s %d waves/SIMD  %8.3f ms  %8.1f G wave-instr/s  %6.2f T lane-ops/s  %.2f cycles per wave-instr per SIMD (2.4 GHz)\n",
           name, wps, best, winstr / best / 1e6, winstr * 64 / best / 1e9, per_simd_cycles);
  }
  hipFree(d);
}

int main() {
  run<0>("v_xad_u32", 1);
  run<1>("v_add_u32", 1);
  run<2>("v_fma_f32", 1);
  run<3>("v_xor_b32", 1);
  run<4>("v_add3_u32", 1);
  run<5>("v_mul_lo_u32", 1);
  run<6>("xor+add", 2);
  run<7>("v_xad(sgpr)", 1);
  run<8>("xor(s)+add", 2);
  run<9>("v_xad(v,v)", 1);
  return 0;
}
