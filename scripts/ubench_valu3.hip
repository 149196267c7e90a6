// micro-benchmark: issue rate on gfx950 of three-source integer instructions that fold two inputs into an
// accumulator in ONE instruction (candidates for a per-walk fold), next to v_xad_u32 and v_fma_f32.
// Same harness as scripts/ubench_valu.hip: 8 independent accumulator chains per lane, 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu3 scripts/ubench_valu3.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define OP3(name) asm volatile(name " %0, %1, %2, %0" : "+v"(a[r]) : "v"(q), "v"(t[r]))
#define OP2C(name) asm volatile(name " %0, %1, %2" : "+v"(a[r]) : "v"(q), "v"(t[r]))

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a[8], t[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    a[r] = threadIdx.x + r;
    t[r] = seed * (r + 1) + threadIdx.x;
  }
  const uint32_t q = seed * 2654435761u + threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        if (MODE == 0) OP3("v_xad_u32");
        if (MODE == 1) OP3("v_sad_u32");
        if (MODE == 2) OP3("v_sad_u16");
        if (MODE == 3) OP3("v_sad_u8");
        if (MODE == 4) OP3("v_mad_u32_u24");
        if (MODE == 5) OP3("v_mad_i32_i24");
        if (MODE == 6) OP2C("v_dot4c_i32_i8");
        if (MODE == 16) OP2C("v_dot2c_i32_i16");
        if (MODE == 18) OP2C("v_fmac_f32");
        if (MODE == 7) OP3("v_dot4_i32_i8");
        if (MODE == 8) OP3("v_lshl_add_u32");
        if (MODE == 9) OP3("v_bfi_b32");
        if (MODE == 10) OP3("v_or3_b32");
        if (MODE == 11) OP3("v_med3_u32");
        if (MODE == 12) OP3("v_perm_b32");
        if (MODE == 13) OP3("v_msad_u8");
        if (MODE == 14) OP3("v_alignbit_b32");
        if (MODE == 15) OP3("v_and_or_b32");
        if (MODE == 17) OP3("v_dot2_i32_i16");
      }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) s += a[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name) {
  uint32_t *d;
  (void)hipMalloc(&d, 256 * 2048 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 4000, grid = 256 * 8;
  k<MODE><<<grid, 256>>>(d, 10, 1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, iters, 7);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double winstr = (double)grid * 4 * iters * 32;
  printf("%-16s 8 waves/SIMD  %8.3f ms  %6.2f T lane-ops/s  %.2f cycles per wave-instr per SIMD (2.4 GHz)\n", name, best,
         winstr * 64 / best / 1e9, 1024.0 * 2.4e9 / (winstr / (best * 1e-3)));
  (void)hipFree(d);
}

int main() {
  run<0>("v_xad_u32");
  run<1>("v_sad_u32");
  run<2>("v_sad_u16");
  run<3>("v_sad_u8");
  run<4>("v_mad_u32_u24");
  run<5>("v_mad_i32_i24");
  run<6>("v_dot4c_i32_i8");
  run<16>("v_dot2c_i32_i16");
  run<18>("v_fmac_f32");
  run<7>("v_dot4_i32_i8");
  run<8>("v_lshl_add_u32");
  run<9>("v_bfi_b32");
  run<10>("v_or3_b32");
  run<11>("v_med3_u32");
  run<12>("v_perm_b32");
  run<13>("v_msad_u8");
  run<14>("v_alignbit_b32");
  run<15>("v_and_or_b32");
  run<17>("v_dot2_i32_i16");
  return 0;
}
