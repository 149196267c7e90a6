// Does a device -> PAGEABLE host copy issued on a side stream, ordered behind the producing stream by an event, always
// deliver the producer's data after hipStreamSynchronize(side stream)?  (gg_result_fetch over the fetch lanes once
// returned garbage rowids to GG_KEY_JOIN, whose destination is a pageable vector.)  The producer kernel writes an
// iteration stamp into 8 KB + 8 KB on stream P; the copies go over 4 side streams round robin, into fresh pageable
// buffers, exactly as the library did; every word is checked.  Then the same with page-locked destinations.
// build: hipcc --offload-arch=gfx950 -O2 -o build/ubench_pageable_lanes scripts/ubench_pageable_lanes.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_stamp(int64_t *a, int64_t *b, int n, int64_t stamp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    a[i] = stamp * 4096 + i;
    b[i] = -(stamp * 4096 + i);
  }
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000, n = 1024;
  int64_t *da, *db;
  (void)hipMalloc(&da, n * 8);
  (void)hipMalloc(&db, n * 8);
  hipStream_t P, lane[4];
  hipEvent_t ready[4];
  (void)hipStreamCreateWithFlags(&P, hipStreamNonBlocking);
  for (int i = 0; i < 4; i++) {
    (void)hipStreamCreateWithFlags(&lane[i], hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&ready[i], hipEventDisableTiming);
  }
  for (int mode = 0; mode < 2; mode++) {
    int64_t *pa = nullptr, *pb = nullptr;
    if (mode == 1) {
      (void)hipHostMalloc((void **)&pa, n * 8, hipHostMallocPortable);
      (void)hipHostMalloc((void **)&pb, n * 8, hipHostMallocPortable);
    }
    long bad_iters = 0, bad_words = 0;
    for (int it = 1; it <= iters; it++) {
      std::vector<int64_t> va, vb;
      int64_t *ha = pa, *hb = pb;
      if (mode == 0) {
        va.resize(n);  // (fresh pageable memory every time, as a vector of the caller's)
        vb.resize(n);
        ha = va.data();
        hb = vb.data();
      }
      k_stamp<<<4, 256, 0, P>>>(da, db, n, it);
      const int l = it & 3;
      (void)hipEventRecord(ready[l], P);
      (void)hipStreamWaitEvent(lane[l], ready[l], 0);
      (void)hipMemcpyAsync(ha, da, n * 8, hipMemcpyDeviceToHost, lane[l]);
      (void)hipMemcpyAsync(hb, db, n * 8, hipMemcpyDeviceToHost, lane[l]);
      (void)hipStreamSynchronize(lane[l]);
      long w = 0;
      for (int i = 0; i < n; i++) w += (ha[i] != (int64_t)it * 4096 + i) + (hb[i] != -((int64_t)it * 4096 + i));
      if (w) {
        if (bad_iters < 5) printf("  iteration %d: %ld wrong words, first a=%lld (want %lld)\n", it, w, (long long)ha[0], (long long)it * 4096);
        bad_iters++;
        bad_words += w;
      }
    }
    printf("%s destinations, %d iterations over 4 side streams: %ld iterations with wrong data (%ld words)\n",
           mode == 0 ? "pageable" : "page-locked", iters, bad_iters, bad_words);
  }
  return 0;
}
