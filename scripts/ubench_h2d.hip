// micro-benchmark: host-to-device copies from page-locked memory — the rate by copy size, from one stream and from
// two (the edge staging sends two 8 MB copies per block on one stream).
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_h2d scripts/ubench_h2d.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

int main() {
  const size_t total = 640ull << 20;
  char *h, *d;
  (void)hipHostMalloc((void **)&h, total, hipHostMallocDefault);
  (void)hipMalloc((void **)&d, total);
  for (size_t i = 0; i < total; i += 4096) h[i] = (char)i;
  hipStream_t s[2];
  (void)hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
  for (int streams = 1; streams <= 2; streams++)
    for (size_t chunk : {1ull << 20, 8ull << 20, 32ull << 20, 128ull << 20, 640ull << 20}) {
      double best = 1e9;
      for (int rep = 0; rep < 4; rep++) {
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        int k = 0;
        for (size_t o = 0; o < total; o += chunk, k++)
          (void)hipMemcpyAsync(d + o, h + o, chunk < total - o ? chunk : total - o, hipMemcpyHostToDevice, s[k % streams]);
        (void)hipStreamSynchronize(s[0]);
        (void)hipStreamSynchronize(s[1]);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep && dt < best) best = dt;
      }
      printf("%d stream(s), copies of %4zu MB: %6.2f ms  %5.1f GB/s\n", streams, chunk >> 20, best * 1e3, total / best / 1e9);
    }
  return 0;
}
