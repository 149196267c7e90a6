// Counter calibration: what do rocprofv3's WRITE_SIZE / FETCH_SIZE report for a KNOWN number of bytes, per store
// form?  Every kernel below moves exactly BYTES bytes per array (line-aligned, fully coalesced, every byte once), so
// counter value / BYTES is the factor to apply to that form.  The forms are the ones the library uses: plain and
// nontemporal stores of 4, 8 and 16 bytes per lane; three arrays written side by side with nontemporal 16-byte
// stores (k_mat_mid2's shape); streaming loads of 4 and 16 bytes per lane.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_write_cal scripts/ubench_write_cal.hip
// run (one counter per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- ./build/ubench_write_cal
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- ./build/ubench_write_cal
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef long long ll2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ inline T make(uint32_t a, uint32_t b);
template <> __device__ inline uint32_t make<uint32_t>(uint32_t a, uint32_t b) { return a ^ b; }
template <> __device__ inline long long make<long long>(uint32_t a, uint32_t b) { return ((long long)a << 32) | b; }
template <> __device__ inline ll2 make<ll2>(uint32_t a, uint32_t b) {
  ll2 v;
  v.x = a;
  v.y = b;
  return v;
}

// ARRAYS arrays of `n` elements of T each; a workgroup writes 4 KB x sizeof(T)/16 per step, grid-strided
template <typename T, bool NT, int ARRAYS>
__global__ __launch_bounds__(256) void k_store(T *__restrict__ a, T *__restrict__ b, T *__restrict__ c, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += stride) {
    const T v = make<T>((uint32_t)q, (uint32_t)(q >> 32));
    if (NT) {
      __builtin_nontemporal_store(v, a + q);
      if (ARRAYS > 1) __builtin_nontemporal_store(v, b + q);
      if (ARRAYS > 2) __builtin_nontemporal_store(v, c + q);
    } else {
      a[q] = v;
      if (ARRAYS > 1) b[q] = v;
      if (ARRAYS > 2) c[q] = v;
    }
  }
}

template <typename T> __device__ inline uint32_t fold(T v);
template <> __device__ inline uint32_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ inline uint32_t fold<ll2>(ll2 v) { return (uint32_t)v.x ^ (uint32_t)v.y; }

template <typename T, bool NT>
__global__ __launch_bounds__(256) void k_load(const T *__restrict__ a, uint64_t n, uint32_t *__restrict__ sink) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint32_t acc = 0;
  for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += stride)
    acc ^= fold<T>(NT ? __builtin_nontemporal_load(a + q) : a[q]);
  if (acc == 0x9E3779B9u) sink[0] = acc;  // keeps the loads
}

static const uint64_t BYTES = 8ull << 30;  // per array: past the 256 MiB Infinity Cache by 32x

template <typename T, bool NT, int ARRAYS> static void store(const char *name, void *a, void *b, void *c) {
  const uint64_t n = BYTES / sizeof(T);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  k_store<T, NT, ARRAYS><<<256 * 16, 256>>>((T *)a, (T *)b, (T *)c, n);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s writes %llu bytes  %7.3f ms  %5.2f TB/s\n", name, (unsigned long long)(BYTES * ARRAYS), ms, BYTES * ARRAYS / ms / 1e9);
}

template <typename T, bool NT> static void load(const char *name, void *a, uint32_t *sink) {
  const uint64_t n = BYTES / sizeof(T);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  k_load<T, NT><<<256 * 16, 256>>>((const T *)a, n, sink);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s reads  %llu bytes  %7.3f ms  %5.2f TB/s\n", name, (unsigned long long)BYTES, ms, BYTES / ms / 1e9);
}

int main() {
  void *a, *b, *c;
  uint32_t *sink;
  if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&b, BYTES) != hipSuccess || hipMalloc(&c, BYTES) != hipSuccess ||
      hipMalloc(&sink, 64) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  for (int rep = 0; rep < 2; rep++) {  // every kernel twice: the summary averages, the first touch shows if it differs
    store<uint32_t, false, 1>("store 4 B plain", a, b, c);
    store<uint32_t, true, 1>("store 4 B nt", a, b, c);
    store<long long, false, 1>("store 8 B plain", a, b, c);
    store<long long, true, 1>("store 8 B nt", a, b, c);
    store<ll2, false, 1>("store 16 B plain", a, b, c);
    store<ll2, true, 1>("store 16 B nt", a, b, c);
    store<ll2, false, 3>("store 16 B plain x3 arrays", a, b, c);
    store<ll2, true, 3>("store 16 B nt x3 arrays", a, b, c);
    load<uint32_t, false>("load 4 B plain", a, sink);
    load<ll2, false>("load 16 B plain", a, sink);
    load<ll2, true>("load 16 B nt", a, sink);
  }
  (void)hipDeviceSynchronize();
  (void)hipFree(a);
  (void)hipFree(b);
  (void)hipFree(c);
  (void)hipFree(sink);
  return 0;
}
