// What does the PCIe link of this box give a host -> device copy of the SF100 key columns (637 MB)?  The staging path
// (gg_edges_append + flush_edge_block) measured 43 GB/s; this program separates the link from the library:
//   (a) hipMemcpyAsync from page-locked memory, by piece size and by number of streams,
//   (b) the same with the page-locked memory first touched by a thread bound to each NUMA node in turn
//       (the copy engine reads host memory through the socket the GPU hangs off; the other socket is one hop further),
//   (c) a kernel reading the page-locked memory itself (no copy engine): 16-byte loads, nontemporal stores to HBM.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_h2d_numa scripts/ubench_h2d_numa.hip
#include <hip/hip_runtime.h>
#include <sched.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef long long ll2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_pull(const ll2 *__restrict__ host, ll2 *__restrict__ dev, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(host + q), dev + q);
}

static std::vector<int> node_cpus(int node) {
  std::vector<int> cpus;
  char path[128];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return cpus;
  char buf[4096];
  if (fgets(buf, sizeof buf, f)) {
    for (char *tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
      int a, b;
      if (sscanf(tok, "%d-%d", &a, &b) == 2) {
        for (int c = a; c <= b; c++) cpus.push_back(c);
      } else if (sscanf(tok, "%d", &a) == 1) {
        cpus.push_back(a);
      }
    }
  }
  fclose(f);
  return cpus;
}

static bool bind_to(const std::vector<int> &cpus) {
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : cpus) CPU_SET(c, &set);
  return sched_setaffinity(0, sizeof set, &set) == 0;
}

static const uint64_t BYTES = 640ull << 20;

static double copy_ms(char *dev, const char *host, uint64_t piece, int n_streams, hipStream_t *streams) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, streams[0]);
  for (int s = 1; s < n_streams; s++) (void)hipStreamWaitEvent(streams[s], e0, 0);
  int s = 0;
  for (uint64_t off = 0; off < BYTES; off += piece, s = (s + 1) % n_streams)
    (void)hipMemcpyAsync(dev + off, host + off, piece < BYTES - off ? piece : BYTES - off, hipMemcpyHostToDevice, streams[s]);
  for (int t = 1; t < n_streams; t++) {
    hipEvent_t e;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    (void)hipEventRecord(e, streams[t]);
    (void)hipStreamWaitEvent(streams[0], e, 0);
  }
  (void)hipEventRecord(e1, streams[0]);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  char *dev;
  if (hipMalloc(&dev, BYTES) != hipSuccess) return 1;
  hipStream_t streams[4];
  for (auto &s : streams) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int gpu_node = -1;
  {
    char bus[64] = {0};
    (void)hipDeviceGetPCIBusId(bus, sizeof bus, 0);
    for (char *p = bus; *p; p++) *p = (char)tolower(*p);
    std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    if (FILE *f = fopen(path.c_str(), "r")) {
      if (fscanf(f, "%d", &gpu_node) != 1) gpu_node = -1;
      fclose(f);
    }
    printf("GPU 0 at %s, numa_node %d\n", bus, gpu_node);
  }
  for (int node = -1; node < 8; node++) {
    std::vector<int> cpus;
    if (node >= 0) {
      cpus = node_cpus(node);
      if (cpus.empty()) continue;
      if (!bind_to(cpus)) {
        printf("node %d: sched_setaffinity refused\n", node);
        continue;
      }
    }
    char *host;
    if (hipHostMalloc((void **)&host, BYTES, hipHostMallocPortable) != hipSuccess) {
      printf("hipHostMalloc failed\n");
      return 1;
    }
    memset(host, 1, BYTES);
    printf("page-locked memory touched by a thread %s (%zu cpus)\n", node < 0 ? "left where it was" : ("bound to node " + std::to_string(node)).c_str(), cpus.size());
    for (uint64_t piece : {(uint64_t)1 << 20, (uint64_t)4 << 20, (uint64_t)16 << 20, (uint64_t)64 << 20, BYTES})
      for (int ns : {1, 2, 4}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
          double ms = copy_ms(dev, host, piece, ns, streams);
          if (ms < best) best = ms;
        }
        printf("  memcpy pieces of %4llu MiB on %d stream(s): %7.3f ms  %5.1f GB/s\n", (unsigned long long)(piece >> 20), ns, best, BYTES / best / 1e6);
      }
    for (int grid : {64, 256, 1024, 4096}) {
      double best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, streams[0]);
        k_pull<<<grid, 256, 0, streams[0]>>>((const ll2 *)host, (ll2 *)dev, BYTES / 16);
        (void)hipEventRecord(e1, streams[0]);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("  kernel pull, %4d workgroups: %7.3f ms  %5.1f GB/s\n", grid, best, BYTES / best / 1e6);
    }
    (void)hipHostFree(host);
  }
  (void)hipFree(dev);
  return 0;
}
