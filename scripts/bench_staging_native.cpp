// The staging path alone, without an interpreter in the way: T host threads append SF100-sized key columns
// (39.8 M rows, 637 MB) in CHUNK-row calls — what the Sink threads of GG_EDGE_SINK do — then gg_staging_sync.
// GG_STAGING_TRACE=1 prints where the appenders wait.  Threads can be bound to one NUMA node (arg 3).
// build: g++ -O2 -std=c++17 -pthread -Iinclude -o build/bench_staging_native scripts/bench_staging_native.cpp \
//            -Lduckdb_pgq_amd -lgg -Wl,-rpath,'$ORIGIN/../duckdb_pgq_amd'
// usage: bench_staging_native [rows=39831322] [chunk=65536] [node=-1]
#include "gg.h"
#include <sched.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static std::vector<int> node_cpus(int node) {
  std::vector<int> cpus;
  char path[128];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return cpus;
  char buf[4096];
  if (fgets(buf, sizeof buf, f))
    for (char *tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
      int a, b;
      if (sscanf(tok, "%d-%d", &a, &b) == 2) {
        for (int c = a; c <= b; c++) cpus.push_back(c);
      } else if (sscanf(tok, "%d", &a) == 1) {
        cpus.push_back(a);
      }
    }
  fclose(f);
  return cpus;
}

int main(int argc, char **argv) {
  const uint64_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 39831322ull;
  const uint64_t chunk = argc > 2 ? strtoull(argv[2], nullptr, 10) : 65536;
  const int node = argc > 3 ? atoi(argv[3]) : -1;
  std::vector<int64_t> src(rows), dst(rows);
  for (uint64_t i = 0; i < rows; i++) {
    src[i] = (int64_t)(i * 2654435761u % 448626);
    dst[i] = (int64_t)(i * 40503u % 448626);
  }
  gg_ctx *ctx = nullptr;
  if (gg_ctx_create(0, &ctx) != GG_OK) {
    fprintf(stderr, "gg_ctx_create: %s\n", gg_last_error());
    return 1;
  }
  gg_ctx_set_edge_rowid(ctx, 0);
  std::vector<int> cpus = node >= 0 ? node_cpus(node) : std::vector<int>();
  printf("rows %llu, %llu-row appends, appender threads %s\n", (unsigned long long)rows, (unsigned long long)chunk,
         node >= 0 ? ("bound to node " + std::to_string(node)).c_str() : "unbound");
  const uint64_t n_chunks = (rows + chunk - 1) / chunk;
  for (int T : {1, 2, 4, 6, 8, 12, 16, 32, 64}) {
    double best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      gg_staging_clear(ctx);
      std::atomic<uint64_t> next{0};
      std::atomic<int> failed{0};
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&] {
          if (!cpus.empty()) {
            cpu_set_t set;
            CPU_ZERO(&set);
            for (int c : cpus) CPU_SET(c, &set);
            sched_setaffinity(0, sizeof set, &set);
          }
          for (;;) {
            const uint64_t i = next.fetch_add(1);
            if (i >= n_chunks) return;
            const uint64_t a = i * chunk, n = a + chunk <= rows ? chunk : rows - a;
            if (gg_edges_append(ctx, src.data() + a, dst.data() + a, nullptr, n) != GG_OK) failed = 1;
          }
        });
      for (auto &t : th) t.join();
      if (gg_staging_sync(ctx) != GG_OK || failed) {
        fprintf(stderr, "staging failed: %s\n", gg_last_error());
        return 1;
      }
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rep && dt < best) best = dt;
    }
    printf("%3d threads: %7.2f ms  %5.1f GB/s\n", T, best * 1e3, rows * 16 / best / 1e9);
    fflush(stdout);
  }
  gg_ctx_destroy(ctx);
  return 0;
}
