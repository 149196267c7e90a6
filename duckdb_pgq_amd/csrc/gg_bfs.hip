// gg_bfs.hip — 64-source bitset BFS (shortest path length) over the device CSR.
//
// Replaces the reference's level loop for `WITH RECURSIVE friends(...) ... UNION ...` +
// `min(hopCount) GROUP BY startPerson, friend` (benchmark/ldbc/queries/bi-10-shortestpath.sql:8-31):
//   PhysicalRecursiveCTE::{Sink,ProbeHT,GetData,ExecuteRecursivePipelines}
//       src/execution/operator/set/physical_recursive_cte.cpp:48-139  (per level: rebuild the knows hash
//       table, probe it with the working table, dedupe whole tuples in a GroupedAggregateHashTable)
//   GroupedAggregateHashTable::FindOrCreateGroups   src/execution/aggregate_hashtable.cpp:367-504
//   PhysicalHashAggregate (min)                     src/execution/operator/aggregate/physical_hash_aggregate.cpp:152-266
// Here each of up to 64 sources owns one bit lane of a uint64 per vertex; one level is
//   k_bfs_compact  frontier words -> active-vertex list (__ballot + popcount compaction), level stats
//   k_bfs_expand   wavefront per active vertex: coalesced CSR row read, 8-byte OR into next[w]
//                  (skipped when the neighbour has already seen every lane in the word)
//   k_bfs_update   new = next & ~seen; seen |= new; dist[lane][v] = level for each new bit
// The CSR is built once and reused by every level (the reference rebuilds its hash table per level,
// SURVEY.md F4).  Algorithmic bytes per level: 8V + 16Va + 24*TE_level (+ 24V update) — SURVEY §8d.
#include "gg_internal.h"

using namespace gg;

namespace gg {

struct BfsLevel {  // device counters for one level
  unsigned long long n_active;
  unsigned long long te;
  unsigned long long reached;
};

__global__ __launch_bounds__(64) void k_bfs_seed(const uint32_t *__restrict__ src_dense, int n_src, uint64_t V,
                                                 uint64_t *__restrict__ frontier, uint64_t *__restrict__ seen,
                                                 int32_t *__restrict__ dist, BfsLevel *__restrict__ lv) {
  const int i = threadIdx.x;
  if (i >= n_src) return;
  const uint32_t v = src_dense[i];
  if (v == INVALID_U32) return;
  atomicOr((unsigned long long *)&frontier[v], 1ULL << i);
  atomicOr((unsigned long long *)&seen[v], 1ULL << i);
  dist[(uint64_t)i * V + v] = 0;
  atomicAdd(&lv->reached, 1ULL);
}

__global__ __launch_bounds__(256) void k_bfs_compact(const uint64_t *__restrict__ frontier, uint64_t V,
                                                     const uint32_t *__restrict__ off, uint32_t *__restrict__ active,
                                                     BfsLevel *__restrict__ lv) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = v < V && frontier[v] != 0;
  const uint64_t m = __ballot(on);
  if (m == 0) return;
  const int lane = threadIdx.x & 63;
  uint64_t deg = on ? (uint64_t)(off[v + 1] - off[v]) : 0;
  deg = wave_reduce_add_u64(deg);
  uint64_t base = 0;
  if (lane == 0) {
    base = atomicAdd(&lv->n_active, (unsigned long long)__popcll(m));
    atomicAdd(&lv->te, (unsigned long long)deg);
  }
  base = __shfl(base, 0, 64);
  if (on) active[base + __popcll(m & ((1ULL << lane) - 1ULL))] = (uint32_t)v;
}

// a wavefront per active vertex (grid-strided)
__global__ __launch_bounds__(256) void k_bfs_expand(const uint32_t *__restrict__ active, uint64_t n_active,
                                                    const uint64_t *__restrict__ frontier,
                                                    const uint64_t *__restrict__ seen, const uint32_t *__restrict__ off,
                                                    const uint32_t *__restrict__ nbr, uint64_t *__restrict__ next) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t a = wave0; a < n_active; a += nwaves) {
    const uint32_t v = active[a];
    const uint64_t f = frontier[v];
    const uint32_t b = off[v], e = off[v + 1];
    for (uint32_t i = b + lane; i < e; i += 64) {
      const uint32_t w = nbr[i];
      const uint64_t nf = f & ~seen[w];
      if (nf) atomicOr((unsigned long long *)&next[w], (unsigned long long)nf);
    }
  }
}

__global__ __launch_bounds__(256) void k_bfs_update(uint64_t *__restrict__ frontier, uint64_t *__restrict__ seen,
                                                    uint64_t *__restrict__ next, uint64_t V, int level,
                                                    int32_t *__restrict__ dist, BfsLevel *__restrict__ lv) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t nw = 0;
  if (v < V) {
    const uint64_t s = seen[v];
    nw = next[v] & ~s;
    next[v] = 0;
    frontier[v] = nw;
    if (nw) seen[v] = s | nw;
  }
  uint64_t cnt = (uint64_t)__popcll(nw);
  uint64_t bits = nw;
  while (bits) {
    const int b = __ffsll((long long)bits) - 1;
    bits &= bits - 1;
    dist[(uint64_t)b * V + v] = level;
  }
  cnt = wave_reduce_add_u64(cnt);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&lv->reached, (unsigned long long)cnt);
}

__global__ __launch_bounds__(256) void k_bfs_gather(const int32_t *__restrict__ dist, uint64_t V, int n_src,
                                                    const uint32_t *__restrict__ dst_dense, uint64_t n_dst,
                                                    int32_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)n_src * n_dst) return;
  const uint64_t s = i / n_dst, j = i % n_dst;
  const uint32_t d = dst_dense[j];
  out[i] = d == INVALID_U32 ? -1 : dist[s * V + d];
}

}  // namespace gg

extern "C" int gg_bfs64(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, int n_src, int max_hops,
                        const int64_t *dst_ids, uint64_t n_dst, int32_t *out_dist, gg_bfs_stats *stats) {
  if (!ctx || !csr || csr->ctx != ctx || !out_dist || n_src < 0 || n_src > GG_BFS_LANES || (n_src && !src_ids) ||
      (n_dst && !dst_ids)) {
    set_error("gg_bfs64: bad argument (n_src must be 0..%d)", GG_BFS_LANES);
    return GG_ERR_INVALID_ARG;
  }
  if (csr->n_parts > 1) {
    set_error("gg_bfs64 needs a whole CSR, not a shard");
    return GG_ERR_STATE;
  }
  GG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint64_t V = csr->V;
  gg_bfs_stats st;
  memset(&st, 0, sizeof(st));
  const uint64_t n_out = dst_ids ? n_dst : V;
  if (n_src == 0 || n_out == 0) {
    if (stats) *stats = st;
    return GG_OK;
  }
  if (V == 0) {
    for (uint64_t i = 0; i < (uint64_t)n_src * n_out; i++) out_dist[i] = -1;
    if (stats) *stats = st;
    return GG_OK;
  }

  int64_t *ids_dev = nullptr;
  uint32_t *src_dense = nullptr, *active = nullptr;
  uint64_t *frontier = nullptr, *seen = nullptr, *next = nullptr;
  int32_t *dist = nullptr;
  BfsLevel *lv = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&ids_dev, GG_BFS_LANES * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&src_dense, GG_BFS_LANES * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&active, V * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&frontier, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&seen, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&next, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&dist, (uint64_t)n_src * V * sizeof(int32_t)));
  GG_TRY(ctx->dev_alloc((void **)&lv, sizeof(BfsLevel)));

  GG_HIP(hipMemcpyAsync(ids_dev, src_ids, (size_t)n_src * sizeof(int64_t), hipMemcpyHostToDevice, s));
  GG_HIP(hipStreamSynchronize(s));
  GG_TRY(lookup_ids(ctx, csr, ids_dev, (uint64_t)n_src, src_dense));
  GG_HIP(hipMemsetAsync(frontier, 0, V * sizeof(uint64_t), s));
  GG_HIP(hipMemsetAsync(seen, 0, V * sizeof(uint64_t), s));
  GG_HIP(hipMemsetAsync(next, 0, V * sizeof(uint64_t), s));
  GG_HIP(hipMemsetAsync(dist, 0xFF, (uint64_t)n_src * V * sizeof(int32_t), s));
  GG_HIP(hipMemsetAsync(lv, 0, sizeof(BfsLevel), s));
  GG_LAUNCH(ctx, "bfs_seed", k_bfs_seed, dim3(1), dim3(64), 0, src_dense, n_src, V, frontier, seen, dist, lv);

  const unsigned vgrid = (unsigned)((V + 255) / 256);
  int level = 0;
  uint64_t reached = 0;
  while (max_hops < 0 || level < max_hops) {
    // active list of this level + its stats; the host needs n_active to size the expand launch
    GG_LAUNCH(ctx, "bfs_compact", k_bfs_compact, dim3(vgrid), dim3(256), 0, frontier, V, csr->off, active, lv);
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, lv, sizeof(BfsLevel), hipMemcpyDeviceToHost, s));
    GG_HIP(hipStreamSynchronize(s));
    BfsLevel h;
    memcpy(&h, ctx->pin_scratch, sizeof(h));
    reached = h.reached;
    if (h.n_active == 0) break;
    st.levels++;
    st.active_vertices += h.n_active;
    st.traversed_edges += h.te;
    // reset per-level counters (reached keeps accumulating)
    GG_HIP(hipMemsetAsync(lv, 0, 2 * sizeof(unsigned long long), s));
    uint64_t waves = h.n_active;
    uint64_t max_waves = (uint64_t)ctx->num_cus * 32;  // one resident set; grid-stride the rest
    if (waves > max_waves) waves = max_waves;
    GG_LAUNCH(ctx, "bfs_expand", k_bfs_expand, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, active,
              (uint64_t)h.n_active, frontier, seen, csr->off, csr->nbr, next);
    level++;
    GG_LAUNCH(ctx, "bfs_update", k_bfs_update, dim3(vgrid), dim3(256), 0, frontier, seen, next, V, level, dist, lv);
  }
  // final reached count (the loop may have exited on the hop bound right after an update)
  GG_HIP(hipMemcpyAsync(ctx->pin_scratch, lv, sizeof(BfsLevel), hipMemcpyDeviceToHost, s));
  GG_HIP(hipStreamSynchronize(s));
  {
    BfsLevel h;
    memcpy(&h, ctx->pin_scratch, sizeof(h));
    reached = h.reached;
  }
  st.reached_pairs = reached;

  int rc = GG_OK;
  if (!dst_ids) {
    hipError_t e = hipMemcpyAsync(out_dist, dist, (uint64_t)n_src * V * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      set_error("gg_bfs64: result copy failed: %s", hipGetErrorString(e));
      rc = GG_ERR_HIP;
    }
  } else {
    int64_t *dst_dev = nullptr;
    uint32_t *dst_dense = nullptr;
    int32_t *out_dev = nullptr;
    rc = ctx->dev_alloc((void **)&dst_dev, n_dst * sizeof(int64_t));
    if (rc == GG_OK) rc = ctx->dev_alloc((void **)&dst_dense, n_dst * sizeof(uint32_t));
    if (rc == GG_OK) rc = ctx->dev_alloc((void **)&out_dev, (uint64_t)n_src * n_dst * sizeof(int32_t));
    if (rc == GG_OK) {
      hipError_t e = hipMemcpyAsync(dst_dev, dst_ids, n_dst * sizeof(int64_t), hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) rc = GG_ERR_HIP;
    }
    if (rc == GG_OK) rc = lookup_ids(ctx, csr, dst_dev, n_dst, dst_dense);
    if (rc == GG_OK) {
      const uint64_t tot = (uint64_t)n_src * n_dst;
      hipLaunchKernelGGL(k_bfs_gather, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, dist, V, n_src, dst_dense,
                         n_dst, out_dev);
      hipError_t e = hipMemcpyAsync(out_dist, out_dev, tot * sizeof(int32_t), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        set_error("gg_bfs64: gather failed: %s", hipGetErrorString(e));
        rc = GG_ERR_HIP;
      }
    }
    ctx->dev_free(dst_dev);
    ctx->dev_free(dst_dense);
    ctx->dev_free(out_dev);
  }
  ctx->dev_free(ids_dev);
  ctx->dev_free(src_dense);
  ctx->dev_free(active);
  ctx->dev_free(frontier);
  ctx->dev_free(seen);
  ctx->dev_free(next);
  ctx->dev_free(dist);
  ctx->dev_free(lv);
  if (stats) *stats = st;
  return rc;
}
