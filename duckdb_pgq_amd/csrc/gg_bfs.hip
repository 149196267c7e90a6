// gg_bfs.hip — 64-source bitset BFS (shortest path length) over the device CSR.
//
// Replaces the reference's level loop for `WITH RECURSIVE friends(...) ... UNION ...` +
// `min(hopCount) GROUP BY startPerson, friend` (benchmark/ldbc/queries/bi-10-shortestpath.sql:8-31):
//   PhysicalRecursiveCTE::{Sink,ProbeHT,GetData,ExecuteRecursivePipelines}
//       src/execution/operator/set/physical_recursive_cte.cpp:48-139  (per level: rebuild the knows hash
//       table, probe it with the working table, dedupe whole tuples in a GroupedAggregateHashTable)
//   GroupedAggregateHashTable::FindOrCreateGroups   src/execution/aggregate_hashtable.cpp:367-504
//   PhysicalHashAggregate (min)                     src/execution/operator/aggregate/physical_hash_aggregate.cpp:152-266
// Here each of up to 64 sources owns one bit lane of a uint64 per vertex.  A level runs in one of two
// directions, chosen from the frontier's edge count (both give identical bits):
//   push (light frontier)  k_bfs_push:    a wavefront takes 64 frontier words, ballots the non-empty ones and walks
//                                         their rows (coalesced CSR row read), 8-byte OR into next[w] only for
//                                         lanes w has not seen; `next` is a third word buffer, all-zero between levels
//                          k_bfs_update:  new = next & ~seen; seen |= new; distances; next frontier + stats; next = 0
//   pull (heavy frontier)  k_bfs_pull:    16 lanes per vertex w that
//                                         still misses lanes: OR of frontier[v] over the in-neighbours
//                                         (reverse CSR row read + 8-byte gathers from the V-sized, L2-resident frontier),
//                                         no atomics; writes the next frontier, seen, distances and the
//                                         next level's stats in the same pass
// Distances live as one byte per (vertex, lane) — 64 contiguous bytes per vertex — and are widened to the
// API's int32 [lane][vertex] layout only when fetched.
// The CSR is built once and reused by every level (the reference rebuilds its hash table per level,
// SURVEY.md F4).  Algorithmic bytes per level: 8V + 16Va + 24*TE_level (+ 24V update) — SURVEY §8d.
#include "gg_internal.h"

#ifndef GG_PULL_LANES
#define GG_PULL_LANES 8   // lanes per vertex in the pull kernels (SF100 pull levels per batch: 32 lanes 443 us, 16: 348, 8: 296, 4: 299, 2: 388)
#endif
#ifndef GG_PULL_CHECK
#define GG_PULL_CHECK 1   // trips between two looks whether the vertex is complete (1: 296 us, 2: 306, 4: 329)
#endif
#ifndef GG_BFS_PULL_FACTOR
#define GG_BFS_PULL_FACTOR 8  // a level pulls when its frontier has more than E / this many edges (16: 0.65 ms per batch, 8: 0.62, 4: 0.66 at SF100)
#endif

using namespace gg;

namespace gg {

struct BfsLevel {  // device counters describing the NEXT frontier
  unsigned long long n_active;
  unsigned long long te;
  unsigned long long reached;
};

// Distance cells are DistT (uint8_t, or uint16_t for BFS deeper than 254 levels), 64 per vertex,
// contiguous.  A vertex's newly reached lanes are written by composing whole 8-byte words.
template <typename DistT>
__device__ __forceinline__ void write_dist(uint64_t *__restrict__ dist /* [V][64] DistT */, uint64_t v, uint64_t nw,
                                           uint32_t level) {
  constexpr int PER_WORD = 8 / (int)sizeof(DistT);     // lanes per 8-byte word
  constexpr int WORDS = 64 / PER_WORD;
  constexpr uint64_t CELL = sizeof(DistT) == 1 ? 0xFFULL : 0xFFFFULL;
  uint64_t *d = dist + v * WORDS;
#pragma unroll
  for (int k = 0; k < WORDS; k++) {
    const uint32_t b = (uint32_t)(nw >> (PER_WORD * k)) & ((1u << PER_WORD) - 1u);
    if (b) {
      uint64_t m = 0, val = 0;
#pragma unroll
      for (int i = 0; i < PER_WORD; i++) {
        m |= ((b >> i) & 1u) ? (CELL << (8 * sizeof(DistT) * i)) : 0ULL;
        val |= (uint64_t)level << (8 * sizeof(DistT) * i);
      }
      d[k] = (d[k] & ~m) | (val & m);  // cells are written at most once (first level that reaches them)
    }
  }
}

// block-aggregate (n_active, te, reached) and add to the level counters with three atomics per block
__device__ __forceinline__ void add_level_stats(uint64_t act, uint64_t te, uint64_t reached,
                                                uint64_t *s_red /* 3 per wave of the block */,
                                                BfsLevel *__restrict__ lv) {
  act = wave_reduce_add_u64(act);
  te = wave_reduce_add_u64(te);
  reached = wave_reduce_add_u64(reached);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  if (lane == 0) {
    s_red[wave * 3] = act;
    s_red[wave * 3 + 1] = te;
    s_red[wave * 3 + 2] = reached;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    uint64_t s = 0;
    for (int w = 0; w < waves; w++) s += s_red[w * 3 + threadIdx.x];
    if (s) atomicAdd(threadIdx.x == 0 ? &lv->n_active : (threadIdx.x == 1 ? &lv->te : &lv->reached),
                     (unsigned long long)s);
  }
}

template <typename DistT>
__global__ __launch_bounds__(64) void k_bfs_seed(const uint32_t *__restrict__ src_dense, int n_src,
                                                 const uint32_t *__restrict__ off, uint64_t *__restrict__ frontier,
                                                 uint64_t *__restrict__ seen, uint64_t *__restrict__ dist8,
                                                 BfsLevel *__restrict__ lv, const int64_t *__restrict__ vid = nullptr,
                                                 uint32_t part = 0, uint32_t n_parts = 1) {
  // one lane per source; a vertex seeded by several sources is counted once (by its lowest lane)
  const int i = threadIdx.x;
  const uint32_t v = i < n_src ? src_dense[i] : INVALID_U32;
  const bool valid = v != INVALID_U32;
  bool first = valid;
  for (int j = 0; j < 64; j++) {
    const uint32_t other = __shfl(v, j, 64);
    if (j < i && other == v) first = false;
  }
  // graph-sharded runs: every rank holds the whole frontier, but a vertex's seen word, distances and
  // statistics live on the rank that owns it
  const bool mine = valid && (n_parts <= 1 || owns(vid[v], part, n_parts));
  if (valid) atomicOr((unsigned long long *)&frontier[v], 1ULL << i);
  if (mine) {
    atomicOr((unsigned long long *)&seen[v], 1ULL << i);
    reinterpret_cast<DistT *>(dist8)[(uint64_t)v * 64 + i] = 0;
  }
  const uint64_t act = (uint64_t)__popcll(__ballot(first && mine));
  const uint64_t te = wave_reduce_add_u64(first && mine ? (uint64_t)(off[v + 1] - off[v]) : 0ULL);
  const uint64_t reached = (uint64_t)__popcll(__ballot(mine));
  if (i == 0) {
    lv->n_active = act;
    lv->te = te;
    lv->reached = reached;
  }
}

// ---- levels driven from the device (bfs_run) --------------------------------------------------------------------
// The host does not read anything back between levels: every level's kernels are launched in advance and each
// decides from the statistics of the frontier it starts from (steps[level - 1], written by the level before)
// whether there is anything to do and in which direction: done (empty frontier), pull (heavy) or push (light).
// Kernels of the direction not taken return at once.  steps[] also records which of the two word buffers holds the
// frontier a level produced (pull levels ping-pong between them; push levels go through a third, always-clean one).
struct BfsStep {
  unsigned long long n_active, te, reached;  // BfsLevel of the frontier this level produced
  unsigned long long cur;                    // buffer (0/1) that holds it
  unsigned long long reserved;
};
enum BfsMode : int { BFS_DONE = 0, BFS_PUSH = 1, BFS_PULL = 2 };
__device__ __forceinline__ int bfs_mode(const BfsStep &prev, uint64_t E) {
  if (prev.n_active == 0) return BFS_DONE;
  return prev.te * GG_BFS_PULL_FACTOR > E ? BFS_PULL : BFS_PUSH;  // heavy frontier: gather instead of scatter
}

// push level, device-driven: no active-vertex list.  A wave reads 64 consecutive frontier words (one coalesced
// 512-byte load), ballots the non-empty ones and walks their rows one after the other; light frontiers put about one
// active vertex in a wave's 64, so the parallelism is the list's without the compaction launch before it.  Pushes
// go into `fnext`, a buffer of its own that is all-zero between levels (k_bfs_update_dev re-zeroes what it folds),
// so nothing has to be cleaned after a pull level either.
__global__ __launch_bounds__(256) void k_bfs_push_dev(const BfsStep *__restrict__ steps, uint32_t level, uint64_t E,
                                                      const uint64_t *__restrict__ f0, const uint64_t *__restrict__ f1,
                                                      uint64_t *__restrict__ fnext, const uint64_t *__restrict__ seen,
                                                      const uint32_t *__restrict__ off,
                                                      const uint32_t *__restrict__ nbr, uint64_t V) {
  const BfsStep prev = steps[level - 1];
  if (bfs_mode(prev, E) != BFS_PUSH) return;
  const uint64_t *__restrict__ frontier = prev.cur ? f1 : f0;
  const int lane = threadIdx.x & 63;
  const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t base = wave0 * 64; base < V; base += nwaves * 64) {
    const uint64_t mine = base + lane < V ? frontier[base + lane] : 0ULL;
    uint64_t m = __ballot(mine != 0);
    while (m) {  // uniform
      const int l = __ffsll((long long)m) - 1;
      m &= m - 1;
      const uint64_t f = __shfl(mine, l, 64);
      const uint32_t b = off[base + l], e = off[base + l + 1];
      // four 64-entry steps of the row at a time: their neighbour loads, then their seen gathers, are in flight
      // together (a row is walked by ONE wavefront; hubs of the light levels have hundreds of entries)
      for (uint32_t i = b + lane; i < e; i += 256) {
        uint32_t w[4];
        uint64_t sw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = i + 64 * k < e ? nbr[i + 64 * k] : INVALID_U32;
#pragma unroll
        for (int k = 0; k < 4; k++) sw[k] = w[k] != INVALID_U32 ? seen[w[k]] : ~0ULL;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint64_t nf = f & ~sw[k];
          if (nf) atomicOr((unsigned long long *)&fnext[w[k]], (unsigned long long)nf);
        }
      }
    }
  }
}

// after a push: fold `next` into the new frontier
template <typename DistT>
__device__ __forceinline__ void update_body(uint64_t *__restrict__ frontier, uint64_t *__restrict__ seen,
                                            uint64_t *__restrict__ next, uint64_t V, uint32_t level,
                                            const uint32_t *__restrict__ off, uint64_t *__restrict__ dist8,
                                            BfsLevel *__restrict__ lv, uint64_t *s_red, uint32_t blocks);

template <typename DistT>
__global__ __launch_bounds__(256) void k_bfs_update(uint64_t *__restrict__ frontier, uint64_t *__restrict__ seen,
                                                    uint64_t *__restrict__ next, uint64_t V, uint32_t level,
                                                    const uint32_t *__restrict__ off, uint64_t *__restrict__ dist8,
                                                    BfsLevel *__restrict__ lv) {
  __shared__ uint64_t s_red[12];
  update_body<DistT>(frontier, seen, next, V, level, off, dist8, lv, s_red, 0);
}

template <typename DistT>
__device__ __forceinline__ void update_body(uint64_t *__restrict__ frontier, uint64_t *__restrict__ seen,
                                            uint64_t *__restrict__ next, uint64_t V, uint32_t level,
                                            const uint32_t *__restrict__ off, uint64_t *__restrict__ dist8,
                                            BfsLevel *__restrict__ lv, uint64_t *s_red, uint32_t blocks) {
  // grid-strided over `blocks` blocks (default: the grid): the level counters take three same-address atomics per
  // BLOCK, so few blocks
  if (!blocks) blocks = gridDim.x;
  uint64_t act = 0, te = 0, reached = 0;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (uint64_t)blocks * blockDim.x) {
    const uint64_t s = seen[v];
    const uint64_t nw = next[v] & ~s;
    next[v] = 0;
    frontier[v] = nw;
    if (nw) {
      seen[v] = s | nw;
      write_dist<DistT>(dist8, v, nw, level);
      te += off[v + 1] - off[v];
      act += 1;
      reached += (uint64_t)__popcll(nw);
    }
  }
  add_level_stats(act, te, reached, s_red, lv);
}

// pull: four vertices per wavefront, sixteen lanes each (grid-strided); reads fin, writes fout (ping-pong
// frontiers).  In-degrees are heavy-tailed (SF100: median 27, mean 89, max ~1000): a whole wavefront per
// vertex leaves 40 % of the lanes idle and runs one dependent seen -> offsets -> list -> frontier chain at a
// time, so every list goes to a 16-lane group, four chains in flight per wave.

// OR over a lane group of L lanes, the result in every lane of the group.  Groups of 8 use data-parallel-primitive
// moves (two swaps inside the quads, then the mirror image of the 8 lanes, which lies in the other quad) instead
// of three dependent ds_bpermute round trips per 32-bit half; the reduction sits inside the pull loop's
// load -> test -> next load chain.  Call with the whole wave active.
#ifndef GG_PULL_DPP
#define GG_PULL_DPP 1
#endif
template <int L>
__device__ __forceinline__ uint64_t group_or(uint64_t g) {
#if GG_PULL_DPP
  if (L == 8) {
    uint32_t lo = (uint32_t)g, hi = (uint32_t)(g >> 32);
    lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false);
    hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, false);
    lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, false);
    hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xF, 0xF, false);
    lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x141 /* row_half_mirror */, 0xF, 0xF, false);
    hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x141, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
  }
#endif
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) g |= __shfl_xor(g, o, 64);  // stays inside the lane group
  return g;
}

template <typename DistT>
__device__ __forceinline__ void pull_body(const uint64_t *__restrict__ fin, uint64_t *__restrict__ fout,
                                          uint64_t *__restrict__ seen, uint64_t V, uint32_t level,
                                          const uint32_t *__restrict__ off, const uint32_t *__restrict__ roff,
                                          const uint32_t *__restrict__ rnbr, uint64_t *__restrict__ dist8,
                                          BfsLevel *__restrict__ lv, uint64_t *s_red);

template <typename DistT>
__global__ __launch_bounds__(256) void k_bfs_pull(const uint64_t *__restrict__ fin, uint64_t *__restrict__ fout,
                                                  uint64_t *__restrict__ seen, uint64_t V, uint32_t level,
                                                  const uint32_t *__restrict__ off, const uint32_t *__restrict__ roff,
                                                  const uint32_t *__restrict__ rnbr, uint64_t *__restrict__ dist8,
                                                  BfsLevel *__restrict__ lv) {
  __shared__ uint64_t s_red[12];
  pull_body<DistT>(fin, fout, seen, V, level, off, roff, rnbr, dist8, lv, s_red);
}

// Second (and last) kernel of a device-driven level: after a push it folds `fnext` into the new frontier (on its
// first `update_blocks` blocks), on a heavy frontier it IS the level (pull).  One launch for either direction: a
// kernel that finds nothing to do still costs 4-5 us of stream time, and a five-level search had eleven of those.
template <typename DistT>
__global__ __launch_bounds__(256) void k_bfs_finish_dev(BfsStep *__restrict__ steps, uint32_t level, uint64_t E,
                                                        uint64_t *__restrict__ f0, uint64_t *__restrict__ f1,
                                                        uint64_t *__restrict__ fnext, uint64_t *__restrict__ seen,
                                                        uint64_t V, const uint32_t *__restrict__ off,
                                                        const uint32_t *__restrict__ roff,
                                                        const uint32_t *__restrict__ rnbr,
                                                        uint64_t *__restrict__ dist8, uint32_t update_blocks) {
  __shared__ uint64_t s_red[12];
  const BfsStep prev = steps[level - 1];
  const int mode = bfs_mode(prev, E);
  if (mode == BFS_DONE) return;
  BfsLevel *lv = reinterpret_cast<BfsLevel *>(&steps[level]);
  if (mode == BFS_PUSH) {
    if (blockIdx.x >= update_blocks) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) steps[level].cur = prev.cur;  // the new frontier stays in the same buffer
    update_body<DistT>(prev.cur ? f1 : f0, seen, fnext, V, level, off, dist8, lv, s_red, update_blocks);
  } else {
    if (blockIdx.x == 0 && threadIdx.x == 0) steps[level].cur = prev.cur ^ 1ULL;  // every word of the other buffer is written
    pull_body<DistT>(prev.cur ? f1 : f0, prev.cur ? f0 : f1, seen, V, level, off, roff, rnbr, dist8, lv, s_red);
  }
}

// the word buffers, seen words and distance cells of a batch in one launch (six memsets were six commands)
__global__ __launch_bounds__(256) void k_bfs_clear(uint64_t *__restrict__ f0, uint64_t *__restrict__ f1,
                                                   uint64_t *__restrict__ fnext, uint64_t *__restrict__ seen,
                                                   uint64_t *__restrict__ dist_words, uint64_t V, uint64_t n_dist_words,
                                                   uint64_t *__restrict__ steps_words, uint64_t n_steps_words) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t i = t; i < V; i += stride) {
    f0[i] = 0;
    f1[i] = 0;
    fnext[i] = 0;
    seen[i] = 0;
  }
  for (uint64_t i = t; i < n_dist_words; i += stride) dist_words[i] = ~0ULL;
  for (uint64_t i = t; i < n_steps_words; i += stride) steps_words[i] = 0;
}

template <typename DistT>
__device__ __forceinline__ void pull_body(const uint64_t *__restrict__ fin, uint64_t *__restrict__ fout,
                                          uint64_t *__restrict__ seen, uint64_t V, uint32_t level,
                                          const uint32_t *__restrict__ off, const uint32_t *__restrict__ roff,
                                          const uint32_t *__restrict__ rnbr, uint64_t *__restrict__ dist8,
                                          BfsLevel *__restrict__ lv, uint64_t *s_red) {
  constexpr int L = GG_PULL_LANES, VPW = 64 / L;  // lanes per vertex, vertices per wavefront
  const int lane = threadIdx.x & 63, group = lane / L, gl = lane % L;
  const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  uint64_t act = 0, te = 0, reached = 0;  // accumulated by the first lane of each group
  for (uint64_t wbase = wave0 * VPW; wbase < V; wbase += nwaves * VPW) {
    const uint64_t w = wbase + group;
    const bool valid = w < V;
    const uint64_t s = valid ? seen[w] : ~0ULL;
    uint32_t b = 0, e = 0;
    if (valid && ~s) {  // some lane still missing at w
      b = roff[w];
      e = roff[w + 1];
    }
    uint64_t acc = 0;
    {
      // a lane group walks its list L entries per trip and stops as soon as the lanes w still misses are all
      // found (checked every GG_PULL_CHECK trips): in the levels where the frontier is most of the graph a vertex is
      // complete after one or two trips instead of the 5.5 an average list takes.  (Long lists used to be handed to
      // the whole wavefront, one after the other: with the early stop that only made the other groups wait —
      // 462 us of pull levels per batch at SF100 against 389 without.)
      bool walking = b < e;
      uint32_t i0 = b;
      for (int trip = 1; __any(walking); trip++) {
        if (walking) {
          const uint32_t i = i0 + gl;
          if (i < e) acc |= fin[rnbr[i]];
          i0 += L;
          if (i0 >= e) walking = false;
        }
        if ((trip % GG_PULL_CHECK) == 0) {
          if ((group_or<L>(acc) | s) == ~0ULL) walking = false;
        }
      }
    }
    acc = group_or<L>(acc);
    const uint64_t nw = acc & ~s;
    if (gl == 0 && valid) {
      fout[w] = nw;
      if (nw) {
        seen[w] = s | nw;
        write_dist<DistT>(dist8, w, nw, level);
        act += 1;
        te += off[w + 1] - off[w];
        reached += (uint64_t)__popcll(nw);
      }
    }
  }
  add_level_stats(act, te, reached, s_red, lv);
}

// dist [V][64] cells -> out int32 [n_src][n_out] (targets: all vertices, or a dense target list)
template <typename DistT>
__global__ __launch_bounds__(256) void k_bfs_widen(const DistT *__restrict__ dist8, uint64_t V, int n_src,
                                                   const uint32_t *__restrict__ dst_dense /* nullable */,
                                                   uint64_t n_out, int32_t *__restrict__ out) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_out) return;
  const uint32_t d = dst_dense ? dst_dense[j] : (uint32_t)j;
  for (int s = 0; s < n_src; s++) {
    int32_t val = -1;
    if (d != INVALID_U32) {
      const DistT b = dist8[(uint64_t)d * 64 + s];
      val = b == (DistT)~(DistT)0 ? -1 : (int32_t)b;
    }
    out[(uint64_t)s * n_out + j] = val;  // coalesced across j
  }
}

// reached (lane, vertex) cells per vertex: count, then (after an exclusive scan) fill
// (source id, vertex id, distance) rows.  One thread per vertex reads its 64 cells (one 64/128-byte line).
template <typename DistT>
__global__ __launch_bounds__(256) void k_bfs_pairs_count(const DistT *__restrict__ dist, uint64_t V, int n_src,
                                                         uint32_t *__restrict__ counts) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  uint32_t c = 0;
  for (int s = 0; s < n_src; s++) c += dist[v * 64 + s] != (DistT)~(DistT)0;
  counts[v] = c;
}

template <typename DistT>
__global__ __launch_bounds__(256) void k_bfs_pairs_fill(const DistT *__restrict__ dist, uint64_t V, int n_src,
                                                        const uint32_t *__restrict__ offsets,
                                                        const int64_t *__restrict__ src_ids,
                                                        const int64_t *__restrict__ vid, int64_t *__restrict__ out_src,
                                                        int64_t *__restrict__ out_vtx, int64_t *__restrict__ out_dist) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  uint64_t pos = offsets[v];
  const int64_t id = vid[v];
  for (int s = 0; s < n_src; s++) {
    const DistT d = dist[v * 64 + s];
    if (d != (DistT)~(DistT)0) {
      out_src[pos] = src_ids[s];
      out_vtx[pos] = id;
      out_dist[pos] = (int64_t)d;
      pos++;
    }
  }
}

// the same rows as ONE 8-byte word each: lane << 58 | distance << 32 | dense vertex index.  A third of the
// bytes over PCIe; the host turns lane and dense index back into ids from two small tables it already has.
template <typename DistT>
__global__ __launch_bounds__(256) void k_bfs_pairs_fill_packed(const DistT *__restrict__ dist, uint64_t V, int n_src,
                                                               const uint32_t *__restrict__ offsets,
                                                               int64_t *__restrict__ out) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  uint64_t pos = offsets[v];
  for (int s = 0; s < n_src; s++) {
    const DistT d = dist[v * 64 + s];
    if (d != (DistT)~(DistT)0) out[pos++] = (int64_t)(((uint64_t)s << 58) | ((uint64_t)d << 32) | v);
  }
}

// ---- distinct walk endpoints (the UNION of the 1..k-hop endpoint sets of a source list) -----------------------
// Set image under the edge relation, level by level: bits[h] = out-neighbours of bits[h - 1].  Unlike the BFS
// there is no seen mask: a vertex that ends walks of several lengths carries several bits.
__global__ __launch_bounds__(256) void k_set_seed(const uint32_t *__restrict__ dense, uint64_t n,
                                                  uint32_t *__restrict__ bits) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && dense[i] != INVALID_U32) atomicOr(&bits[dense[i] >> 5], 1u << (dense[i] & 31u));
}

// sixteen lanes per member vertex walk its row
__global__ __launch_bounds__(256) void k_set_expand(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                    const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                    uint64_t V) {
  const uint64_t v = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const uint32_t sub = threadIdx.x & 15u;
  if (v >= V || !((in[v >> 5] >> (v & 31u)) & 1u)) return;
  const uint32_t a = off[v], b = off[v + 1];
  for (uint32_t e = a + sub; e < b; e += 16) {
    const uint32_t w = nbr[e];
    const uint32_t bit = 1u << (w & 31u);
    if (!(out[w >> 5] & bit)) atomicOr(&out[w >> 5], bit);
  }
}

// bits is (k_max + 1) consecutive arrays of `words` u32: mask of vertex v = bit h for every level h >= 1 that holds it
__device__ __forceinline__ uint32_t endpoint_mask(const uint32_t *__restrict__ bits, uint64_t words, int k_max,
                                                  uint64_t v) {
  uint32_t m = 0;
  for (int h = 1; h <= k_max; h++) m |= ((bits[(uint64_t)h * words + (v >> 5)] >> (v & 31u)) & 1u) << h;
  return m;
}

__global__ __launch_bounds__(256) void k_endpoint_count(const uint32_t *__restrict__ bits, uint64_t words, int k_max,
                                                        uint64_t V, uint32_t *__restrict__ counts) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) counts[v] = endpoint_mask(bits, words, k_max, v) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_endpoint_fill(const uint32_t *__restrict__ bits, uint64_t words, int k_max,
                                                       uint64_t V, const uint32_t *__restrict__ offsets,
                                                       const int64_t *__restrict__ vid, int64_t *__restrict__ out_id,
                                                       int64_t *__restrict__ out_mask) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const uint32_t m = endpoint_mask(bits, words, k_max, v);
  if (m) {
    out_id[offsets[v]] = vid[v];
    out_mask[offsets[v]] = (int64_t)m;
  }
}

}  // namespace gg

// One BFS with DistT distance cells.  *overflow is set (and nothing is returned) if the frontier is
// still non-empty at the deepest level DistT can record.
template <typename DistT>
static int bfs_run(gg_ctx *ctx, gg_csr *csr, const int64_t *src_ids, int n_src, int max_hops, const int64_t *dst_ids,
                   uint64_t n_dst, int32_t *out_dist, gg_bfs_stats *stats, bool *overflow,
                   gg_result *pairs = nullptr /* filled with (source, vertex, distance) rows if non-null */) {
  constexpr int MAX_LEVEL = sizeof(DistT) == 1 ? 254 : 65534;
  *overflow = false;
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
    for (uint64_t i = 0; out_dist && i < (uint64_t)n_src * n_out; i++) out_dist[i] = -1;
    if (stats) *stats = st;
    return GG_OK;
  }

  int64_t *ids_dev = nullptr;
  uint32_t *src_dense = nullptr;
  uint64_t *fa = nullptr, *fb = nullptr, *fnext = nullptr, *seen = nullptr, *dist8 = nullptr;
  BfsStep *steps = nullptr;            // steps[L]: the frontier level L produced (L = 0: the seed)
  constexpr size_t STEP_SLOTS = (size_t)MAX_LEVEL + 2;
  GG_TRY(ctx->dev_alloc((void **)&ids_dev, GG_BFS_LANES * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&src_dense, GG_BFS_LANES * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&fa, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&fb, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&fnext, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&seen, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&dist8, V * 64 * sizeof(DistT)));
  GG_TRY(ctx->dev_alloc((void **)&steps, STEP_SLOTS * sizeof(BfsStep)));

  // (pageable source: the runtime stages it before the call returns; no synchronisation needed here)
  GG_HIP(hipMemcpyAsync(ids_dev, src_ids, (size_t)n_src * sizeof(int64_t), hipMemcpyHostToDevice, s));
  GG_TRY(lookup_ids(ctx, csr, ids_dev, (uint64_t)n_src, src_dense));
  {
    const uint64_t dist_words = V * 64 * sizeof(DistT) / sizeof(uint64_t);
    const uint64_t steps_words = STEP_SLOTS * sizeof(BfsStep) / sizeof(uint64_t);
    const uint64_t most = dist_words > steps_words ? dist_words : steps_words;
    uint64_t blocks = (most + 256 * 8 - 1) / (256 * 8);  // eight words per thread
    if (blocks > (uint64_t)ctx->num_cus * 16) blocks = (uint64_t)ctx->num_cus * 16;
    GG_LAUNCH(ctx, "bfs_clear", k_bfs_clear, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, fa, fb, fnext, seen, dist8, V,
              dist_words, reinterpret_cast<uint64_t *>(steps), steps_words);
  }
  GG_LAUNCH(ctx, "bfs_seed", (k_bfs_seed<DistT>), dim3(1), dim3(64), 0, src_dense, n_src, csr->off, fa, seen, dist8,
            reinterpret_cast<BfsLevel *>(steps));  // steps[0]: cur = 0 (fa) from the memset

  const unsigned vgrid = (unsigned)((V + 255) / 256);
  const uint64_t max_waves = (uint64_t)ctx->num_cus * 32;  // one resident set; grid-stride the rest
  // Levels are launched ahead, BFS_CHUNK at a time, and decide on the device what they do (BfsStep); the host
  // reads the steps back once per chunk — once per batch for the usual five or six levels — instead of once
  // per level (0.83 ms wall for 0.60 ms of kernels at SF100 when every level waited for a D2H copy).
#ifndef GG_BFS_CHUNK
#define GG_BFS_CHUNK 6
#endif
#ifndef GG_BFS_UGRID
#define GG_BFS_UGRID 2  // blocks per CU of the update kernel
#endif
  constexpr int BFS_CHUNK = GG_BFS_CHUNK;
  const int level_cap = max_hops >= 0 && max_hops < MAX_LEVEL ? max_hops : MAX_LEVEL;
  const unsigned ugrid = vgrid < (unsigned)ctx->num_cus * GG_BFS_UGRID ? vgrid : (unsigned)ctx->num_cus * GG_BFS_UGRID;
  const uint64_t chunks = (V + 63) / 64;  // push: 64 frontier words per wavefront
  const unsigned push_grid = (unsigned)(((chunks < max_waves ? chunks : max_waves) * 64 + 255) / 256);
  const uint64_t quads = (V + 64 / GG_PULL_LANES - 1) / (64 / GG_PULL_LANES);  // pull: 64 / GG_PULL_LANES vertices per wavefront
  const unsigned pull_grid = (unsigned)(((quads < max_waves ? quads : max_waves) * 64 + 255) / 256);
  std::vector<BfsStep> host_steps((size_t)level_cap + 2);
  int launched = 0;  // levels 1..launched are enqueued
  uint64_t reached = 0;
  bool pull_ready = csr->roff != nullptr;
  while (true) {
    const int upto = launched + BFS_CHUNK < level_cap ? launched + BFS_CHUNK : level_cap;
    for (int level = launched + 1; level <= upto; level++) {
      if (!pull_ready) {  // (whole builds make the reverse CSR eagerly; legacy builds on first use)
        GG_TRY(ensure_reverse(ctx, csr));
        pull_ready = true;
      }
      GG_LAUNCH(ctx, "bfs_push", k_bfs_push_dev, dim3(push_grid), dim3(256), 0, (const BfsStep *)steps, (uint32_t)level,
                csr->E, (const uint64_t *)fa, (const uint64_t *)fb, fnext, (const uint64_t *)seen, csr->off, csr->nbr, V);
      GG_LAUNCH(ctx, "bfs_finish", (k_bfs_finish_dev<DistT>), dim3(pull_grid > ugrid ? pull_grid : ugrid), dim3(256), 0, steps,
                (uint32_t)level, csr->E, fa, fb, fnext, seen, V, csr->off, csr->roff, csr->rnbr, dist8, (uint32_t)ugrid);
    }
    launched = upto;
    GG_HIP(hipMemcpyAsync(host_steps.data(), steps, (size_t)(launched + 1) * sizeof(BfsStep), hipMemcpyDeviceToHost, s));
    GG_HIP(hipStreamSynchronize(s));
    // level L ran iff the frontier it started from (steps[L - 1]) was not empty
    int ran = 0;
    while (ran < launched && host_steps[ran].n_active != 0) ran++;
    const bool frontier_left = ran == launched && host_steps[launched].n_active != 0;
    if (!frontier_left || launched == level_cap) {
      st.levels = (uint32_t)ran;
      reached = 0;
      for (int l = 0; l <= ran; l++) reached += host_steps[l].reached;
      for (int l = 0; l < ran; l++) {
        st.active_vertices += host_steps[l].n_active;
        st.traversed_edges += host_steps[l].te;
      }
      // the deepest recordable level is reached with work left and the caller did not bound the search there
      if (frontier_left && launched == MAX_LEVEL && (max_hops < 0 || max_hops > MAX_LEVEL)) *overflow = true;
      break;
    }
  }
  st.reached_pairs = reached;

  int rc = GG_OK;
  if (pairs && !*overflow && reached) {
    if (reached >= (uint64_t)INVALID_U32) {
      set_error("gg_bfs64_pairs: more than 2^32-2 reached pairs in one batch");
      rc = GG_ERR_TOO_LARGE;
    }
    const bool packed = pairs->k_max == 0;  // one packed word per row (table 0) instead of three ids (table 2)
    const int table = packed ? 0 : 2, ncols = packed ? 1 : 3;
    uint32_t *counts = nullptr;
    if (rc == GG_OK) rc = ctx->dev_alloc((void **)&counts, V * sizeof(uint32_t));
    for (int c = 0; c < ncols && rc == GG_OK; c++)
      rc = ctx->dev_alloc((void **)&pairs->cols[table][c], reached * sizeof(int64_t));
    if (rc == GG_OK) {
      hipLaunchKernelGGL((k_bfs_pairs_count<DistT>), dim3(vgrid), dim3(256), 0, s, (const DistT *)dist8, V, n_src, counts);
      rc = scan_exclusive_u32(ctx, counts, counts, V, nullptr);
    }
    if (rc == GG_OK) {
      if (packed)
        hipLaunchKernelGGL((k_bfs_pairs_fill_packed<DistT>), dim3(vgrid), dim3(256), 0, s, (const DistT *)dist8, V, n_src,
                           (const uint32_t *)counts, pairs->cols[0][0]);
      else
        hipLaunchKernelGGL((k_bfs_pairs_fill<DistT>), dim3(vgrid), dim3(256), 0, s, (const DistT *)dist8, V, n_src,
                           (const uint32_t *)counts, (const int64_t *)ids_dev, (const int64_t *)csr->vid,
                           pairs->cols[2][0], pairs->cols[2][1], pairs->cols[2][2]);
      if (hipGetLastError() != hipSuccess) rc = GG_ERR_HIP;
    }
    if (rc == GG_OK) rc = scan_error_fetch(ctx);
    if (rc == GG_OK && hipStreamSynchronize(s) != hipSuccess) rc = GG_ERR_HIP;
    if (rc == GG_OK) rc = scan_error_test(ctx);
    if (rc == GG_OK) {
      for (int c = 0; c < ncols; c++) ctx->keep(pairs->cols[table][c]);
      pairs->rows[table] = reached;
    }
    ctx->dev_free(counts);
  }
  if (out_dist && !*overflow) {
    int64_t *dst_dev = nullptr;
    uint32_t *dst_dense = nullptr;
    int32_t *out_dev = nullptr;
    rc = ctx->dev_alloc((void **)&out_dev, (uint64_t)n_src * n_out * sizeof(int32_t));
    if (rc == GG_OK && dst_ids) {
      rc = ctx->dev_alloc((void **)&dst_dev, n_dst * sizeof(int64_t));
      if (rc == GG_OK) rc = ctx->dev_alloc((void **)&dst_dense, n_dst * sizeof(uint32_t));
      if (rc == GG_OK) {
        hipError_t e = hipMemcpyAsync(dst_dev, dst_ids, n_dst * sizeof(int64_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = GG_ERR_HIP;
      }
      if (rc == GG_OK) rc = lookup_ids(ctx, csr, dst_dev, n_dst, dst_dense);
    }
    if (rc == GG_OK) {
      hipLaunchKernelGGL((k_bfs_widen<DistT>), dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, (const DistT *)dist8, V,
                         n_src, (const uint32_t *)dst_dense, n_out, out_dev);
      hipError_t e =
          hipMemcpyAsync(out_dist, out_dev, (uint64_t)n_src * n_out * sizeof(int32_t), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        set_error("gg_bfs64: result copy failed: %s", hipGetErrorString(e));
        rc = GG_ERR_HIP;
      }
    }
    ctx->dev_free(dst_dev);
    ctx->dev_free(dst_dense);
    ctx->dev_free(out_dev);
  }
  ctx->dev_free(ids_dev);
  ctx->dev_free(src_dense);
  ctx->dev_free(fnext);
  ctx->dev_free(fa);
  ctx->dev_free(fb);
  ctx->dev_free(seen);
  ctx->dev_free(dist8);
  ctx->dev_free(steps);
  if (stats) *stats = st;
  return rc;
}

static int bfs_dispatch(gg_ctx *ctx, const gg_csr *csr_c, const int64_t *src_ids, int n_src, int max_hops,
                        const int64_t *dst_ids, uint64_t n_dst, int32_t *out_dist, gg_bfs_stats *stats,
                        gg_result *pairs);

extern "C" int gg_bfs64(gg_ctx *ctx, const gg_csr *csr_c, const int64_t *src_ids, int n_src, int max_hops,
                        const int64_t *dst_ids, uint64_t n_dst, int32_t *out_dist, gg_bfs_stats *stats) {
  return bfs_dispatch(ctx, csr_c, src_ids, n_src, max_hops, dst_ids, n_dst, out_dist, stats, nullptr);
}

static int bfs_pairs(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, int n_src, int max_hops,
                     gg_bfs_stats *stats, gg_result **out_result, int table);

extern "C" int gg_bfs64_pairs(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, int n_src, int max_hops,
                              gg_bfs_stats *stats, gg_result **out_result) {
  return bfs_pairs(ctx, csr, src_ids, n_src, max_hops, stats, out_result, 2);
}

extern "C" int gg_bfs64_pairs_packed(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, int n_src, int max_hops,
                                     gg_bfs_stats *stats, gg_result **out_result) {
  return bfs_pairs(ctx, csr, src_ids, n_src, max_hops, stats, out_result, 0);
}

static int bfs_pairs(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, int n_src, int max_hops,
                     gg_bfs_stats *stats, gg_result **out_result, int table) {
  if (!out_result) return GG_ERR_INVALID_ARG;
  *out_result = nullptr;
  if (!ctx) return GG_ERR_INVALID_ARG;
  gg_result *res = new gg_result();
  res->ctx = ctx;
  res->k_min = res->k_max = table;  // table 2: three id columns; table 0: one packed column
  int rc = bfs_dispatch(ctx, csr, src_ids, n_src, max_hops, nullptr, 0, nullptr, stats, res);
  if (rc != GG_OK) {
    gg_result_destroy(res);
    return rc;
  }
  *out_result = res;
  return GG_OK;
}

static int bfs_dispatch(gg_ctx *ctx, const gg_csr *csr_c, const int64_t *src_ids, int n_src, int max_hops,
                        const int64_t *dst_ids, uint64_t n_dst, int32_t *out_dist, gg_bfs_stats *stats,
                        gg_result *pairs) {
  gg_csr *csr = const_cast<gg_csr *>(csr_c);
  ApiScope scope(ctx);
  if (!ctx || !csr || csr->ctx != ctx || n_src < 0 || n_src > GG_BFS_LANES || (n_src && !src_ids) ||
      (n_dst && !dst_ids)) {
    set_error("gg_bfs64: bad argument (n_src must be 0..%d)", GG_BFS_LANES);
    return GG_ERR_INVALID_ARG;
  }
  if (csr->n_parts > 1) {
    set_error("gg_bfs64 needs a whole CSR, not a shard");
    return GG_ERR_STATE;
  }
  // one byte per (vertex, lane) covers 254 levels; deeper searches rerun with two-byte cells
  bool overflow = false;
  if (max_hops < 0 || max_hops > 254) {
    int rc = bfs_run<uint8_t>(ctx, csr, src_ids, n_src, max_hops, dst_ids, n_dst, out_dist, stats, &overflow, pairs);
    if (rc != GG_OK || !overflow) return rc;
    rc = bfs_run<uint16_t>(ctx, csr, src_ids, n_src, max_hops, dst_ids, n_dst, out_dist, stats, &overflow, pairs);
    if (rc == GG_OK && overflow) {
      set_error("gg_bfs64: search deeper than 65534 levels is not supported");
      return GG_ERR_TOO_LARGE;
    }
    return rc;
  }
  return bfs_run<uint8_t>(ctx, csr, src_ids, n_src, max_hops, dst_ids, n_dst, out_dist, stats, &overflow, pairs);
}

// ------------------------------------------------------------------------------------------------------
// Graph-sharded BFS (north_star's layout): the graph is vertex-partitioned over the
// GPUs (gg_csr_build_shard: a rank holds the reverse-CSR rows of the vertices it owns), every rank holds
// the whole 8*V-byte frontier, and a level is
//     expand   each rank pulls the next frontier words of ITS vertices over its reverse rows (k_bfs_pull on
//              the shard: rows of vertices it does not own are empty, so their words come out zero)
//     exchange the ranks combine their word arrays — supports are disjoint, so a SUM all-reduce over
//              RCCL is the bitwise OR (RCCL has no OR reduction) — outside this library (the host owns
//              the communicator); 3.6 MB per level at SF100
//     commit   the combined words become the frontier of the next level
// seen words, distances and the reached (source, vertex, distance) rows stay with the owner: the result is
// sharded like the graph, its union over the ranks is gg_bfs64_pairs of the whole graph.
// ------------------------------------------------------------------------------------------------------
struct gg_bfs_run {
  gg_ctx *ctx = nullptr;
  gg_csr *csr = nullptr;
  int n_src = 0;
  uint32_t level = 0;
  int64_t *ids_dev = nullptr;
  uint64_t *front = nullptr, *next = nullptr, *seen = nullptr, *dist8 = nullptr;
  uint64_t *acc = nullptr;  // push levels OR into this buffer (all-zero between levels)
  gg::BfsLevel *lv = nullptr;  // [0] the level's result, [1] statistics of the frontier it starts from
  uint64_t levels_push = 0, levels_pull = 0;
};

namespace gg {

// ---- graph-sharded levels: direction chosen per level and per rank -----------------------------------------------
// Every rank holds the whole frontier.  A heavy frontier is pulled over the reverse rows of the owned vertices
// (k_bfs_pull).  A light one is PUSHED along the same edges grouped by source (pin_off / pin_nbr: row u = the owned
// destinations of u's edges): only owned words are touched, so seen words, distances and statistics stay with the
// owner and the ranks' word arrays keep disjoint supports — the one all-reduce per level stays a SUM, whichever
// direction each rank took (a rank decides from ITS edges: te = entries of the frontier's rows in its pin CSR).
__global__ __launch_bounds__(256) void k_pin_count(const uint32_t *__restrict__ rnbr, uint64_t n,
                                                   uint32_t *__restrict__ cnt) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) atomicAdd(&cnt[rnbr[e]], 1u);
}
__global__ __launch_bounds__(256) void k_pin_fill(const uint32_t *__restrict__ rnbr, const uint32_t *__restrict__ rrow,
                                                  uint64_t n, uint32_t *__restrict__ cursor,
                                                  uint32_t *__restrict__ pin_nbr) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) pin_nbr[atomicAdd(&cursor[rnbr[e]], 1u)] = rrow[e];  // (order inside a row does not matter to a BFS)
}
// cursor[v] = pin_off[v] after the fill ran: shift back by the row lengths is not needed, the scan is kept apart
__global__ __launch_bounds__(256) void k_copy_u32(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

static int ensure_push_in(gg_ctx *ctx, gg_csr *csr) {
  if (csr->pin_off) return GG_OK;
  const uint64_t V = csr->V, n = csr->E_rev;
  hipStream_t s = ctx->stream;
  uint32_t *cnt = nullptr, *pin_off = nullptr, *pin_nbr = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&pin_off, (V + 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&pin_nbr, (n ? n : 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&cnt, (V + 1) * sizeof(uint32_t)));
  GG_HIP(hipMemsetAsync(cnt, 0, (V + 1) * sizeof(uint32_t), s));
  if (n)
    GG_LAUNCH(ctx, "pin_count", k_pin_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, csr->rnbr, n, cnt);
  GG_TRY(scan_exclusive_u32(ctx, cnt, pin_off, V + 1, nullptr));
  GG_LAUNCH(ctx, "pin_copy", k_copy_u32, dim3((unsigned)((V + 1 + 255) / 256)), dim3(256), 0, pin_off, cnt, V + 1);
  if (n)
    GG_LAUNCH(ctx, "pin_fill", k_pin_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, csr->rnbr, csr->rrow, n, cnt,
              pin_nbr);
  GG_TRY(scan_error_fetch(ctx));
  GG_HIP(hipStreamSynchronize(s));
  GG_TRY(scan_error_test(ctx));
  ctx->dev_free(cnt);
  csr->pin_off = pin_off;
  csr->pin_nbr = pin_nbr;
  ctx->keep(pin_off);
  ctx->keep(pin_nbr);
  return GG_OK;
}

// statistics of the frontier a sharded level starts from, against this rank's push rows
__global__ __launch_bounds__(256) void k_shard_frontier_stats(const uint64_t *__restrict__ front, uint64_t V,
                                                              const uint32_t *__restrict__ pin_off,
                                                              BfsLevel *__restrict__ st) {
  __shared__ uint64_t s_red[12];
  uint64_t act = 0, te = 0;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (uint64_t)gridDim.x * blockDim.x) {
    if (front[v]) {
      act += 1;
      te += pin_off[v + 1] - pin_off[v];
    }
  }
  add_level_stats(act, te, 0, s_red, st);
}
__device__ __forceinline__ bool shard_pushes(const BfsLevel &st, uint64_t E_rev) {
  return st.te * GG_BFS_PULL_FACTOR <= E_rev;  // (an empty frontier pushes nothing and folds nothing: cheapest)
}
// push along the pin rows: the same walk as k_bfs_push_dev, into `acc`
__global__ __launch_bounds__(256) void k_shard_push(const BfsLevel *__restrict__ st, uint64_t E_rev,
                                                    const uint64_t *__restrict__ frontier, uint64_t *__restrict__ acc,
                                                    const uint64_t *__restrict__ seen,
                                                    const uint32_t *__restrict__ pin_off,
                                                    const uint32_t *__restrict__ pin_nbr, uint64_t V) {
  if (!shard_pushes(*st, E_rev)) return;
  const int lane = threadIdx.x & 63;
  const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t base = wave0 * 64; base < V; base += nwaves * 64) {
    const uint64_t mine = base + lane < V ? frontier[base + lane] : 0ULL;
    uint64_t m = __ballot(mine != 0);
    while (m) {  // uniform
      const int l = __ffsll((long long)m) - 1;
      m &= m - 1;
      const uint64_t f = __shfl(mine, l, 64);
      const uint32_t b = pin_off[base + l], e = pin_off[base + l + 1];
      for (uint32_t i = b + lane; i < e; i += 64) {
        const uint32_t w = pin_nbr[i];
        const uint64_t nf = f & ~seen[w];
        if (nf) atomicOr((unsigned long long *)&acc[w], (unsigned long long)nf);
      }
    }
  }
}
// second kernel of a sharded level: fold the pushes into the exchange words, or pull them
template <typename DistT>
__global__ __launch_bounds__(256) void k_shard_finish(const BfsLevel *__restrict__ st, uint64_t E_rev,
                                                      const uint64_t *__restrict__ front, uint64_t *__restrict__ next,
                                                      uint64_t *__restrict__ acc, uint64_t *__restrict__ seen, uint64_t V,
                                                      uint32_t level, const uint32_t *__restrict__ off,
                                                      const uint32_t *__restrict__ roff,
                                                      const uint32_t *__restrict__ rnbr, uint64_t *__restrict__ dist8,
                                                      BfsLevel *__restrict__ lv, uint32_t update_blocks) {
  __shared__ uint64_t s_red[12];
  if (shard_pushes(*st, E_rev)) {
    if (blockIdx.x >= update_blocks) return;
    update_body<DistT>(next, seen, acc, V, level, off, dist8, lv, s_red, update_blocks);
  } else {
    pull_body<DistT>(front, next, seen, V, level, off, roff, rnbr, dist8, lv, s_red);
  }
}

}  // namespace gg

extern "C" void gg_bfs_sharded_end(gg_bfs_run *run) {
  if (!run) return;
  gg_ctx *ctx = run->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (void *p : {(void *)run->ids_dev, (void *)run->front, (void *)run->next, (void *)run->seen, (void *)run->dist8,
                  (void *)run->acc, (void *)run->lv})
    ctx->dev_free(p);
  delete run;
}

extern "C" int gg_bfs_sharded_begin(gg_ctx *ctx, const gg_csr *csr_c, const int64_t *src_ids, int n_src,
                                    gg_bfs_run **out) {
  if (!out) return GG_ERR_INVALID_ARG;
  *out = nullptr;
  gg_csr *csr = const_cast<gg_csr *>(csr_c);
  if (!ctx || !csr || csr->ctx != ctx || n_src < 0 || n_src > GG_BFS_LANES || (n_src && !src_ids))
    return GG_ERR_INVALID_ARG;
  if (!csr->roff || !csr->rnbr) GG_TRY(ensure_reverse(ctx, csr));  // a whole CSR works too (one shard)
  ApiScope scope(ctx);
  GG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint64_t V = csr->V ? csr->V : 1;
  gg_bfs_run *run = new gg_bfs_run();
  run->ctx = ctx;
  run->csr = csr;
  run->n_src = n_src;
  struct Guard {
    gg_bfs_run *r;
    bool armed = true;
    ~Guard() {
      if (armed) gg_bfs_sharded_end(r);
    }
  } guard{run};
  uint32_t *src_dense = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&run->ids_dev, GG_BFS_LANES * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&src_dense, GG_BFS_LANES * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&run->front, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&run->next, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&run->seen, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&run->dist8, V * 64 * sizeof(uint8_t)));
  GG_TRY(ctx->dev_alloc((void **)&run->acc, V * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&run->lv, 2 * sizeof(BfsLevel)));
  for (void *p : {(void *)run->ids_dev, (void *)run->front, (void *)run->next, (void *)run->seen, (void *)run->dist8,
                  (void *)run->acc, (void *)run->lv})
    ctx->keep(p);
  if (csr->V && csr->n_parts > 1) {  // (a whole CSR pushes along its forward rows)
    if (!csr->rrow) {
      set_error("gg_bfs_sharded: the shard has no reverse COO column");
      return GG_ERR_INVALID_ARG;
    }
    GG_TRY(ensure_push_in(ctx, csr));
  }
  if (n_src) GG_HIP(hipMemcpyAsync(run->ids_dev, src_ids, (size_t)n_src * sizeof(int64_t), hipMemcpyHostToDevice, s));
  GG_HIP(hipStreamSynchronize(s));
  GG_TRY(lookup_ids(ctx, csr, run->ids_dev, (uint64_t)n_src, src_dense));
  GG_HIP(hipMemsetAsync(run->front, 0, V * sizeof(uint64_t), s));
  GG_HIP(hipMemsetAsync(run->next, 0, V * sizeof(uint64_t), s));
  GG_HIP(hipMemsetAsync(run->seen, 0, V * sizeof(uint64_t), s));
  GG_HIP(hipMemsetAsync(run->dist8, 0xFF, V * 64 * sizeof(uint8_t), s));
  GG_HIP(hipMemsetAsync(run->acc, 0, V * sizeof(uint64_t), s));
  GG_HIP(hipMemsetAsync(run->lv, 0, 2 * sizeof(BfsLevel), s));
  if (csr->V)
    GG_LAUNCH(ctx, "bfs_seed", (k_bfs_seed<uint8_t>), dim3(1), dim3(64), 0, src_dense, n_src, csr->off, run->front,
              run->seen, run->dist8, run->lv, (const int64_t *)csr->vid, (uint32_t)csr->part, (uint32_t)csr->n_parts);
  GG_HIP(hipStreamSynchronize(s));
  guard.armed = false;
  *out = run;
  return GG_OK;
}

extern "C" int gg_bfs_sharded_expand(gg_bfs_run *run, void **next_words_dev, uint64_t *n_words,
                                     uint64_t *new_pairs_local) {
  if (!run) return GG_ERR_INVALID_ARG;
  gg_ctx *ctx = run->ctx;
  gg_csr *csr = run->csr;
  GG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (run->level >= 254) {
    set_error("gg_bfs_sharded: deeper than 254 levels is not supported");
    return GG_ERR_TOO_LARGE;
  }
  const uint64_t V = csr->V;
  GG_HIP(hipMemsetAsync(run->lv, 0, 2 * sizeof(BfsLevel), s));
  run->level++;
  if (V) {
    // three launches, the direction decided on the device from the frontier's statistics (no read-back in between)
    const uint32_t *pin_off = csr->n_parts > 1 ? csr->pin_off : csr->off;
    const uint32_t *pin_nbr = csr->n_parts > 1 ? csr->pin_nbr : csr->nbr;
    const uint64_t E_rev = csr->E_rev;
    const unsigned vgrid = (unsigned)((V + 255) / 256);
    const unsigned ugrid = vgrid < (unsigned)ctx->num_cus * 2 ? vgrid : (unsigned)ctx->num_cus * 2;
    const uint64_t max_waves = (uint64_t)ctx->num_cus * 32;
    const uint64_t chunks = (V + 63) / 64;
    const unsigned push_grid = (unsigned)(((chunks < max_waves ? chunks : max_waves) * 64 + 255) / 256);
    const uint64_t quads = (V + 64 / GG_PULL_LANES - 1) / (64 / GG_PULL_LANES);
    const unsigned pull_grid = (unsigned)(((quads < max_waves ? quads : max_waves) * 64 + 255) / 256);
    GG_LAUNCH(ctx, "bfs_shard_stats", k_shard_frontier_stats, dim3(ugrid), dim3(256), 0, run->front, V, pin_off,
              run->lv + 1);
    GG_LAUNCH(ctx, "bfs_shard_push", k_shard_push, dim3(push_grid), dim3(256), 0, run->lv + 1, E_rev, run->front, run->acc,
              run->seen, pin_off, pin_nbr, V);
    GG_LAUNCH(ctx, "bfs_shard_finish", (k_shard_finish<uint8_t>), dim3(pull_grid > ugrid ? pull_grid : ugrid), dim3(256), 0,
              run->lv + 1, E_rev, run->front, run->next, run->acc, run->seen, V, run->level, csr->off, csr->roff, csr->rnbr,
              run->dist8, run->lv, ugrid);
  }
  GG_HIP(hipMemcpyAsync(ctx->pin_scratch, run->lv, 2 * sizeof(BfsLevel), hipMemcpyDeviceToHost, s));
  GG_HIP(hipStreamSynchronize(s));  // the words are complete in memory: the caller's collective may read them
  BfsLevel h, st;
  memcpy(&h, ctx->pin_scratch, sizeof(h));
  memcpy(&st, (const char *)ctx->pin_scratch + sizeof(h), sizeof(st));
  if (V) (st.te * GG_BFS_PULL_FACTOR <= csr->E_rev ? run->levels_push : run->levels_pull)++;
  if (next_words_dev) *next_words_dev = run->next;
  if (n_words) *n_words = V;
  if (new_pairs_local) *new_pairs_local = h.reached;
  return GG_OK;
}

extern "C" int gg_bfs_sharded_levels(const gg_bfs_run *run, uint64_t *push_levels, uint64_t *pull_levels) {
  if (!run) return GG_ERR_INVALID_ARG;
  if (push_levels) *push_levels = run->levels_push;
  if (pull_levels) *pull_levels = run->levels_pull;
  return GG_OK;
}

extern "C" int gg_bfs_sharded_words(gg_bfs_run *run, uint64_t *host_words, int write_back) {
  if (!run || !host_words) return GG_ERR_INVALID_ARG;
  gg_ctx *ctx = run->ctx;
  GG_HIP(hipSetDevice(ctx->device));
  const size_t bytes = run->csr->V * sizeof(uint64_t);
  if (write_back)
    GG_HIP(hipMemcpyAsync(run->next, host_words, bytes, hipMemcpyHostToDevice, ctx->stream));
  else
    GG_HIP(hipMemcpyAsync(host_words, run->next, bytes, hipMemcpyDeviceToHost, ctx->stream));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  return GG_OK;
}

extern "C" int gg_bfs_sharded_commit(gg_bfs_run *run) {
  if (!run) return GG_ERR_INVALID_ARG;
  uint64_t *t = run->front;  // the combined words are the next level's frontier
  run->front = run->next;
  run->next = t;
  return GG_OK;
}

extern "C" int gg_bfs_sharded_pairs(gg_bfs_run *run, gg_result **out_result) {
  if (!run || !out_result) return GG_ERR_INVALID_ARG;
  *out_result = nullptr;
  gg_ctx *ctx = run->ctx;
  gg_csr *csr = run->csr;
  ApiScope scope(ctx);
  GG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint64_t V = csr->V;
  gg_result *res = new gg_result();
  res->ctx = ctx;
  res->k_min = res->k_max = 2;
  struct Guard {
    gg_result *r;
    bool armed = true;
    ~Guard() {
      if (armed) gg_result_destroy(r);
    }
  } guard{res};
  if (V) {
    uint32_t *counts = nullptr;
    uint64_t *total = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&counts, V * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&total, sizeof(uint64_t)));
    const unsigned vgrid = (unsigned)((V + 255) / 256);
    GG_LAUNCH(ctx, "bfs_pairs_count", (k_bfs_pairs_count<uint8_t>), dim3(vgrid), dim3(256), 0,
              (const uint8_t *)run->dist8, V, run->n_src, counts);
    GG_TRY(scan_exclusive_u32(ctx, counts, counts, V, total));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, total, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    GG_TRY(scan_error_fetch(ctx));
    GG_HIP(hipStreamSynchronize(s));
    GG_TRY(scan_error_test(ctx));
    const uint64_t rows = ctx->pin_scratch[0];
    if (rows) {
      for (int c = 0; c < 3; c++) GG_TRY(ctx->dev_alloc((void **)&res->cols[2][c], rows * sizeof(int64_t)));
      GG_LAUNCH(ctx, "bfs_pairs_fill", (k_bfs_pairs_fill<uint8_t>), dim3(vgrid), dim3(256), 0, (const uint8_t *)run->dist8,
                V, run->n_src, (const uint32_t *)counts, (const int64_t *)run->ids_dev, (const int64_t *)csr->vid,
                res->cols[2][0], res->cols[2][1], res->cols[2][2]);
      for (int c = 0; c < 3; c++) ctx->keep(res->cols[2][c]);
      res->rows[2] = rows;
    }
  }
  GG_HIP(hipStreamSynchronize(s));
  guard.armed = false;
  *out_result = res;
  return GG_OK;
}

extern "C" int gg_walk_endpoints(gg_ctx *ctx, const gg_csr *csr_c, const int64_t *src_ids, uint64_t n_src, int k_max,
                                 gg_result **out_result) {
  if (!out_result) return GG_ERR_INVALID_ARG;
  *out_result = nullptr;
  gg_csr *csr = const_cast<gg_csr *>(csr_c);
  ApiScope scope(ctx);
  if (!ctx || !csr || csr->ctx != ctx || (n_src && !src_ids) || k_max < 1 || k_max > GG_MAX_HOPS) {
    set_error("gg_walk_endpoints: bad argument (1 <= k_max <= %d)", GG_MAX_HOPS);
    return GG_ERR_INVALID_ARG;
  }
  if (csr->n_parts > 1) {
    set_error("gg_walk_endpoints needs a whole CSR, not a shard");
    return GG_ERR_STATE;
  }
  GG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint64_t V = csr->V;
  gg_result *res = new gg_result();
  res->ctx = ctx;
  res->k_min = res->k_max = 1;  // table 1: (vertex id, mask of walk lengths)
  struct Guard {  // hand the result out only on success
    gg_result *r;
    ~Guard() {
      if (r) gg_result_destroy(r);
    }
  } guard{res};
  if (V && n_src) {
    const uint64_t words = (V + 31) / 32;
    int64_t *ids_dev = nullptr;
    uint32_t *dense = nullptr, *bits = nullptr, *counts = nullptr;
    uint64_t *total = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&ids_dev, n_src * sizeof(int64_t)));
    GG_TRY(ctx->dev_alloc((void **)&dense, n_src * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&bits, (uint64_t)(k_max + 1) * words * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&counts, V * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&total, sizeof(uint64_t)));
    GG_HIP(hipMemcpyAsync(ids_dev, src_ids, n_src * sizeof(int64_t), hipMemcpyHostToDevice, s));
    GG_HIP(hipStreamSynchronize(s));  // src_ids is caller memory: consumed before return
    GG_HIP(hipMemsetAsync(bits, 0, (uint64_t)(k_max + 1) * words * sizeof(uint32_t), s));
    GG_TRY(lookup_ids(ctx, csr, ids_dev, n_src, dense));
    GG_LAUNCH(ctx, "set_seed", k_set_seed, dim3((unsigned)((n_src + 255) / 256)), dim3(256), 0, (const uint32_t *)dense, n_src,
              bits);
    const unsigned vgrid = (unsigned)((V + 255) / 256), xgrid = (unsigned)((V * 16 + 255) / 256);
    for (int h = 1; h <= k_max; h++)
      GG_LAUNCH(ctx, "set_expand", k_set_expand, dim3(xgrid), dim3(256), 0, csr->off, csr->nbr,
                (const uint32_t *)(bits + (uint64_t)(h - 1) * words), bits + (uint64_t)h * words, V);
    GG_LAUNCH(ctx, "endpoint_count", k_endpoint_count, dim3(vgrid), dim3(256), 0, (const uint32_t *)bits, words, k_max, V,
              counts);
    GG_TRY(scan_exclusive_u32(ctx, counts, counts, V, total));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, total, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    GG_TRY(scan_error_fetch(ctx));
    GG_HIP(hipStreamSynchronize(s));
    GG_TRY(scan_error_test(ctx));
    const uint64_t rows = ctx->pin_scratch[0];
    if (rows) {
      for (int c = 0; c < 2; c++) GG_TRY(ctx->dev_alloc((void **)&res->cols[1][c], rows * sizeof(int64_t)));
      GG_LAUNCH(ctx, "endpoint_fill", k_endpoint_fill, dim3(vgrid), dim3(256), 0, (const uint32_t *)bits, words, k_max, V,
                (const uint32_t *)counts, (const int64_t *)csr->vid, res->cols[1][0], res->cols[1][1]);
      GG_HIP(hipStreamSynchronize(s));
      for (int c = 0; c < 2; c++) ctx->keep(res->cols[1][c]);
      res->rows[1] = rows;
    }
  }
  guard.r = nullptr;
  *out_result = res;
  return GG_OK;
}
