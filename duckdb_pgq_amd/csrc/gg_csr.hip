// gg_csr.hip — CSR construction from the staged vertex/edge base-table columns (Finalize side).
//
// Replaces the reference's adjacency index = hash-join build side:
//   PhysicalHashJoin::Finalize -> JoinHashTable::Finalize/InsertHashes
//   (src/execution/operator/join/physical_hash_join.cpp:165-185, src/execution/join_hashtable.cpp:240-302),
// which inserts every build row single-threaded into a chained pointer table.  Here:
//   1. k_ht_insert       vertex ids -> open-addressing id hash table (16-byte slots, one cache line
//                        per probe; dense index = vertex-table position); k_id_minmax / k_direct_*: a
//                        direct-address dictionary instead when the ids span < 2^20 values (the
//                        reference's perfect-hash-join case)
//   2. k_densify_hist    (src,dst) ids -> dense (u,v); edges with a non-vertex endpoint get an invalid
//                        key; fused with the first radix pass's per-wave digit histogram (LDS)
//   3. LSD radix passes  stable radix-bucket scatter of (u, v, edge position) by u: global prefix scan
//                        over (digit, wave) counters, then a wave-ordered scatter whose in-wave ranks
//                        come from __ballot match masks.  Stable => inside a CSR row neighbours keep
//                        ascending edge-rowid (append) order; no atomic ever decides a position, so the
//                        build is bit-reproducible.
//   4. k_row_offsets     row offsets from the sorted source column (run boundaries) — no degree
//                        histogram, hence no global atomics at all on the per-edge path.
// No host synchronisation happens between the first launch and the final status read-back.
// The reverse CSR (in-neighbour lists, needed by the 2-hop product kernel and by pull-style BFS) is
// derived from the forward CSR's COO view by the same radix machinery (ensure_reverse).
// gg_vertices_from_edges (end of the file) derives the vertex table itself — the sorted distinct endpoint
// ids — for join chains that name no vertex table.
// All integer work, HBM-bound: algorithmic bytes 40E + 16V (with rowid).
#include "gg_internal.h"
#include "gg_dict.h"

using namespace gg;

namespace gg {

__global__ __launch_bounds__(256) void k_rowid_iota(int64_t *__restrict__ out, int64_t first, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = first + (int64_t)i;
}

__global__ __launch_bounds__(256) void k_ht_init(HtSlot *__restrict__ ht, uint64_t cap) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) {
    uint4 e;
    e.x = 0u;
    e.y = 0x80000000u;  // key = INT64_MIN
    e.z = INVALID_U32;
    e.w = 0u;
    *reinterpret_cast<uint4 *>(&ht[i]) = e;
  }
}

__global__ __launch_bounds__(256) void k_ht_insert(const int64_t *__restrict__ vid, uint64_t V,
                                                   HtSlot *__restrict__ ht, uint64_t cap,
                                                   BuildStatus *__restrict__ st, uint32_t part, uint32_t n_parts) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_parts > 1) {  // owned-vertex count (whole builds own everything: set on the host)
    __shared__ uint32_t s_owned;
    if (threadIdx.x == 0) s_owned = 0;
    __syncthreads();
    const bool mine = i < V && owns(vid[i < V ? i : 0], part, n_parts);
    const uint64_t om = __ballot(mine);
    if ((threadIdx.x & 63) == 0 && om) atomicAdd(&s_owned, (uint32_t)__popcll(om));
    __syncthreads();
    if (threadIdx.x == 0 && s_owned) atomicAdd(&st->owned, (unsigned long long)s_owned);
  }
  if (i >= V) return;
  int64_t key = vid[i];
  if (key == HT_EMPTY) {  // the sentinel value itself is a legal id: keep it outside the table
    long long prev = (long long)atomicCAS((unsigned long long *)&st->min_idx, (unsigned long long)-1LL,
                                          (unsigned long long)i);
    if (prev != -1LL) atomicOr(&st->dup_vertex, 1ULL);
    return;
  }
  uint64_t slot = ht_slot(key, cap);
  while (true) {
    unsigned long long prev =
        atomicCAS((unsigned long long *)&ht[slot].key, (unsigned long long)HT_EMPTY, (unsigned long long)key);
    if (prev == (unsigned long long)HT_EMPTY) {
      ht[slot].val = (uint32_t)i;
      return;
    }
    if (prev == (unsigned long long)key) {
      atomicOr(&st->dup_vertex, 1ULL);
      return;
    }
    slot = ht_next(slot, cap);
  }
}

// ---- stable LSD radix machinery -----------------------------------------------------------------
// One workgroup owns a tile of RB_TILE consecutive elements.  Inside the tile, wave w owns the
// contiguous quarter [w*1024, (w+1)*1024) and walks it in order, 64 elements per step, so element
// order is fully determined (stable).  Elements are first ranked INSIDE the tile and staged in LDS in
// digit order, then written out: consecutive LDS slots of one digit go to consecutive global
// addresses, so the global stores are coalesced runs instead of 4-byte scatters.
#ifndef GG_RB_THREADS
#define GG_RB_THREADS 512
#endif
constexpr int RB_THREADS = GG_RB_THREADS;          // 8 waves share one 4096-element tile (LDS-bound occupancy)
constexpr int RB_WAVES = RB_THREADS / 64;
#ifndef GG_RB_ITEMS
#define GG_RB_ITEMS 8
#endif
constexpr int RB_ITEMS = GG_RB_ITEMS;              // elements per lane
constexpr int RB_TILE = RB_THREADS * RB_ITEMS;     // 4096 elements per workgroup
constexpr int RB_WTILE = RB_TILE / RB_WAVES;       // 1024 per wave
constexpr int RB_MAX_BITS = 8;                     // <= 256 digits per pass

// Densify + pass-0 histogram.  One lane per edge row: two id lookups (hash table is V-sized, L2 /
// Infinity-Cache resident), dense endpoints written coalesced, digit counted in the tile's LDS histogram.
// counts[digit * nblocks + block] = number of valid elements of that tile with that digit
__global__ __launch_bounds__(RB_THREADS) void k_densify_hist(const int64_t *__restrict__ src,
                                                             const int64_t *__restrict__ dst, uint64_t E,
                                                             const HtSlot *__restrict__ ht, uint64_t cap,
                                                             const BuildStatus *__restrict__ st,
                                                             const uint32_t *__restrict__ dir,
                                                             const DirectMap *__restrict__ dm,
                                                             uint32_t *__restrict__ fk, uint32_t *__restrict__ fv,
                                                             uint32_t bits, uint64_t nblocks,
                                                             uint32_t *__restrict__ counts) {
  __shared__ uint32_t hist[1 << RB_MAX_BITS];
  const uint32_t ndig = 1u << bits;
  for (uint32_t i = threadIdx.x; i < ndig; i += RB_THREADS) hist[i] = 0;
  __syncthreads();
  const int64_t min_idx = st->min_idx;
  const uint64_t base = (uint64_t)blockIdx.x * RB_TILE;
  if (dm->enabled) {  // uniform over the grid: dense ids, one 4-byte read per endpoint
    const uint64_t min_id = (uint64_t)dm->min_id;
    int64_t ks[RB_ITEMS], kd[RB_ITEMS];
    uint32_t us[RB_ITEMS], vs[RB_ITEMS];
#pragma unroll
    for (int it = 0; it < RB_ITEMS; it++) {  // all row loads first, then all dictionary reads
      const uint64_t e = base + (uint64_t)it * RB_THREADS + threadIdx.x;
      ks[it] = e < E ? src[e] : 0;
      kd[it] = e < E ? dst[e] : 0;
    }
#pragma unroll
    for (int it = 0; it < RB_ITEMS; it++) {
      const uint64_t e = base + (uint64_t)it * RB_THREADS + threadIdx.x;
      us[it] = e < E ? direct_lookup(dir, min_id, ks[it]) : INVALID_U32;
      vs[it] = e < E ? direct_lookup(dir, min_id, kd[it]) : INVALID_U32;
    }
#pragma unroll
    for (int it = 0; it < RB_ITEMS; it++) {
      const uint64_t e = base + (uint64_t)it * RB_THREADS + threadIdx.x;
      if (e >= E) continue;
      uint32_t u = us[it], v = vs[it];
      if (u == INVALID_U32 || v == INVALID_U32) {
        u = INVALID_U32;
        v = INVALID_U32;
      } else {
        atomicAdd(&hist[u & (ndig - 1)], 1u);
      }
      fk[e] = u;
      fv[e] = v;
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < ndig; d += RB_THREADS) counts[(uint64_t)d * nblocks + blockIdx.x] = hist[d];
    return;
  }
  constexpr int B = 4;  // edges per batch: 2*B independent first-probe loads in flight per lane
  for (int it0 = 0; it0 < RB_ITEMS; it0 += B) {
    int64_t ks[B], kd[B];
    uint64_t ss[B], sd[B];
    uint4 rs[B], rd[B];
#pragma unroll
    for (int j = 0; j < B; j++) {
      const uint64_t e = base + (uint64_t)(it0 + j) * RB_THREADS + threadIdx.x;
      ks[j] = e < E ? src[e] : HT_EMPTY;
      kd[j] = e < E ? dst[e] : HT_EMPTY;
    }
#pragma unroll
    for (int j = 0; j < B; j++) {  // first probes of the whole batch issue back to back
      ss[j] = ht_slot(ks[j], cap);
      sd[j] = ht_slot(kd[j], cap);
      rs[j] = *reinterpret_cast<const uint4 *>(&ht[ss[j]]);
      rd[j] = *reinterpret_cast<const uint4 *>(&ht[sd[j]]);
    }
#pragma unroll
    for (int j = 0; j < B; j++) {
      const uint64_t e = base + (uint64_t)(it0 + j) * RB_THREADS + threadIdx.x;
      if (e >= E) continue;
      uint32_t u = ht_resolve(ht, cap, min_idx, ks[j], ss[j], rs[j]);
      uint32_t v = ht_resolve(ht, cap, min_idx, kd[j], sd[j], rd[j]);
      if (u == INVALID_U32 || v == INVALID_U32) {
        u = INVALID_U32;  // dropped: an endpoint is not a vertex (inner-join semantics)
        v = INVALID_U32;
      } else {
        atomicAdd(&hist[u & (ndig - 1)], 1u);
      }
      fk[e] = u;
      fv[e] = v;
    }
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < ndig; d += RB_THREADS) counts[(uint64_t)d * nblocks + blockIdx.x] = hist[d];
}

// Shard builds (gg_csr_build_shard): only the rows this rank owns survive, 1/N of the table per
// direction, so they are COMPACTED inside their tile (stable: element order kept) and the first radix
// pass of each direction reads only the tile's valid prefix (tile_valid[]).  Non-owned rows are dropped
// before the id lookups.  Forward stream: (u, v) of edges whose SOURCE is owned; reverse stream: (v, u) of
// edges whose DESTINATION is owned.
__global__ __launch_bounds__(RB_THREADS) void k_densify_shard(
    const int64_t *__restrict__ src, const int64_t *__restrict__ dst, uint64_t E, const HtSlot *__restrict__ ht,
    uint64_t cap, const BuildStatus *__restrict__ st, uint32_t part, uint32_t n_parts, uint32_t *__restrict__ fk,
    uint32_t *__restrict__ fv, uint32_t *__restrict__ rk, uint32_t *__restrict__ rv, uint32_t bits, uint64_t nblocks,
    uint32_t *__restrict__ counts, uint32_t *__restrict__ counts_r, uint32_t *__restrict__ tile_valid_f,
    uint32_t *__restrict__ tile_valid_r) {
  __shared__ uint32_t hist[1 << RB_MAX_BITS];
  __shared__ uint32_t hist_r[1 << RB_MAX_BITS];
  __shared__ uint32_t s_wf[RB_WAVES], s_wr[RB_WAVES];
  const uint32_t ndig = 1u << bits;
  for (uint32_t i = threadIdx.x; i < ndig; i += RB_THREADS) {
    hist[i] = 0;
    hist_r[i] = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t min_idx = st->min_idx;
  const uint64_t base = (uint64_t)blockIdx.x * RB_TILE;
  uint32_t u[RB_ITEMS], v[RB_ITEMS];
  uint32_t fmask = 0, rmask = 0;  // bit `it`: item survives in the forward / reverse stream
  constexpr int B = 4;
  for (int it0 = 0; it0 < RB_ITEMS; it0 += B) {
    int64_t ks[B], kd[B];
    uint64_t ss[B], sd[B];
    uint4 rs[B], rd[B];
    bool of[B], orv[B];
#pragma unroll
    for (int j = 0; j < B; j++) {
      const uint64_t e = base + (uint64_t)(it0 + j) * RB_THREADS + threadIdx.x;
      ks[j] = e < E ? src[e] : HT_EMPTY;
      kd[j] = e < E ? dst[e] : HT_EMPTY;
      of[j] = e < E && owns(ks[j], part, n_parts);
      orv[j] = e < E && owns(kd[j], part, n_parts);
    }
#pragma unroll
    for (int j = 0; j < B; j++) {
      ss[j] = ht_slot(ks[j], cap);
      sd[j] = ht_slot(kd[j], cap);
      if (of[j] || orv[j]) {
        rs[j] = *reinterpret_cast<const uint4 *>(&ht[ss[j]]);
        rd[j] = *reinterpret_cast<const uint4 *>(&ht[sd[j]]);
      }
    }
#pragma unroll
    for (int j = 0; j < B; j++) {
      uint32_t uu = INVALID_U32, vv = INVALID_U32;
      if (of[j] || orv[j]) {
        uu = ht_resolve(ht, cap, min_idx, ks[j], ss[j], rs[j]);
        vv = ht_resolve(ht, cap, min_idx, kd[j], sd[j], rd[j]);
      }
      const bool ok = uu != INVALID_U32 && vv != INVALID_U32;
      u[it0 + j] = uu;
      v[it0 + j] = vv;
      if (ok && of[j]) {
        fmask |= 1u << (it0 + j);
        atomicAdd(&hist[uu & (ndig - 1)], 1u);
      }
      if (ok && orv[j]) {
        rmask |= 1u << (it0 + j);
        atomicAdd(&hist_r[vv & (ndig - 1)], 1u);
      }
    }
  }
  // stable compaction inside the tile: element order = (item, thread)
  uint32_t base_f = 0, base_r = 0;
  const uint64_t lane_lt = (1ULL << lane) - 1ULL;
#pragma unroll
  for (int it = 0; it < RB_ITEMS; it++) {
    const bool f = (fmask >> it) & 1u, r = (rmask >> it) & 1u;
    const uint64_t bf = __ballot(f), br = __ballot(r);
    if (lane == 0) {
      s_wf[wave] = (uint32_t)__popcll(bf);
      s_wr[wave] = (uint32_t)__popcll(br);
    }
    __syncthreads();
    uint32_t wf = 0, wr = 0, tf = 0, tr = 0;
#pragma unroll
    for (int w = 0; w < RB_WAVES; w++) {
      if (w < wave) {
        wf += s_wf[w];
        wr += s_wr[w];
      }
      tf += s_wf[w];
      tr += s_wr[w];
    }
    if (f) {
      const uint64_t o = base + base_f + wf + __popcll(bf & lane_lt);
      fk[o] = u[it];
      fv[o] = v[it];
    }
    if (r) {
      const uint64_t o = base + base_r + wr + __popcll(br & lane_lt);
      rk[o] = v[it];
      rv[o] = u[it];
    }
    base_f += tf;
    base_r += tr;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tile_valid_f[blockIdx.x] = base_f;
    tile_valid_r[blockIdx.x] = base_r;
  }
  for (uint32_t d = threadIdx.x; d < ndig; d += RB_THREADS) {
    counts[(uint64_t)d * nblocks + blockIdx.x] = hist[d];
    counts_r[(uint64_t)d * nblocks + blockIdx.x] = hist_r[d];
  }
}

// histogram of a later pass: n comes from device memory (edges kept is only known on the device).
// Per-wave private LDS histograms (8 x ndig) keep same-address LDS atomics apart; summed at the end.
__global__ __launch_bounds__(RB_THREADS) void k_radix_hist(const uint32_t *__restrict__ key,
                                                           const unsigned long long *__restrict__ n_dev,
                                                           uint32_t lo_bit, uint32_t bits, uint64_t nblocks,
                                                           uint32_t *__restrict__ counts) {
  __shared__ uint32_t hist[RB_WAVES << RB_MAX_BITS];
  const uint32_t ndig = 1u << bits;
  const int wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < RB_WAVES * ndig; i += RB_THREADS) hist[i] = 0;
  __syncthreads();
  const uint64_t n = *n_dev;
  const uint64_t base = (uint64_t)blockIdx.x * RB_TILE;
  uint32_t *h = hist + wave * ndig;
  uint32_t k[RB_ITEMS];
#pragma unroll
  for (int it = 0; it < RB_ITEMS; it++) {  // all loads first: 8 independent 4-byte loads in flight per lane
    const uint64_t idx = base + (uint64_t)it * RB_THREADS + threadIdx.x;
    k[it] = idx < n ? key[idx] : INVALID_U32;
  }
#pragma unroll
  for (int it = 0; it < RB_ITEMS; it++)
    if (k[it] != INVALID_U32) atomicAdd(&h[(k[it] >> lo_bit) & (ndig - 1)], 1u);
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < ndig; d += RB_THREADS) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < RB_WAVES; w++) c += hist[w * ndig + d];
    counts[(uint64_t)d * nblocks + blockIdx.x] = c;
  }
}

// Scatter.  `bases` is the exclusive scan of `counts` (same layout).
//   GEN_B : payload b is the element's own position (pass 0: the edge's append position)
//   HAS_B : a second payload column exists
template <bool HAS_B, bool GEN_B>
__global__ __launch_bounds__(RB_THREADS) void k_radix_scatter(
    const uint32_t *__restrict__ key_in, const uint32_t *__restrict__ a_in, const uint32_t *__restrict__ b_in,
    uint64_t n_host, const unsigned long long *__restrict__ n_dev, uint32_t lo_bit, uint32_t bits, uint64_t nblocks,
    const uint32_t *__restrict__ bases, const uint32_t *__restrict__ tile_valid /* nullable: valid prefix per tile */,
    uint32_t *__restrict__ key_out, uint32_t *__restrict__ a_out, uint32_t *__restrict__ b_out,
    const unsigned long long *__restrict__ err) {
  extern __shared__ uint32_t lds[];
  // a scan that gave up leaves prefixes that are too small: this pass would leave slots of its output unwritten,
  // and the next kernels would take whatever those slots hold for keys.  Nothing is written after an error.
  if (*err) return;
  const uint32_t ndig = 1u << bits;
  uint32_t *xk = lds;                                  // staged keys, digit order
  uint32_t *xa = xk + RB_TILE;
  uint32_t *xb = xa + RB_TILE;                         // only if HAS_B
  uint32_t *hw = HAS_B ? xb + RB_TILE : xb;            // [RB_WAVES][ndig] per-wave counts -> running cursors
  uint32_t *dbase = hw + RB_WAVES * ndig;              // [ndig] first tile slot of each digit
  uint32_t *gb = dbase + ndig;                         // [ndig] global position of tile slot 0 of the digit, minus dbase
  uint32_t *misc = gb + ndig;                          // [RB_WAVES + 1] block-scan scratch + valid count

  const uint64_t n = n_dev ? (uint64_t)*n_dev : n_host;
  const uint64_t tile_base = (uint64_t)blockIdx.x * RB_TILE;
  if (tile_base >= n) return;  // block-uniform
  const uint32_t tcnt = tile_valid ? tile_valid[blockIdx.x] : (uint32_t)RB_TILE;  // compacted tiles: valid prefix
  if (tcnt == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < RB_WAVES * ndig; i += RB_THREADS) hw[i] = 0;
  __syncthreads();

  // load (coalesced: 64 consecutive elements per wave step) and count per wave
  uint32_t k[RB_ITEMS], a[RB_ITEMS], b[RB_ITEMS];
  uint32_t *myh = hw + wave * ndig;
  const uint64_t wbase = tile_base + (uint64_t)wave * RB_WTILE;
#pragma unroll
  for (int it = 0; it < RB_ITEMS; it++) {
    const uint64_t idx = wbase + (uint64_t)it * 64 + lane;
    k[it] = INVALID_U32;
    a[it] = 0;
    b[it] = 0;
    if (idx < n && (uint32_t)(wave * RB_WTILE + it * 64 + lane) < tcnt) {
      k[it] = key_in[idx];
      a[it] = a_in[idx];
      if (HAS_B) b[it] = GEN_B ? (uint32_t)idx : b_in[idx];
    }
  }
#pragma unroll
  for (int it = 0; it < RB_ITEMS; it++)
    if (k[it] != INVALID_U32) atomicAdd(&myh[(k[it] >> lo_bit) & (ndig - 1)], 1u);
  __syncthreads();

  // per digit: wave counts -> exclusive offsets across waves; digit totals -> exclusive scan = dbase
  {
    uint32_t tot = 0;
    const uint32_t d = threadIdx.x;
    if (d < ndig) {
#pragma unroll
      for (int w = 0; w < RB_WAVES; w++) {
        const uint32_t c = hw[w * ndig + d];
        hw[w * ndig + d] = tot;
        tot += c;
      }
    }
    // block exclusive scan of tot over threads (ndig <= 256 <= RB_THREADS)
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) misc[wave] = incl;
    __syncthreads();
    uint32_t wb = 0, all = 0;
#pragma unroll
    for (int w = 0; w < RB_WAVES; w++) {
      const uint32_t sv = misc[w];
      if (w < wave) wb += sv;
      all += sv;
    }
    const uint32_t ex = wb + incl - tot;
    if (d < ndig) {
      dbase[d] = ex;
      gb[d] = bases[(uint64_t)d * nblocks + blockIdx.x] - ex;
    }
    if (threadIdx.x == 0) misc[RB_WAVES] = all;  // valid elements in this tile
  }
  __syncthreads();

  // rank inside the tile (stable) and stage in LDS in digit order
  const uint64_t lane_lt = (1ULL << lane) - 1ULL;
  volatile uint32_t *cur = myh;
#pragma unroll
  for (int it = 0; it < RB_ITEMS; it++) {
    const bool valid = (k[it] != INVALID_U32);
    const uint32_t d = (k[it] >> lo_bit) & (ndig - 1);
    uint64_t m = __ballot(valid);
    for (uint32_t bit = 0; bit < bits; bit++) {  // match mask: same digit within the wave
      const uint64_t bb = __ballot((d >> bit) & 1u);
      m &= ((d >> bit) & 1u) ? bb : ~bb;
    }
    if (valid) {
      const uint32_t pos = dbase[d] + cur[d] + __popcll(m & lane_lt);
      xk[pos] = k[it];
      xa[pos] = a[it];
      if (HAS_B) xb[pos] = b[it];
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && (m & lane_lt) == 0) cur[d] += __popcll(m);  // lowest lane of each digit group
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // write out: consecutive staged slots of one digit -> consecutive global positions
  const uint32_t nvalid = misc[RB_WAVES];
#pragma unroll
  for (int it = 0; it < RB_ITEMS; it++) {
    const uint32_t idx = (uint32_t)it * RB_THREADS + threadIdx.x;
    if (idx < nvalid) {
      const uint32_t kk = xk[idx];
      const uint32_t pos = gb[(kk >> lo_bit) & (ndig - 1)] + idx;
      key_out[pos] = kk;
      a_out[pos] = xa[idx];
      if (HAS_B) b_out[pos] = xb[idx];
    }
  }
}

// Row offsets from the sorted key column: off[u] = first position whose key >= u.
// Each thread owns 4 consecutive keys (one 16-byte load) and looks one key back.
__global__ __launch_bounds__(256) void k_row_offsets(const uint32_t *__restrict__ key, uint64_t n_host,
                                                     const unsigned long long *__restrict__ n_dev, uint64_t V,
                                                     uint32_t *__restrict__ off,
                                                     const unsigned long long *__restrict__ err) {
  if (*err) return;  // the keys are not a sorted column after a scan error (see k_radix_scatter)
  const uint64_t n = n_dev ? (uint64_t)*n_dev : n_host;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n == 0) {
    for (uint64_t u = t; u <= V; u += (uint64_t)gridDim.x * blockDim.x) off[u] = 0;
    return;
  }
  const uint64_t i0 = t * 4;
  if (i0 >= n) return;
  uint32_t k[4];
  if (i0 + 4 <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(key + i0);
    k[0] = v.x;
    k[1] = v.y;
    k[2] = v.z;
    k[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) k[j] = i0 + j < n ? key[i0 + j] : 0;
  }
  int64_t kp = i0 ? (int64_t)key[i0 - 1] : -1;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint64_t i = i0 + j;
    if (i >= n) break;
    for (int64_t u = kp + 1; u <= (int64_t)k[j]; u++) off[u] = (uint32_t)i;  // runs of empty rows are short
    kp = (int64_t)k[j];
    if (i == n - 1)
      for (uint64_t u = (uint64_t)k[j] + 1; u <= V; u++) off[u] = (uint32_t)n;
  }
}

__global__ __launch_bounds__(64) void k_publish_kept(const uint64_t *__restrict__ total,
                                                     const uint64_t *__restrict__ total_rev,
                                                     BuildStatus *__restrict__ st) {
  if (threadIdx.x == 0) {
    st->kept = *total;
    st->kept_rev = total_rev ? *total_rev : *total;
  }
}

__global__ __launch_bounds__(256) void k_gather_rowid(const uint32_t *__restrict__ epos,
                                                      const int64_t *__restrict__ rowid,
                                                      const unsigned long long *__restrict__ n_dev,
                                                      int64_t *__restrict__ eid,
                                                      const unsigned long long *__restrict__ err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (*err) return;  // epos is not valid after a scan error
  if (i < *n_dev) eid[i] = rowid[epos[i]];
}

}  // namespace gg

static int ceil_log2_u64(uint64_t v) {
  int b = 0;
  while ((1ULL << b) < v) b++;
  return b;
}

namespace {

// Stable sort of (key, a[, b]) by key < 2^key_bits.  Pass 0 input is (key0, a0, b generated from the
// position when gen_b); counts of pass 0 may already be available (fused producer) in counts0.
// Outputs land in (key_out, a_out, b_out).  n: host bound (grid sizing); n_dev: actual count for
// passes >= 1 (null: n is exact everywhere).  *total_dev receives the number of valid elements.
struct RadixIO {
  const uint32_t *key0, *a0, *b0;
  uint32_t *key_out, *a_out, *b_out;
};

int radix_sort_stable(gg_ctx *ctx, const RadixIO &io, uint64_t n, bool has_b, bool gen_b, int key_bits,
                      uint32_t *counts0 /* nullable: pass-0 histogram already computed */, int bits0,
                      unsigned long long *total_dev /* nullable: where pass 0's valid count goes / comes from */,
                      bool n_exact, const uint32_t *tile_valid0 = nullptr /* pass-0 input is tile-compacted */) {
  if (n == 0) return GG_OK;
  int passes = (key_bits + RB_MAX_BITS - 1) / RB_MAX_BITS;
  if (passes < 1) passes = 1;
  int bits_per = counts0 ? bits0 : (key_bits + passes - 1) / passes;
  if (bits_per < 1) bits_per = 1;
  uint32_t *kbuf[2] = {nullptr, nullptr}, *abuf[2] = {nullptr, nullptr}, *bbuf[2] = {nullptr, nullptr};
  const int nbuf = passes > 2 ? 2 : (passes > 1 ? 1 : 0);
  for (int i = 0; i < nbuf; i++) {
    GG_TRY(ctx->dev_alloc((void **)&kbuf[i], n * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&abuf[i], n * sizeof(uint32_t)));
    if (has_b) GG_TRY(ctx->dev_alloc((void **)&bbuf[i], n * sizeof(uint32_t)));
  }
  unsigned long long *own_total = nullptr;
  if (!total_dev) {
    GG_TRY(ctx->dev_alloc((void **)&own_total, sizeof(unsigned long long)));
    total_dev = own_total;
  }
  const uint32_t *kin = io.key0, *ain = io.a0, *bin = io.b0;
  const uint64_t nblocks64 = (n + RB_TILE - 1) / RB_TILE;
  const unsigned nblocks = (unsigned)nblocks64;
  for (int p = 0; p < passes; p++) {
    const bool last = (p == passes - 1);
    const uint32_t lo_bit = (uint32_t)(p * bits_per);
    int rem = key_bits - (int)lo_bit;
    const uint32_t bits = (uint32_t)(rem < bits_per ? (rem < 1 ? 1 : rem) : bits_per);
    const uint32_t ndig = 1u << bits;
    const size_t lds = ((size_t)RB_TILE * (has_b ? 3 : 2) + (size_t)(RB_WAVES + 2) * ndig + RB_WAVES + 8) * sizeof(uint32_t);
    const uint64_t ncount = (uint64_t)ndig * nblocks64;
    uint32_t *counts = nullptr;
    if (p == 0 && counts0) {
      counts = counts0;
    } else {
      GG_TRY(ctx->dev_alloc((void **)&counts, ncount * sizeof(uint32_t)));
      // passes >= 1 see only the valid elements; pass 0 without a fused producer sees n (exact)
      GG_LAUNCH(ctx, "radix_hist", k_radix_hist, dim3(nblocks), dim3(RB_THREADS), 0, kin,
                (const unsigned long long *)total_dev, lo_bit, bits, nblocks64, counts);
    }
    // pass 0's scan total is the number of valid elements = n for every later pass
    GG_TRY(scan_exclusive_u32(ctx, counts, counts, ncount, p == 0 ? (uint64_t *)total_dev : nullptr));
    uint32_t *kout = last ? io.key_out : kbuf[p & 1];
    uint32_t *aout = last ? io.a_out : abuf[p & 1];
    uint32_t *bout = last ? io.b_out : bbuf[p & 1];
    const unsigned long long *nd = (p == 0) ? nullptr : total_dev;
    if (has_b && gen_b && p == 0)
      GG_LAUNCH(ctx, "radix_scatter", (k_radix_scatter<true, true>), dim3(nblocks), dim3(RB_THREADS), lds, kin, ain,
                bin, n, nd, lo_bit, bits, nblocks64, counts, p == 0 ? tile_valid0 : nullptr, kout, aout, bout,
                (const unsigned long long *)ctx->dev_err);
    else if (has_b)
      GG_LAUNCH(ctx, "radix_scatter", (k_radix_scatter<true, false>), dim3(nblocks), dim3(RB_THREADS), lds, kin, ain,
                bin, n, nd, lo_bit, bits, nblocks64, counts, p == 0 ? tile_valid0 : nullptr, kout, aout, bout,
                (const unsigned long long *)ctx->dev_err);
    else
      GG_LAUNCH(ctx, "radix_scatter", (k_radix_scatter<false, false>), dim3(nblocks), dim3(RB_THREADS), lds, kin, ain,
                bin, n, nd, lo_bit, bits, nblocks64, counts, p == 0 ? tile_valid0 : nullptr, kout, aout, bout,
                (const unsigned long long *)ctx->dev_err);
    if (!(p == 0 && counts0)) ctx->dev_free(counts);
    kin = kout;
    ain = aout;
    bin = bout;
  }
  (void)n_exact;
  for (int i = 0; i < 2; i++) {
    ctx->dev_free(kbuf[i]);
    ctx->dev_free(abuf[i]);
    ctx->dev_free(bbuf[i]);
  }
  if (own_total) ctx->dev_free(own_total);
  return GG_OK;
}

}  // namespace

namespace gg {
// (key, value) pairs sorted by key < 2^key_bits, stable (gg_internal.h): the LSD passes of the multi-pass build
int sort_pairs_by_key(gg_ctx *ctx, const uint32_t *key, const uint32_t *val, uint64_t n, int key_bits, uint32_t *key_out,
                      uint32_t *val_out) {
  if (n == 0) return GG_OK;
  unsigned long long *tot = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&tot, sizeof(unsigned long long)));
  const unsigned long long n_host = n;  // (pageable source: staged by the runtime before the call returns)
  GG_HIP(hipMemcpyAsync(tot, &n_host, sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
  RadixIO io{key, val, nullptr, key_out, val_out, nullptr};
  GG_TRY(radix_sort_stable(ctx, io, n, false, false, key_bits < 1 ? 1 : key_bits, nullptr, 0, tot, true));
  ctx->dev_free(tot);
  return GG_OK;
}
int sort_triples_by_key(gg_ctx *ctx, const uint32_t *key, const uint32_t *a, const uint32_t *b, uint64_t n, int key_bits,
                        uint32_t *key_out, uint32_t *a_out, uint32_t *b_out) {
  if (n == 0) return GG_OK;
  unsigned long long *tot = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&tot, sizeof(unsigned long long)));
  const unsigned long long n_host = n;
  GG_HIP(hipMemcpyAsync(tot, &n_host, sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
  RadixIO io{key, a, b, key_out, a_out, b_out};
  GG_TRY(radix_sort_stable(ctx, io, n, true, false, key_bits < 1 ? 1 : key_bits, nullptr, 0, tot, true));
  ctx->dev_free(tot);
  return GG_OK;
}
}  // namespace gg

extern "C" void gg_csr_destroy(gg_csr *csr) {
  if (!csr) return;
  gg_ctx *ctx = csr->ctx;
  if (ctx) {
    // no synchronisation: the blocks go back to the context's pool, and whatever takes them next is queued on the
    // same stream behind the kernels that still read them (one stream per context; the pool never hands a block to
    // another context, and hipFree — the pool's last resort — waits for the device itself)
    ctx->dev_free(csr->off);
    ctx->dev_free(csr->nbr);
    ctx->dev_free(csr->row);
    ctx->dev_free(csr->epos);
    ctx->dev_free(csr->eid);
    ctx->dev_free(csr->vid);
    ctx->dev_free(csr->ht);
    ctx->dev_free(csr->roff);
    ctx->dev_free(csr->rnbr);
    ctx->dev_free(csr->rrow);
    ctx->dev_free(csr->pin_off);
    ctx->dev_free(csr->pin_nbr);
  }
  delete csr;
}

static int csr_build_impl(gg_ctx *ctx, int part, int n_parts, gg_csr **out) {
  if (!ctx || !out || n_parts < 1 || part < 0 || part >= n_parts) return GG_ERR_INVALID_ARG;
  *out = nullptr;
  ApiScope scope(ctx);
  const bool shard = n_parts > 1;
  GG_TRY(gg_staging_sync(ctx));
  GG_HIP(hipSetDevice(ctx->device));
  const uint64_t V = ctx->n_vertices, E = ctx->n_edges;
  hipStream_t s = ctx->stream;

  gg_csr *csr = new gg_csr();
  csr->ctx = ctx;
  csr->V = V;
  csr->E_cap = E;
  csr->part = part;
  csr->n_parts = n_parts;
  struct Guard {  // frees the half-built CSR on any early return
    gg_csr *c;
    bool armed = true;
    ~Guard() {
      if (armed) gg_csr_destroy(c);
    }
  } guard{csr};

  // ---- id hash table --------------------------------------------------------------------
  // load factor ~0.7: a smaller table keeps more of it in the 4 MiB per-XCD L2 (probes are random)
  csr->ht_cap = V + V / 2 < 1024 ? 1024 : V + (V * 2) / 5;
  GG_TRY(ctx->dev_alloc((void **)&csr->ht, csr->ht_cap * sizeof(HtSlot)));
  GG_TRY(ctx->dev_alloc((void **)&csr->vid, (V ? V : 1) * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->off, (V + 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->nbr, (E ? E : 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->epos, (E ? E : 1) * sizeof(uint32_t)));
  // status block and, 64 bytes behind it, the dictionary descriptor of the bucketed build: one seed copy for both
  static_assert(sizeof(BuildStatus) <= 64 && sizeof(DirectMap) <= 64, "status + dictionary descriptor: 128 bytes");
  BuildStatus *st = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&st, 128));
  BuildStatus init{0ULL, -1LL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL};
  const DirectMap dm_init{INT64_MAX, INT64_MIN, 0ULL, (unsigned long long)DICT_PACKED8, 0ULL, 0ULL, 0ULL, 0ULL};
  memset(ctx->pin_scratch, 0, 128);
  memcpy(ctx->pin_scratch, &init, sizeof(init));
  memcpy(ctx->pin_scratch + 8, &dm_init, sizeof(dm_init));
  GG_HIP(hipMemcpyAsync(st, ctx->pin_scratch, 128, hipMemcpyHostToDevice, s));
  // whole graphs and shards of up to 2^22 vertices: forward and reverse CSR by the bucketed build
  // (gg_csr_fast.hip), which also picks and fills the id dictionary; the 16-byte table is then filled only if the
  // densification needs it (ensure_ht does it later for gg_csr_lookup, source lists, the neighbour filter)
  int fast = 0;
  ctx->status_early = false;
  staging_trace_print();
  if (!ctx->legacy_build) GG_TRY(csr_build_fast(ctx, csr, st, &fast));  // (copies the staged vertex ids itself)
  if (!fast) {
    if (V) GG_HIP(hipMemcpyAsync(csr->vid, ctx->c_vid.dev, V * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    GG_LAUNCH(ctx, "ht_init", k_ht_init, dim3((unsigned)((csr->ht_cap + 255) / 256)), dim3(256), 0, csr->ht,
              csr->ht_cap);
    if (V)
      GG_LAUNCH(ctx, "ht_insert", k_ht_insert, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, csr->vid, V, csr->ht,
                csr->ht_cap, st, (uint32_t)part, (uint32_t)n_parts);
  }
  if (!fast) GG_TRY(ctx->dev_alloc((void **)&csr->row, (E ? E : 1) * sizeof(uint32_t)));

  unsigned long long *kept_dev = nullptr, *kept_rev_dev = nullptr;
  if (!fast) {
    GG_TRY(ctx->dev_alloc((void **)&kept_dev, sizeof(unsigned long long)));
    GG_HIP(hipMemsetAsync(kept_dev, 0, sizeof(unsigned long long), s));
  }
  uint32_t *rkey_sorted = nullptr;
  if (shard && !fast) {
    GG_TRY(ctx->dev_alloc((void **)&kept_rev_dev, sizeof(unsigned long long)));
    GG_HIP(hipMemsetAsync(kept_rev_dev, 0, sizeof(unsigned long long), s));
    GG_TRY(ctx->dev_alloc((void **)&csr->roff, (V + 1) * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&csr->rnbr, (E ? E : 1) * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&rkey_sorted, (E ? E : 1) * sizeof(uint32_t)));
  }
  if (E && !fast) {
    // ---- densify + fused pass-0 histogram --------------------------------------------------------
    const int key_bits = ceil_log2_u64(V < 2 ? 2 : V);
    const int passes = (key_bits + RB_MAX_BITS - 1) / RB_MAX_BITS;
    const int bits0 = (key_bits + passes - 1) / passes;
    const uint64_t nblocks64 = (E + RB_TILE - 1) / RB_TILE;
    const unsigned nblocks = (unsigned)nblocks64;
    uint32_t *su = nullptr, *dv = nullptr, *counts0 = nullptr, *rk = nullptr, *rv = nullptr, *counts0r = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&su, E * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&dv, E * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&counts0, (uint64_t)(1u << bits0) * nblocks64 * sizeof(uint32_t)));
    if (shard) {
      GG_TRY(ctx->dev_alloc((void **)&rk, E * sizeof(uint32_t)));
      GG_TRY(ctx->dev_alloc((void **)&rv, E * sizeof(uint32_t)));
      GG_TRY(ctx->dev_alloc((void **)&counts0r, (uint64_t)(1u << bits0) * nblocks64 * sizeof(uint32_t)));
    }
    uint32_t *tvf = nullptr, *tvr = nullptr;
    if (shard) {
      GG_TRY(ctx->dev_alloc((void **)&tvf, nblocks64 * sizeof(uint32_t)));
      GG_TRY(ctx->dev_alloc((void **)&tvr, nblocks64 * sizeof(uint32_t)));
      GG_LAUNCH(ctx, "densify_shard", k_densify_shard, dim3(nblocks), dim3(RB_THREADS), 0, ctx->c_src.dev,
                ctx->c_dst.dev, E, csr->ht, csr->ht_cap, st, (uint32_t)part, (uint32_t)n_parts, su, dv, rk, rv,
                (uint32_t)bits0, nblocks64, counts0, counts0r, tvf, tvr);
      // a shard only serves all-source 2-hop counting: no edge-rowid payload (csr->epos stays unset)
      RadixIO io{su, dv, nullptr, csr->row, csr->nbr, nullptr};
      GG_TRY(radix_sort_stable(ctx, io, E, false, false, key_bits, counts0, bits0, kept_dev, false, tvf));
      // reverse CSR of the edges whose DESTINATION this shard owns (rowid order inside a row)
      RadixIO ior{rk, rv, nullptr, rkey_sorted, csr->rnbr, nullptr};
      GG_TRY(radix_sort_stable(ctx, ior, E, false, false, key_bits, counts0r, bits0, kept_rev_dev, false, tvr));
    } else {
      // direct-address dictionary for dense ids (decided on the device; see DirectMap)
      DirectMap *dm = nullptr;
      uint32_t *dir = nullptr;
      GG_TRY(ctx->dev_alloc((void **)&dm, sizeof(DirectMap)));
      GG_TRY(ctx->dev_alloc((void **)&dir, DIRECT_MAX_RANGE * sizeof(uint32_t)));
      const DirectMap dm_init{INT64_MAX, INT64_MIN, 0ULL};
      memcpy(ctx->pin_scratch + 8, &dm_init, sizeof(dm_init));  // (the first words carry the BuildStatus seed)
      GG_HIP(hipMemcpyAsync(dm, ctx->pin_scratch + 8, sizeof(dm_init), hipMemcpyHostToDevice, s));
      if (V) {
        const unsigned mm_blocks = (V + 255) / 256 < 64 ? (unsigned)((V + 255) / 256) : 64u;
        GG_LAUNCH(ctx, "id_minmax", k_id_minmax, dim3(mm_blocks), dim3(256), 0, csr->vid, V, dm);
      }
      GG_LAUNCH(ctx, "direct_decide", k_dict_decide, dim3(1), dim3(64), 0, dm, V, 0u, 0u);
      GG_LAUNCH(ctx, "direct_init", k_direct_init, dim3((unsigned)(DIRECT_MAX_RANGE / 256)), dim3(256), 0, dir, dm);
      if (V)
        GG_LAUNCH(ctx, "direct_fill", k_direct_fill, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, csr->vid, V, dir,
                  dm);
      GG_LAUNCH(ctx, "densify_hist", k_densify_hist, dim3(nblocks), dim3(RB_THREADS), 0, ctx->c_src.dev,
                ctx->c_dst.dev, E, csr->ht, csr->ht_cap, st, (const uint32_t *)dir, (const DirectMap *)dm, su, dv,
                (uint32_t)bits0, nblocks64, counts0);
      // ---- stable radix scatter by source: (u, v, position) -> (row, nbr, epos) ------------------------
      const bool rowid = ctx->keep_edge_rowid;
      RadixIO io{su, dv, nullptr, csr->row, csr->nbr, rowid ? csr->epos : nullptr};
      GG_TRY(radix_sort_stable(ctx, io, E, rowid, rowid, key_bits, counts0, bits0, kept_dev, false));
    }
    ctx->dev_free(tvf);
    ctx->dev_free(tvr);
    ctx->dev_free(counts0);
    ctx->dev_free(counts0r);
    ctx->dev_free(su);
    ctx->dev_free(dv);
    ctx->dev_free(rk);
    ctx->dev_free(rv);
  }
  if (shard && !fast) {
    GG_LAUNCH(ctx, "row_offsets", k_row_offsets, dim3((unsigned)(((E ? E : 1) + 1023) / 1024)), dim3(256), 0,
              rkey_sorted, (uint64_t)0, (const unsigned long long *)kept_rev_dev, V, csr->roff, (const unsigned long long *)ctx->dev_err);
  }
  if (!fast) {
    GG_LAUNCH(ctx, "row_offsets", k_row_offsets, dim3((unsigned)(((E ? E : 1) + 1023) / 1024)), dim3(256), 0, csr->row,
              (uint64_t)0, (const unsigned long long *)kept_dev, V, csr->off, (const unsigned long long *)ctx->dev_err);
    GG_LAUNCH(ctx, "publish_kept", k_publish_kept, dim3(1), dim3(64), 0, (const uint64_t *)kept_dev,
              (const uint64_t *)kept_rev_dev, st);
  }
  // (the bucketed build left the kept count in the status block: the gather of explicit rowids reads it there)
  const unsigned long long *kept_count = fast ? (const unsigned long long *)&st->kept : kept_dev;
  csr->has_rowid = !shard && ctx->keep_edge_rowid;
  if (ctx->rowid_explicit && E && csr->has_rowid) {
    // rows staged without rowids never wrote the rowid column: their rowid is their position
    GG_TRY(grow_column(ctx, ctx->c_rowid, ctx->c_rowid.cap < E ? ctx->c_rowid.cap : (size_t)E, (size_t)E));
    for (auto &range : ctx->implicit_rowid_ranges)
      GG_LAUNCH(ctx, "rowid_iota", k_rowid_iota, dim3((unsigned)((range.second + 255) / 256)), dim3(256), 0,
                ctx->c_rowid.dev + range.first, (int64_t)range.first, range.second);
    GG_TRY(ctx->dev_alloc((void **)&csr->eid, E * sizeof(int64_t)));
    GG_LAUNCH(ctx, "gather_rowid", k_gather_rowid, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, csr->epos,
              ctx->c_rowid.dev, kept_count, csr->eid,
              (const unsigned long long *)ctx->dev_err);
  }

  // status back to the host (the only synchronisation of the build): duplicate check, sentinel
  // vertex, kept-edge count
  // (the bucketed build runs no chained scan; k_gather_rowid still does nothing while the word is set — an earlier
  // call that failed may have left it — so a build with explicit rowids reports it rather than an unwritten column)
  if (fast && ctx->status_early && !csr->eid) {
    GG_HIP(hipEventSynchronize(ctx->status_ev));  // (the status left the device behind the column scan)
  } else {
    if (!fast || csr->eid)
      GG_HIP(hipMemcpyAsync(&st->scan_error, ctx->dev_err, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, st, sizeof(BuildStatus), hipMemcpyDeviceToHost, s));
    GG_HIP(hipStreamSynchronize(s));
  }
  BuildStatus hs;
  memcpy(&hs, ctx->pin_scratch, sizeof(hs));
  ctx->dev_free(st);
  ctx->dev_free(kept_dev);
  ctx->dev_free(kept_rev_dev);
  if (!fast) csr->rrow = rkey_sorted;  // null unless this is a shard build (ensure_reverse fills it lazily otherwise)
  if (hs.scan_error) {
    GG_HIP(hipMemsetAsync(ctx->dev_err, 0, sizeof(unsigned long long), s));
    set_error("CSR build: a chained prefix scan gave up waiting for a predecessor tile");
    return GG_ERR_HIP;
  }
  if (hs.dup_vertex) {
    set_error("vertex key column is not unique (duplicate vertex id)");
    return GG_ERR_DUPLICATE_VERTEX;
  }
  csr->ht_min_idx = hs.min_idx;
  csr->ht_built = !fast || hs.dict_mode == DICT_WIDE16;
  csr->E = hs.kept;
  csr->E_rev = hs.kept_rev;
  csr->owned_vertices = shard ? hs.owned : V;
  csr->dropped = shard ? 0 : E - hs.kept;  // a shard cannot tell dropped edges from other shards' edges
  for (void *p : {(void *)csr->off, (void *)csr->nbr, (void *)csr->row, (void *)csr->epos, (void *)csr->eid,
                  (void *)csr->vid, (void *)csr->ht, (void *)csr->roff, (void *)csr->rnbr, (void *)csr->rrow})
    ctx->keep(p);
  guard.armed = false;
  *out = csr;
  return GG_OK;
}

extern "C" int gg_csr_build(gg_ctx *ctx, gg_csr **out) { return csr_build_impl(ctx, 0, 1, out); }

extern "C" int gg_csr_build_shard(gg_ctx *ctx, int part, int n_parts, gg_csr **out) {
  return csr_build_impl(ctx, part, n_parts, out);
}

namespace gg {

// The 16-byte id table of a CSR whose build did not need it (direct or packed dictionary): filled from csr->vid on
// first use.  Duplicate ids cannot occur any more (the build checked); the sentinel id's index is read back.
int ensure_ht(gg_ctx *ctx, gg_csr *csr) {
  if (csr->ht_built) return GG_OK;
  GG_HIP(hipSetDevice(ctx->device));
  BuildStatus *st = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&st, sizeof(BuildStatus)));
  BuildStatus init{0ULL, -1LL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL};
  memcpy(ctx->pin_scratch + 16, &init, sizeof(init));
  GG_HIP(hipMemcpyAsync(st, ctx->pin_scratch + 16, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  GG_LAUNCH(ctx, "ht_init", k_ht_init, dim3((unsigned)((csr->ht_cap + 255) / 256)), dim3(256), 0, csr->ht, csr->ht_cap);
  if (csr->V)
    GG_LAUNCH(ctx, "ht_insert", k_ht_insert, dim3((unsigned)((csr->V + 255) / 256)), dim3(256), 0,
              (const int64_t *)csr->vid, csr->V, csr->ht, csr->ht_cap, st, 0u, 1u);
  GG_HIP(hipMemcpyAsync(ctx->pin_scratch + 16, st, sizeof(BuildStatus), hipMemcpyDeviceToHost, ctx->stream));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  BuildStatus hs;
  memcpy(&hs, ctx->pin_scratch + 16, sizeof(hs));
  ctx->dev_free(st);
  csr->ht_min_idx = hs.min_idx;
  csr->ht_built = true;
  return GG_OK;
}

// Reverse CSR from the forward COO view (row, nbr), which is sorted by (source, rowid): a stable sort
// by destination leaves every in-neighbour list in ascending (source, rowid) order.
int ensure_reverse(gg_ctx *ctx, gg_csr *csr) {
  if (csr->roff && csr->rrow) return GG_OK;
  struct Reset {  // a failed attempt must not leave half-built members behind
    gg_csr *c;
    bool armed = true;
    ~Reset() {
      if (armed) {
        c->ctx->dev_free(c->roff);
        c->ctx->dev_free(c->rnbr);
        c->roff = c->rnbr = c->rrow = nullptr;
      }
    }
  } reset{csr};
  const uint64_t V = csr->V, E = csr->E;
  uint32_t *rkey = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&csr->roff, (V + 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->rnbr, (E ? E : 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&rkey, (E ? E : 1) * sizeof(uint32_t)));
  if (E) {
    const int key_bits = ceil_log2_u64(V < 2 ? 2 : V);
    RadixIO io{csr->nbr, csr->row, nullptr, rkey, csr->rnbr, nullptr};
    // pass 0 needs a device-side n for its histogram: reuse total slot seeded with E
    unsigned long long *tot = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&tot, sizeof(unsigned long long)));
    const unsigned long long e_host = E;  // pageable source: staged by the runtime before the call returns
    GG_HIP(hipMemcpyAsync(tot, &e_host, sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
    GG_TRY(radix_sort_stable(ctx, io, E, false, false, key_bits, nullptr, 0, tot, true));
    ctx->dev_free(tot);
  }
  GG_LAUNCH(ctx, "row_offsets", k_row_offsets, dim3((unsigned)(((E ? E : 1) + 1023) / 1024)), dim3(256), 0, rkey, E,
            (const unsigned long long *)nullptr, V, csr->roff, (const unsigned long long *)ctx->dev_err);
  GG_TRY(scan_error_fetch(ctx));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  GG_TRY(scan_error_test(ctx));
  csr->rrow = rkey;
  ctx->keep(csr->roff);
  ctx->keep(csr->rnbr);
  ctx->keep(csr->rrow);
  reset.armed = false;
  return GG_OK;
}

}  // namespace gg

// ---- vertex table from the edge endpoints (gg_vertices_from_edges) ---------------------------------
// distinct ids via an open-addressing key set (8-byte slots), compacted and then sorted ascending with
// three stable radix sorts over 22/22/20-bit chunks of the sign-flipped id, so the dense numbering does
// not depend on insertion races.
namespace gg {

struct SetStatus {
  unsigned long long count;     // distinct keys inserted (the sentinel id is counted through has_min)
  unsigned long long has_min;   // the id equal to the empty-slot sentinel occurs
  unsigned long long overflow;  // the table got fuller than `limit` or a probe sequence exceeded its bound
};

// (the three kernels of the general path stride over their input: 2E endpoints or a table grown past 2^32 slots are
// more items than one launch has threads — gg_internal.h, GG_LAUNCH)
constexpr unsigned SET_MAX_BLOCKS = 1u << 22;

__global__ __launch_bounds__(256) void k_set_init(int64_t *__restrict__ set, uint64_t cap) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += stride) set[i] = HT_EMPTY;
}

__global__ __launch_bounds__(256) void k_set_insert(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                    uint64_t E, const int64_t *__restrict__ extra, uint64_t n_extra,
                                                    int64_t *__restrict__ set, uint64_t cap, uint64_t limit,
                                                    uint32_t max_probes, SetStatus *__restrict__ st) {
  __shared__ uint32_t s_new;
  if (threadIdx.x == 0) s_new = 0;
  __syncthreads();
  const uint64_t total = 2 * E + n_extra, stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t mine = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    if (*(volatile unsigned long long *)&st->overflow != 0ULL) break;
    const int64_t key = i < E ? src[i] : (i < 2 * E ? dst[i - E] : extra[i - 2 * E]);
    if (key == HT_EMPTY) {
      st->has_min = 1ULL;  // benign race: every writer stores the same value
      continue;
    }
    uint64_t slot = ht_slot(key, cap);
    uint32_t probes = 0;
    while (true) {
      int64_t k = set[slot];
      if (k == HT_EMPTY) {
        k = (int64_t)atomicCAS((unsigned long long *)&set[slot], (unsigned long long)HT_EMPTY, (unsigned long long)key);
        if (k == HT_EMPTY) {
          mine++;
          break;
        }
      }
      if (k == key) break;
      if (++probes > max_probes) {  // only reachable while the table may still fill up: ask for a bigger one
        st->overflow = 1ULL;
        break;
      }
      slot = ht_next(slot, cap);
    }
  }
  if (mine) atomicAdd(&s_new, mine);
  __syncthreads();
  if (threadIdx.x == 0 && s_new) {
    const unsigned long long before = atomicAdd(&st->count, (unsigned long long)s_new);
    if (before + s_new > limit) st->overflow = 1ULL;
  }
}

// non-empty slots -> (low half, sign-flipped high half); slot order is arbitrary, the sort fixes it
__global__ __launch_bounds__(256) void k_set_compact(const int64_t *__restrict__ set, uint64_t cap,
                                                     unsigned long long *__restrict__ cursor,
                                                     uint32_t *__restrict__ lo, uint32_t *__restrict__ hi) {
  __shared__ uint32_t s_cnt[4];
  __shared__ unsigned long long s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < cap; base += stride) {  // (block-uniform trip count)
    const uint64_t i = base + threadIdx.x;
    const int64_t k = i < cap ? set[i] : HT_EMPTY;
    const bool live = k != HT_EMPTY;
    const uint64_t m = __ballot(live);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ULL;
    }
    __syncthreads();
    if (live) {
      uint32_t before = 0;
      for (int w = 0; w < wave; w++) before += s_cnt[w];
      const uint64_t pos = s_base + before + (uint64_t)__popcll(m & ((1ULL << lane) - 1ULL));
      lo[pos] = (uint32_t)(uint64_t)k;
      hi[pos] = (uint32_t)((uint64_t)k >> 32) ^ 0x80000000u;
    }
    __syncthreads();  // (s_cnt and s_base are rewritten by the next trip)
  }
}

// sort key of one LSD round: 22-bit chunk `round` (0,1) or the top 20 bits (2) of the sign-flipped id.
// Chunks stay below 2^22, clear of the radix kernels' 0xFFFFFFFF "dropped" marker.
__global__ __launch_bounds__(256) void k_set_chunk(const uint32_t *__restrict__ lo, const uint32_t *__restrict__ hi,
                                                   uint64_t n, int round, uint32_t *__restrict__ key) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t x = ((uint64_t)hi[i] << 32) | (uint64_t)lo[i];
  key[i] = (uint32_t)(x >> (22 * round)) & 0x3FFFFFu;
}

__global__ __launch_bounds__(256) void k_set_emit(const uint32_t *__restrict__ lo, const uint32_t *__restrict__ hi,
                                                  uint64_t n, uint32_t has_min, int64_t *__restrict__ vid) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && has_min) vid[0] = HT_EMPTY;  // INT64_MIN sorts first
  if (i < n) vid[i + has_min] = (int64_t)(((uint64_t)(hi[i] ^ 0x80000000u) << 32) | (uint64_t)lo[i]);
}

}  // namespace gg

// The general path: a table that grows by a factor of eight until it holds the ids (one host round trip per attempt),
// three LSD rounds over the ids it found.  Taken when the two-stage table of the fast path overflows or a bucket of its
// bucket sort is too large for a workgroup (ids clustered far from uniform).  start_cap: first capacity to try.
static int vertices_from_edges_general(gg_ctx *ctx, int keep_staged_vertices, uint64_t *n_vertices, uint64_t start_cap) {
  hipStream_t s = ctx->stream;
  const uint64_t E = ctx->n_edges;
  const uint64_t n_old = keep_staged_vertices ? ctx->n_vertices : 0;  // ids already in the vertex table
  if (!keep_staged_vertices) ctx->n_vertices = 0;
  ctx->fill_v = 0;
  if (E == 0) {
    if (n_vertices) *n_vertices = ctx->n_vertices;
    return GG_OK;
  }

  SetStatus *st = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&st, sizeof(SetStatus)));
  // the table starts small (graphs have far fewer vertices than edge rows) and is rebuilt larger when it
  // passes half full; at cap > 2E it cannot fill, so the probe bound is lifted and the loop ends
  uint64_t cap = 1u << 16;
  while (cap < E / 8 + 2 * n_old || cap < start_cap) cap <<= 1;
  int64_t *set = nullptr;
  SetStatus host{};
  while (true) {
    const bool cannot_fill = cap > 2 * E + n_old + 1;
    GG_TRY(ctx->dev_alloc((void **)&set, cap * sizeof(int64_t)));
    GG_HIP(hipMemsetAsync(st, 0, sizeof(SetStatus), s));
    GG_LAUNCH(ctx, "set_init", k_set_init, dim3((unsigned)std::min<uint64_t>((cap + 255) / 256, SET_MAX_BLOCKS)), dim3(256), 0,
              set, cap);
    GG_LAUNCH(ctx, "set_insert", k_set_insert, dim3((unsigned)std::min<uint64_t>((2 * E + n_old + 255) / 256, SET_MAX_BLOCKS)),
              dim3(256), 0,
              ctx->c_src.dev, ctx->c_dst.dev, E, (const int64_t *)ctx->c_vid.dev, n_old, set, cap,
              cannot_fill ? ~0ULL : cap / 2, cannot_fill ? 0xFFFFFFFFu : 4096u, st);
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, st, sizeof(SetStatus), hipMemcpyDeviceToHost, s));
    GG_HIP(hipStreamSynchronize(s));
    memcpy(&host, ctx->pin_scratch, sizeof(SetStatus));
    if (!host.overflow) break;
    ctx->dev_free(set);
    cap <<= 3;
  }
  const uint64_t n = host.count, has_min = host.has_min ? 1 : 0, V = n + has_min;
  if (V >= (uint64_t)INVALID_U32) {
    set_error("more than 2^32-2 distinct endpoint ids are not supported");
    return GG_ERR_TOO_LARGE;
  }
  GG_TRY(grow_column(ctx, ctx->c_vid, 0, V));
  if (n) {
    uint32_t *lo[2] = {nullptr, nullptr}, *hi[2] = {nullptr, nullptr}, *key = nullptr, *key_sorted = nullptr;
    unsigned long long *cursor = nullptr, *tot = nullptr;
    for (uint32_t **p : {&lo[0], &lo[1], &hi[0], &hi[1], &key, &key_sorted})
      GG_TRY(ctx->dev_alloc((void **)p, n * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&cursor, sizeof(unsigned long long)));
    GG_TRY(ctx->dev_alloc((void **)&tot, sizeof(unsigned long long)));
    GG_HIP(hipMemsetAsync(cursor, 0, sizeof(unsigned long long), s));
    GG_LAUNCH(ctx, "set_compact", k_set_compact, dim3((unsigned)std::min<uint64_t>((cap + 255) / 256, SET_MAX_BLOCKS)),
              dim3(256), 0, set, cap, cursor, lo[0], hi[0]);
    // three stable LSD rounds over (22, 22, 20)-bit chunks, the two id halves riding along as payloads
    int cur = 0;
    for (int round = 0; round < 3; round++) {
      GG_LAUNCH(ctx, "set_chunk", k_set_chunk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lo[cur], hi[cur], n,
                round, key);
      // pass 0 of a sort reads its element count from the device: `cursor` holds exactly n
      GG_HIP(hipMemcpyAsync(tot, cursor, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
      RadixIO io{key, lo[cur], hi[cur], key_sorted, lo[cur ^ 1], hi[cur ^ 1]};
      GG_TRY(radix_sort_stable(ctx, io, n, true, false, round == 2 ? 20 : 22, nullptr, 0, tot, true));
      cur ^= 1;
    }
    GG_LAUNCH(ctx, "set_emit", k_set_emit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lo[cur], hi[cur], n,
              (uint32_t)has_min, ctx->c_vid.dev);
  } else {
    const int64_t only = HT_EMPTY;  // has_min with nothing else
    memcpy(ctx->pin_scratch, &only, sizeof(only));
    GG_HIP(hipMemcpyAsync(ctx->c_vid.dev, ctx->pin_scratch, sizeof(only), hipMemcpyHostToDevice, s));
  }
  GG_TRY(scan_error_fetch(ctx));
  GG_HIP(hipStreamSynchronize(s));
  GG_TRY(scan_error_test(ctx));
  ctx->n_vertices = V;
  if (n_vertices) *n_vertices = V;
  return GG_OK;
}

namespace gg {

// ---- the fast path of gg_vertices_from_edges ------------------------------------------------------------------------
// One pass over the edge rows into a table sized for the graph (a vertex occurs ~2E/V times: E/64 slots hold the ids
// of every LDBC scale at load <= 0.45; 8 MB at SF100 instead of the 64 MB a table sized from E/8 takes — the insert is a
// random 8-byte probe per endpoint, and what a probe costs is decided by how far the table exceeds an XCD's L2), a
// second table eight times larger that is only touched if the first overflows (the launch is made either way and
// returns at once otherwise: no host round trip in between), then the ids are numbered by ONE bucket sort: buckets by
// linear interpolation between the smallest and the largest id (monotone, so bucket order is id order), each bucket
// ranked by counting inside a workgroup's LDS.  One synchronisation in all (count, overflow, largest bucket).
struct SetStatus2 {
  unsigned long long count[2];     // distinct ids inserted into table 0 / table 1 (atomics: a cache line of their own)
  unsigned long long pad[14];
  unsigned long long has_min;      // the id equal to the empty-slot sentinel occurs
  unsigned long long overflow[2];  // table 0 / 1 got fuller than its limit or a probe sequence exceeded its bound
  unsigned long long n;            // ids compacted (from whichever table holds them)
  unsigned long long id_min_u, id_max_u;  // over the compacted ids, sign bit flipped (unsigned order = signed order)
  unsigned long long max_bucket;   // largest bucket of the bucket sort
};

// The table is probed two slots at a time: a PAIR of adjacent 8-byte slots is one aligned 16-byte load (what a probe
// costs is the request, not the bytes), a key's home is a pair, and linear probing moves from pair to pair — at load
// 0.4 a lookup takes ~1.05 requests instead of the ~1.3 of single slots.
typedef long long set_ll2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint64_t set_home_pair(int64_t key, uint64_t pairs) {
  return __umul64hi((uint64_t)key * DIG_GOLD, pairs);
}

// finish the probe chain of `key` that started at pair `pair` with the values `v` already loaded from it
__device__ __forceinline__ void set_insert_chain(int64_t key, uint64_t pair, set_ll2 v, int64_t *__restrict__ set,
                                                 uint64_t pairs, uint32_t max_probes, unsigned long long *overflow,
                                                 uint32_t *inserted) {
  uint32_t probes = 0;
  while (true) {
    int64_t k0 = v.x, k1 = v.y;
    if (k0 == key || k1 == key) return;
    if (k0 == HT_EMPTY) {
      k0 = (int64_t)atomicCAS((unsigned long long *)&set[2 * pair], (unsigned long long)HT_EMPTY, (unsigned long long)key);
      if (k0 == HT_EMPTY) {
        *inserted += 1;
        return;
      }
      if (k0 == key) return;
      k1 = set[2 * pair + 1];  // (someone else took slot 0 meanwhile: slot 1 may have changed too)
      if (k1 == key) return;
    }
    if (k1 == HT_EMPTY) {
      k1 = (int64_t)atomicCAS((unsigned long long *)&set[2 * pair + 1], (unsigned long long)HT_EMPTY,
                              (unsigned long long)key);
      if (k1 == HT_EMPTY) {
        *inserted += 1;
        return;
      }
      if (k1 == key) return;
    }
    if (++probes > max_probes) {
      *overflow = 1ULL;
      return;
    }
    // (a table that has overflowed keeps filling under the threads already in flight: their probe runs grow towards
    //  the bound, so they look at the flag now and then and give up with the others)
    if ((probes & 15u) == 0 && *(volatile unsigned long long *)overflow != 0ULL) return;
    pair = pair + 1 == pairs ? 0 : pair + 1;
    v = *reinterpret_cast<const set_ll2 *>(set + 2 * pair);
  }
}

// stage 0 or 1: every edge row's two endpoints and the `n_extra` ids already staged as vertices.  A fixed grid walks the
// rows in tiles of ROWS x 256: a thread loads the ids of ROWS rows, issues their 2 x ROWS first probes back to back
// (the kernel is bound by the latency of random 16-byte loads: what counts is how many are in flight) and then
// finishes the chains.  Stage 1 runs only if stage 0 overflowed.
// `rows_limit` < E + n_extra: the WARM-UP launch over the first rows with a small grid and one row per thread.  An empty
// table under the full grid is a storm: the ~4 M probes in flight all find their slot empty and compare-and-swap it —
// ten (SF100) to a hundred (SF10) device-scope atomics per slot, serialised, and each XCD's L2 keeps the empty lines
// it fetched meanwhile.  A quarter of a million probes at a time over the first rows fill the table almost without
// contention (a vertex of degree d is in the first s of E rows with probability 1 - (1 - s/E)^2d), and the full grid
// then probes a table whose lines are what it reads.
constexpr int SET_ROWS = 4;
template <int ROWS>
__global__ __launch_bounds__(256) void k_set_insert2(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                     uint64_t E, const int64_t *__restrict__ extra, uint64_t n_extra,
                                                     uint64_t rows_limit, int64_t *__restrict__ set, uint64_t pairs,
                                                     uint64_t limit, uint32_t max_probes, int stage,
                                                     SetStatus2 *__restrict__ st) {
  __shared__ uint32_t s_new;
  if (stage == 1 && st->overflow[0] == 0ULL) return;  // (set, if at all, by an earlier kernel: uniform over the grid)
  if (threadIdx.x == 0) s_new = 0;
  __syncthreads();
  const uint64_t rows = E + n_extra < rows_limit ? E + n_extra : rows_limit, tile_rows = (uint64_t)ROWS * 256;
  uint32_t inserted = 0, trip = 0;
  bool saw_min = false;
  for (uint64_t t0 = (uint64_t)blockIdx.x * tile_rows; t0 < rows; t0 += (uint64_t)gridDim.x * tile_rows, trip++) {
    // (the flag is ONE address read past every cache: polled by every wavefront on every trip it was most of the
    //  kernel's time.  Every 64th trip will do: an overflowing table is the rare case, and the chains bound themselves)
    if ((trip & 63u) == 63u && *(volatile unsigned long long *)&st->overflow[stage] != 0ULL) break;
    int64_t key[2 * ROWS];
    set_ll2 got[2 * ROWS];
    uint64_t pair[2 * ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      const uint64_t i = t0 + (uint64_t)j * 256 + threadIdx.x;
      int64_t ka = HT_EMPTY, kb = HT_EMPTY;  // (HT_EMPTY: nothing to insert)
      if (i >= rows) {
      } else if (i < E) {
        ka = src[i];
        kb = dst[i];
        saw_min = saw_min || ka == HT_EMPTY || kb == HT_EMPTY;
        if (kb == ka) kb = HT_EMPTY;
      } else {
        ka = extra[i - E];
        saw_min = saw_min || ka == HT_EMPTY;
      }
      key[2 * j] = ka;
      key[2 * j + 1] = kb;
    }
#pragma unroll
    for (int q = 0; q < 2 * ROWS; q++) {
      pair[q] = set_home_pair(key[q], pairs);
      if (key[q] != HT_EMPTY) got[q] = *reinterpret_cast<const set_ll2 *>(set + 2 * pair[q]);
    }
#pragma unroll
    for (int q = 0; q < 2 * ROWS; q++)
      if (key[q] != HT_EMPTY && got[q].x != key[q] && got[q].y != key[q])
        set_insert_chain(key[q], pair[q], got[q], set, pairs, max_probes, &st->overflow[stage], &inserted);
    if (ROWS > 1) {
      // the full grid: new ids are counted per wavefront and tile (behind the warm-up hardly any wavefront has one), so
      // that a table about to pass its limit is noticed while it still has room
      uint32_t wave_new = inserted;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wave_new += __shfl_xor(wave_new, o, 64);
      if ((threadIdx.x & 63) == 0 && wave_new) {
        const unsigned long long before = atomicAdd(&st->count[stage], (unsigned long long)wave_new);
        if (before + wave_new > limit) st->overflow[stage] = 1ULL;
      }
      inserted = 0;
    }
  }
  if (saw_min) st->has_min = 1ULL;  // benign race: every writer stores the same value
  if (ROWS == 1) {
    // the warm-up: nearly every wavefront inserts on nearly every trip, and one counter cannot take an atomic from
    // each of them (0.4 ms for 2.5 M rows); it is over quickly and bounded (2 x the table's slots in rows), so its
    // workgroups count in registers and report once
    uint32_t wave_new = inserted;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_new += __shfl_xor(wave_new, o, 64);
    if ((threadIdx.x & 63) == 0 && wave_new) atomicAdd(&s_new, wave_new);
    __syncthreads();
    if (threadIdx.x == 0 && s_new) {
      const unsigned long long before = atomicAdd(&st->count[stage], (unsigned long long)s_new);
      if (before + s_new > limit) st->overflow[stage] = 1ULL;
    }
  }
}

__global__ __launch_bounds__(256) void k_set_init2(int64_t *__restrict__ set0, uint64_t cap0, SetStatus2 *__restrict__ st) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap0) set0[i] = HT_EMPTY;
  if (i == 0) {
    st->count[0] = st->count[1] = st->has_min = st->overflow[0] = st->overflow[1] = st->n = st->max_bucket = 0ULL;
    st->id_min_u = ~0ull;
    st->id_max_u = 0ull;
  }
}

// the second table is cleared only if it is going to be used
__global__ __launch_bounds__(256) void k_set_init_stage1(int64_t *__restrict__ set1, uint64_t cap1,
                                                         const SetStatus2 *__restrict__ st) {
  if (st->overflow[0] == 0ULL) return;  // (written, if at all, by an earlier kernel: a plain load)
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap1; i += (uint64_t)gridDim.x * blockDim.x)
    set1[i] = HT_EMPTY;
}

// non-empty slots of the table that holds the ids -> keys[] (slot order: arbitrary, the sort fixes it), their number,
// smallest and largest.  Launched once per table; only the launch over the table in use does anything.  A workgroup
// owns a contiguous range of slots: it counts the live ones, reserves its piece of keys[] with ONE atomic (an atomic
// per 256 slots on one counter was most of an earlier form's 160 us) and reads the range again (L2-hot) to write.
__global__ __launch_bounds__(256) void k_set_compact2(const int64_t *__restrict__ set, uint64_t cap, int stage,
                                                      SetStatus2 *__restrict__ st, int64_t *__restrict__ keys,
                                                      uint64_t keys_cap) {
  __shared__ uint32_t s_cnt[4];
  __shared__ unsigned long long s_base, s_lo[4], s_hi[4];
  // (the flags were written by earlier kernels: plain loads — read past the caches, one address under 4 096 workgroups,
  //  they were most of this kernel's 280 us)
  const bool second = st->overflow[0] != 0ULL;
  if ((int)second != stage) return;
  if (second && st->overflow[1] != 0ULL) return;  // nothing usable: the host retries
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t per = ((cap + gridDim.x - 1) / gridDim.x + 255) & ~255ull;  // slots per workgroup, whole trips
  const uint64_t r_lo = per * blockIdx.x, r_hi = r_lo + per < cap ? r_lo + per : cap;
  // smallest / largest id: sign-flipped to unsigned order (native 64-bit atomics; the signed forms are CAS loops, and
  // ten thousand waves on one address took 3 ms in them), reduced over the workgroup first
  unsigned long long lo = ~0ull, hi = 0ull;
  uint32_t mine = 0;
  for (uint64_t i = r_lo + threadIdx.x; i < r_hi; i += 256) {
    const int64_t k = set[i];
    if (k != HT_EMPTY) {
      const unsigned long long u = (unsigned long long)k ^ 0x8000000000000000ull;
      lo = u < lo ? u : lo;
      hi = u > hi ? u : hi;
      mine++;
    }
  }
  uint32_t wsum = mine;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long lo2 = __shfl_xor(lo, o, 64), hi2 = __shfl_xor(hi, o, 64);
    lo = lo2 < lo ? lo2 : lo;
    hi = hi2 > hi ? hi2 : hi;
    wsum += __shfl_xor(wsum, o, 64);
  }
  if (lane == 0) {
    s_lo[wave] = lo;
    s_hi[wave] = hi;
    s_cnt[wave] = wsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    s_base = tot ? atomicAdd(&st->n, (unsigned long long)tot) : 0ULL;
    unsigned long long l = s_lo[0], h = s_hi[0];
    for (int w = 1; w < 4; w++) {
      l = s_lo[w] < l ? s_lo[w] : l;
      h = s_hi[w] > h ? s_hi[w] : h;
    }
    if (tot) {
      atomicMin(&st->id_min_u, l);
      atomicMax(&st->id_max_u, h);
    }
  }
  __syncthreads();
  uint64_t pos = s_base;
  for (uint64_t i0 = r_lo; i0 < r_hi; i0 += 256) {  // (uniform trip count)
    const uint64_t i = i0 + threadIdx.x;
    const int64_t k = i < r_hi ? set[i] : HT_EMPTY;
    const bool live = k != HT_EMPTY;
    const uint64_t m = __ballot(live);
    __syncthreads();  // (the trip before is done with s_cnt)
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < 4; w++) {
      if (w < wave) before += s_cnt[w];
      all += s_cnt[w];
    }
    if (live) {
      const uint64_t p = pos + before + (uint64_t)__popcll(m & ((1ULL << lane) - 1ULL));
      if (p < keys_cap) keys[p] = k;
    }
    pos += all;
  }
}

// bucket of an id: floor((id - min) * B / (span + 1)) in double arithmetic — conversions, the product with a positive
// constant and the floor are all monotone, so a larger id never lands in a smaller bucket; clamped to B - 1
__device__ __forceinline__ uint32_t id_bucket(int64_t id, unsigned long long id_min_u, double scale, uint32_t B) {
  const double x = (double)(((unsigned long long)id ^ 0x8000000000000000ull) - id_min_u) * scale;
  const unsigned long long b = (unsigned long long)x;
  return b >= B ? B - 1 : (uint32_t)b;
}

// SET_WG workgroups, each over a contiguous slice of the ids: bucket sizes in LDS, then ONE returning atomic per
// (workgroup, bucket) reserves the workgroup's range inside the bucket (base[wg][b]) — 448 k atomics onto 1 200
// addresses took 90 us, 64 x 1 200 take none to speak of
constexpr uint32_t SET_WG = 64, SET_MAX_B = 8192;
__global__ __launch_bounds__(1024) void k_set_bucket_hist(const int64_t *__restrict__ keys, const SetStatus2 *__restrict__ st,
                                                          uint32_t B, uint32_t *__restrict__ hist,
                                                          uint32_t *__restrict__ base) {
  __shared__ uint32_t s_cnt[SET_MAX_B];
  const uint64_t n = st->n, per = (n + SET_WG - 1) / SET_WG;
  const uint64_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  const unsigned long long id_min_u = st->id_min_u;
  const double scale = (double)B / ((double)(st->id_max_u - id_min_u) + 1.0);
  for (uint32_t b = threadIdx.x; b < B; b += 1024) s_cnt[b] = 0;
  __syncthreads();
  for (uint64_t i = lo + threadIdx.x; i < hi; i += 1024) atomicAdd(&s_cnt[id_bucket(keys[i], id_min_u, scale, B)], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < B; b += 1024) {
    const uint32_t c = s_cnt[b];
    base[(size_t)blockIdx.x * B + b] = c ? atomicAdd(&hist[b], c) : 0u;
  }
}

// one workgroup: exclusive scan of the B bucket sizes into start[0..B], largest bucket -> status
__global__ __launch_bounds__(1024) void k_set_bucket_scan(const uint32_t *__restrict__ hist, uint32_t B,
                                                          uint32_t *__restrict__ start, SetStatus2 *__restrict__ st) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry, s_max;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0, s_max = 0;
  __syncthreads();
  uint32_t mx = 0;
  for (uint32_t b0 = 0; b0 < B; b0 += 1024) {
    const uint32_t i = b0 + threadIdx.x;
    const uint32_t v = i < B ? hist[i] : 0;
    mx = v > mx ? v : mx;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = s_carry;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    if (i < B) start[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = before + incl;
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
  }
  if (lane == 0) atomicMax(&s_max, mx);
  __syncthreads();
  if (threadIdx.x == 0) {
    start[B] = s_carry;
    st->max_bucket = s_max;
  }
}

// the same slices: an id goes to start[b] + base[wg][b] + its rank among the workgroup's ids of that bucket (an LDS
// counter; order inside a bucket is whatever — it is sorted next)
__global__ __launch_bounds__(1024) void k_set_bucket_scatter(const int64_t *__restrict__ keys,
                                                             const SetStatus2 *__restrict__ st, uint32_t B,
                                                             const uint32_t *__restrict__ start,
                                                             const uint32_t *__restrict__ base, int64_t *__restrict__ out) {
  __shared__ uint32_t s_cnt[SET_MAX_B];
  const uint64_t n = st->n, per = (n + SET_WG - 1) / SET_WG;
  const uint64_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  const unsigned long long id_min_u = st->id_min_u;
  const double scale = (double)B / ((double)(st->id_max_u - id_min_u) + 1.0);
  for (uint32_t b = threadIdx.x; b < B; b += 1024) s_cnt[b] = 0;
  __syncthreads();
  for (uint64_t i = lo + threadIdx.x; i < hi; i += 1024) {
    const int64_t k = keys[i];
    const uint32_t b = id_bucket(k, id_min_u, scale, B);
    out[(uint64_t)start[b] + base[(size_t)blockIdx.x * B + b] + atomicAdd(&s_cnt[b], 1u)] = k;
  }
}

// one workgroup per bucket: the ids are distinct, so the rank of an id is the number of smaller ids in its bucket
constexpr uint32_t SET_BUCKET_CAP = 4096;
__global__ __launch_bounds__(256) void k_set_bucket_sort(const int64_t *__restrict__ in, const uint32_t *__restrict__ start,
                                                         uint32_t has_min, int64_t *__restrict__ vid) {
  __shared__ int64_t s_key[SET_BUCKET_CAP];
  const uint32_t lo = start[blockIdx.x], m = start[blockIdx.x + 1] - lo;
  if (blockIdx.x == 0 && threadIdx.x == 0 && has_min) vid[0] = HT_EMPTY;  // INT64_MIN sorts first
  for (uint32_t i = threadIdx.x; i < m; i += 256) s_key[i] = in[lo + i];
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < m; i += 256) {
    const int64_t k = s_key[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < m; j++) rank += s_key[j] < k;  // (LDS broadcast reads)
    vid[(uint64_t)has_min + lo + rank] = k;
  }
}

}  // namespace gg

extern "C" int gg_vertices_from_edges(gg_ctx *ctx, int keep_staged_vertices, uint64_t *n_vertices) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  if (n_vertices) *n_vertices = 0;
  GG_TRY(gg_staging_sync(ctx));
  std::lock_guard<std::mutex> lk(ctx->mu);
  ApiScope scope(ctx);
  GG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const uint64_t E = ctx->n_edges;
  const uint64_t n_old = keep_staged_vertices ? ctx->n_vertices : 0;  // ids already in the vertex table
  if (E == 0 || ctx->legacy_build) return vertices_from_edges_general(ctx, keep_staged_vertices, n_vertices, 0);

  // two tables: E/48 slots (at least 2^16) and eight times that; they may fill to 7/8 and 3/4 (probes look at a PAIR
  // of slots per request: ~1.2 requests per lookup at that load).  What a probe costs is decided by how far the table
  // exceeds an XCD's 4 MB of L2 (profiles/r03_ubench_gather_sizes.txt: 546 / 650 / 785 us for 80 M probes of a 4 / 6.4 /
  // 8 MB table), so the first table is cut for graphs like LDBC's knows at the large scales (64+ rows per person: 6.6 MB
  // at SF100, load 0.54) and sparser ones pay for the second attempt — made on the device, without the host.
  uint64_t cap0 = 1u << 16;
  if (cap0 < E / 48 + 2 * n_old) cap0 = (E / 48 + 2 * n_old + 1) & ~1ull;  // (slots come in pairs)
  const uint64_t cap1 = 8 * cap0;
  const uint64_t lim0 = cap0 / 8 * 7, lim1 = cap1 / 4 * 3;  // (LDBC SF10 has 58 rows per person: load 0.82 of the first table)
  const uint64_t keys_cap = lim1 + 2048;
  const uint32_t B = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cap0 / 512, 64), SET_MAX_B);  // ~<= 320 ids per bucket
  SetStatus2 *st = nullptr;
  int64_t *set0 = nullptr, *set1 = nullptr, *keys = nullptr, *sorted_in = nullptr;
  uint32_t *hist = nullptr, *start = nullptr, *base = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&st, sizeof(SetStatus2)));
  GG_TRY(ctx->dev_alloc((void **)&set0, cap0 * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&set1, cap1 * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&keys, keys_cap * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&hist, (size_t)B * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&start, ((size_t)B + 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&base, (size_t)SET_WG * B * sizeof(uint32_t)));
  GG_HIP(hipMemsetAsync(hist, 0, (size_t)B * sizeof(uint32_t), s));
  const unsigned rows_grid = (unsigned)std::min<uint64_t>((E + n_old + SET_ROWS * 256 - 1) / (SET_ROWS * 256), 256 * 16);
  // (probe runs are bounded at 64: at load <= 0.625 a longer one means the hash clusters these ids — the next stage,
  //  or the general path with its unbounded probes, takes over)
  GG_LAUNCH(ctx, "set_init", k_set_init2, dim3((unsigned)((cap0 + 255) / 256)), dim3(256), 0, set0, cap0, st);
  const uint64_t warm_rows = std::min<uint64_t>(E, cap0 + cap0 / 2);
  GG_LAUNCH(ctx, "set_warm", k_set_insert2<1>, dim3(1024), dim3(256), 0, (const int64_t *)ctx->c_src.dev,
            (const int64_t *)ctx->c_dst.dev, E, (const int64_t *)ctx->c_vid.dev, n_old, warm_rows, set0, cap0 / 2, lim0, 64u, 0, st);
  GG_LAUNCH(ctx, "set_insert", k_set_insert2<SET_ROWS>, dim3(rows_grid), dim3(256), 0, (const int64_t *)ctx->c_src.dev,
            (const int64_t *)ctx->c_dst.dev, E, (const int64_t *)ctx->c_vid.dev, n_old, ~0ull, set0, cap0 / 2, lim0, 64u, 0, st);
  GG_LAUNCH(ctx, "set_init", k_set_init_stage1, dim3(2048), dim3(256), 0, set1, cap1, (const SetStatus2 *)st);
  GG_LAUNCH(ctx, "set_warm", k_set_insert2<1>, dim3(1024), dim3(256), 0, (const int64_t *)ctx->c_src.dev,
            (const int64_t *)ctx->c_dst.dev, E, (const int64_t *)ctx->c_vid.dev, n_old, cap1 + cap1 / 2, set1, cap1 / 2, lim1, 64u, 1, st);
  GG_LAUNCH(ctx, "set_insert", k_set_insert2<SET_ROWS>, dim3(rows_grid), dim3(256), 0, (const int64_t *)ctx->c_src.dev,
            (const int64_t *)ctx->c_dst.dev, E, (const int64_t *)ctx->c_vid.dev, n_old, ~0ull, set1, cap1 / 2, lim1, 64u, 1, st);
  GG_LAUNCH(ctx, "set_compact", k_set_compact2, dim3((unsigned)std::min<uint64_t>((cap0 + 255) / 256, 1024)), dim3(256), 0,
            (const int64_t *)set0, cap0, 0, st, keys, keys_cap);
  GG_LAUNCH(ctx, "set_compact", k_set_compact2, dim3((unsigned)std::min<uint64_t>((cap1 + 255) / 256, 1024)), dim3(256), 0,
            (const int64_t *)set1, cap1, 1, st, keys, keys_cap);
  GG_LAUNCH(ctx, "set_bucket_hist", k_set_bucket_hist, dim3(SET_WG), dim3(1024), 0, (const int64_t *)keys,
            (const SetStatus2 *)st, B, hist, base);
  GG_LAUNCH(ctx, "set_bucket_scan", k_set_bucket_scan, dim3(1), dim3(1024), 0, (const uint32_t *)hist, B, start, st);
  GG_HIP(hipMemcpyAsync(ctx->pin_scratch, st, sizeof(SetStatus2), hipMemcpyDeviceToHost, s));
  GG_HIP(hipStreamSynchronize(s));
  SetStatus2 host;
  memcpy(&host, ctx->pin_scratch, sizeof(SetStatus2));
  const bool second = host.overflow[0] != 0;
  if ((second && host.overflow[1]) || host.max_bucket > SET_BUCKET_CAP) {
    // more ids than both tables hold, or ids too clustered for the bucket sort: the general path
    ctx->dev_free(set0);
    ctx->dev_free(set1);
    ctx->dev_free(keys);
    return vertices_from_edges_general(ctx, keep_staged_vertices, n_vertices, second && host.overflow[1] ? 8 * cap1 : 0);
  }
  const uint64_t n = host.n, has_min = host.has_min ? 1 : 0, V = n + has_min;
  if (!keep_staged_vertices) ctx->n_vertices = 0;
  ctx->fill_v = 0;
  if (V >= (uint64_t)INVALID_U32) {
    set_error("more than 2^32-2 distinct endpoint ids are not supported");
    return GG_ERR_TOO_LARGE;
  }
  GG_TRY(grow_column(ctx, ctx->c_vid, 0, V));
  if (n) {
    GG_TRY(ctx->dev_alloc((void **)&sorted_in, n * sizeof(int64_t)));
    GG_LAUNCH(ctx, "set_bucket_scatter", k_set_bucket_scatter, dim3(SET_WG), dim3(1024), 0, (const int64_t *)keys,
              (const SetStatus2 *)st, B, (const uint32_t *)start, (const uint32_t *)base, sorted_in);
    GG_LAUNCH(ctx, "set_bucket_sort", k_set_bucket_sort, dim3(B), dim3(256), 0, (const int64_t *)sorted_in,
              (const uint32_t *)start, (uint32_t)has_min, (int64_t *)ctx->c_vid.dev);
  } else if (has_min) {
    const int64_t only = HT_EMPTY;  // has_min with nothing else
    memcpy(ctx->pin_scratch, &only, sizeof(only));
    GG_HIP(hipMemcpyAsync(ctx->c_vid.dev, ctx->pin_scratch, sizeof(only), hipMemcpyHostToDevice, s));
    GG_HIP(hipStreamSynchronize(s));  // (pin_scratch is reused by the next call)
  }
  // no synchronisation: the build that follows is queued on the same stream behind these kernels
  ctx->n_vertices = V;
  if (n_vertices) *n_vertices = V;
  return GG_OK;
}

extern "C" int gg_csr_info(const gg_csr *csr, uint64_t *n_vertices, uint64_t *n_edges_kept,
                           uint64_t *n_edges_dropped) {
  if (!csr) return GG_ERR_INVALID_ARG;
  if (n_vertices) *n_vertices = csr->V;
  if (n_edges_kept) *n_edges_kept = csr->E;
  if (n_edges_dropped) *n_edges_dropped = csr->dropped;
  return GG_OK;
}

extern "C" int gg_csr_export(const gg_csr *csr, int64_t *off, int64_t *nbr, int64_t *eid, int64_t *vid) {
  if (!csr) return GG_ERR_INVALID_ARG;
  gg_ctx *ctx = csr->ctx;
  GG_HIP(hipSetDevice(ctx->device));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  if (off) {
    std::vector<uint32_t> h(csr->V + 1);
    GG_HIP(hipMemcpy(h.data(), csr->off, (csr->V + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i <= csr->V; i++) off[i] = (int64_t)h[i];
  }
  if (nbr && csr->E) {
    std::vector<uint32_t> h(csr->E);
    GG_HIP(hipMemcpy(h.data(), csr->nbr, csr->E * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < csr->E; i++) nbr[i] = (int64_t)h[i];
  }
  if (eid && csr->E) {
    if (!csr->has_rowid) {  // shards and rowid-free builds carry no edge rowids
      for (uint64_t i = 0; i < csr->E; i++) eid[i] = -1;
    } else if (csr->eid) {
      GG_HIP(hipMemcpy(eid, csr->eid, csr->E * sizeof(int64_t), hipMemcpyDeviceToHost));
    } else {  // implicit rowids: the edge's append position
      std::vector<uint32_t> h(csr->E);
      GG_HIP(hipMemcpy(h.data(), csr->epos, csr->E * sizeof(uint32_t), hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < csr->E; i++) eid[i] = (int64_t)h[i];
    }
  }
  if (vid && csr->V) GG_HIP(hipMemcpy(vid, csr->vid, csr->V * sizeof(int64_t), hipMemcpyDeviceToHost));
  return GG_OK;
}
