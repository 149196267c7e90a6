// gg_csr_fast.hip — bucketed two-level CSR build: forward and reverse CSR in one pass over the edge table.
//
// Replaces the same reference work as gg_csr.hip (the hash-join build side, JoinHashTable::Build/Finalize/
// InsertHashes, src/execution/join_hashtable.cpp:150-302) for graphs of up to 2^22 vertices.  The legacy
// LSD build sorts (u, v) three times by 7-bit digits per direction (six scatter passes, five histogram
// passes, six scans, two row-offset passes at SF100).  Here the dense source index u is split into a HIGH
// part (<= 10 bits: the bucket) and a LOW part (<= 12 bits: the vertex inside the bucket):
//
//   D  k_densify_pairs    one read of the (src, dst) id columns; two dictionary probes per edge row
//                         (gg_dict.h: direct array / packed 8-byte slots / 16-byte slots); writes the dense
//                         pair (u, v) in rowid order; counts, per tile, the bucket of u (forward) and the
//                         bucket of v (reverse) in LDS
//      k_col_*            three small kernels: the (tile x bucket) counters of both directions -> global positions
//   A  k_partition_dual   one read of the pairs, TWO stable bucket partitions written from one LDS staging
//                         area: forward (low(u), v[, edge position]) by bucket(u), reverse (low(v), u) by
//                         bucket(v); runs of one bucket go out as contiguous stores
//   B  k_sub_sort         a workgroup per 8192-entry chunk of a bucket: stable sort by sub-bucket, in place
//      k_sub_totals       where each (bucket, sub-bucket) leaf starts
//      k_leaf_rows        a wave per leaf (<= 64 vertices): gathers the leaf's run of every chunk, writes the row
//                         offsets (no separate row-offset pass) and the rows
//
// Stable everywhere: inside a forward row neighbours keep ascending edge-rowid order (bit-identical to the
// legacy build and to the oracle's counting sort); inside a reverse row in-neighbours are in rowid order too
// (the order shard builds already produce).  Ranks inside a wave come from LDS cursors private to that wave:
// either straight from ds_add_rtn, whose lane order is checked per context (k_lds_order_probe), or from
// __ballot match masks; waves, tiles and chunks are ordered by prefix sums.
// Algorithmic bytes: SURVEY.md §8d, 32E + 8V (densification) + 32E + 16V (CSR without rowid).
#include "gg_dict.h"
#include "gg_internal.h"
#include "gg_runs.h"

using namespace gg;

namespace gg {

constexpr int FB_THREADS = 512;
constexpr int FB_WAVES = FB_THREADS / 64;
#ifndef GG_FB_ITEMS
#define GG_FB_ITEMS 16
#endif
constexpr int FB_ITEMS = GG_FB_ITEMS;            // edge rows per lane and tile
constexpr int FB_TILE = FB_THREADS * FB_ITEMS;   // 8192 rows per workgroup (D and A share the tiling)
constexpr int FB_WTILE = FB_TILE / FB_WAVES;     // contiguous rows per wave in A
constexpr int FB_MAX_HB = 10;                    // bucket bits
#ifndef GG_FB_LOAD_PCT
#define GG_FB_LOAD_PCT 50                        // load factor of the packed dictionary, percent
#endif
#ifndef GG_FB_SUBPIPE
#define GG_FB_SUBPIPE 2  // workgroups per CU of the pipelined k_sub_sort_pipe: its resident set (0: k_sub_sort, one chunk per workgroup)
#endif

template <typename T>
__device__ __forceinline__ T ld_stream(const T *p) {
  return __builtin_nontemporal_load(p);
}
template <typename T>
__device__ __forceinline__ void st_stream(T *p, T v) {
  __builtin_nontemporal_store(v, p);
}

// Match mask: which lanes of the wave hold the same small value d as this lane.  Every lane ORs its bit
// into the wave's private 64-bit LDS word of d and reads the word back (LDS executes a wave's instructions in
// order, and OR commutes, so the mask is exact whatever order the lanes of one instruction are served in);
// the lowest lane of each group clears the word again.  ~10 instructions instead of ~7 per key bit with
// __ballot masks.
__device__ __forceinline__ uint64_t match_or(unsigned long long *mm, uint32_t d, bool valid, int lane) {
  if (valid) atomicOr(&mm[d], 1ULL << lane);
  __builtin_amdgcn_wave_barrier();
  const uint64_t m = valid ? *reinterpret_cast<volatile unsigned long long *>(&mm[d]) : 0ULL;
  __builtin_amdgcn_wave_barrier();
  if (valid && (m & ((1ULL << lane) - 1ULL)) == 0) *reinterpret_cast<volatile unsigned long long *>(&mm[d]) = 0ULL;
  __builtin_amdgcn_wave_barrier();
  return m;
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));  // a dense (u, v) pair: one 8-byte access

struct FastGeom {
  uint32_t key_bits;  // bits of V - 1 (>= 1)
  uint32_t low;       // vertex-in-bucket bits
  uint32_t hb;        // bucket bits (key_bits - low)
  uint32_t pack;      // 1: B's input is one u32 word low(key) << key_bits | payload
  uint32_t sub;       // low = (sub, leaf): sub-bucket bits (<= FB_MAX_SUB) ...
  uint32_t leaf;      // ... and vertex-in-leaf bits (<= 6)
  uint32_t ss;        // row stride of the per-chunk / per-bucket sub-bucket offset tables: 65, or 129 with 7 sub bits
  uint32_t part;      // shard builds (gg_csr_build_shard): forward rows of owned sources, reverse rows of owned
  uint32_t n_parts;   // destinations only; 1: whole graph
  uint32_t rank_atomic;  // 1: stable ranks straight from ds_add_rtn (lds_order_ok), 0: from match masks
};
// Shard builds mark, in the dense pair, the direction a row does not take part in: bit 31 of u = "source not
// owned: no forward entry", bit 31 of v = "destination not owned: no reverse entry" (dense indices stay below 2^22).
constexpr uint32_t FB_SKIP = 0x80000000u;

// Stable ranking inside a wave.  Every ranking kernel below gives a wave its own row of LDS cursors, so the only
// lanes that meet on a cursor belong to one instruction of one wave.  gfx950 serves the lanes of one ds_add_rtn_u32
// that hit the same word in increasing lane order (scripts/ubench_ldsorder.hip: 1.3e11 lane-ops, none out of
// order; 5x the rate of a 9-bit match-mask loop), which is exactly a stable rank: position = atomicAdd(cursor, 1).
// That order is not an architectural promise, so k_lds_order_probe checks it once per context before the first
// build relies on it; where it does not hold the kernels build match masks with ballots instead.
__global__ __launch_bounds__(512) void k_lds_order_probe(uint32_t rounds, uint32_t *__restrict__ bad) {
  __shared__ uint32_t rows[8 * 512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ULL << lane) - 1ULL;
  const uint32_t K = blockIdx.x % 4 == 0 ? 1u : blockIdx.x % 4 == 1 ? 3u : blockIdx.x % 4 == 2 ? 16u : 512u;
  uint32_t *row = rows + wave * 512;
  for (uint32_t i = lane; i < 512; i += 64) row[i] = 0;
  __builtin_amdgcn_wave_barrier();
  uint32_t nbad = 0;
  for (uint32_t r = 0; r < rounds; r++) {
    uint32_t h = (blockIdx.x * 8 + wave) * 0x9E3779B9u + r * 64 + lane;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    const uint32_t a = (h >> 8) % K;
    const bool valid = (h & 7u) != 0;
    volatile uint32_t *vr = row;
    const uint32_t before = valid ? vr[a] : 0u;
    uint64_t m = __ballot(valid);
    for (uint32_t b = 0; b < 9; b++) {
      const uint64_t bb = __ballot((a >> b) & 1u);
      m &= ((a >> b) & 1u) ? bb : ~bb;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t got = 0;
    if (valid) got = atomicAdd(&row[a], 1u);
    __builtin_amdgcn_wave_barrier();
    if (valid && got != before + (uint32_t)__popcll(m & lane_lt)) nbad++;
  }
  if (nbad) atomicAdd(bad, nbad);
}

// rank_mode: 0 decide by the probe (once), 1 ds_add_rtn ranks, 2 match masks
static int lds_order_ok(gg_ctx *ctx, uint32_t *ok) {
  if (ctx->rank_mode == 0) {
    uint32_t *bad = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&bad, sizeof(uint32_t)));
    GG_HIP(hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_lds_order_probe, dim3(256), dim3(512), 0, ctx->stream, 256u, bad);
    const bool launched = hipGetLastError() == hipSuccess;  // a probe that did not run proves nothing: match masks
    uint32_t h = 1;
    GG_HIP(hipMemcpyAsync(&h, bad, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    ctx->dev_free(bad);
    ctx->rank_mode = launched && h == 0 ? 1 : 2;
  }
  *ok = ctx->rank_mode == 1 ? 1u : 0u;
  return GG_OK;
}

// ---- D: densify + per-tile bucket histograms of both directions ---------------------------------------------
// counts[tile * 2 nb + dir * nb + bucket]: tile-major, one contiguous 2 nb row per tile
template <int MODE>
__device__ __forceinline__ void densify_tile(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                             uint64_t E, uint64_t base, const HtSlot *__restrict__ ht, uint64_t cap,
                                             int64_t min_idx, const uint32_t *__restrict__ dir,
                                             const unsigned long long *__restrict__ tab,
                                             const DirectMap *__restrict__ dm, u32x2 *__restrict__ pairs,
                                             uint32_t low, uint32_t part, uint32_t n_parts, uint32_t *hist_f,
                                             uint32_t *hist_r) {
#ifndef GG_FB_DB
#define GG_FB_DB 2  // (4 rows per batch need 102 VGPRs = 16 waves per CU: 731 us at SF100 against 695 with 2 and 32 waves)
#endif
  constexpr int B = GG_FB_DB;  // edge rows per batch: 2*B independent first probes in flight per lane
  static_assert(FB_ITEMS % B == 0, "a tile is a whole number of batches");
  const int64_t min_id = dm->min_id, max_id = dm->max_id;
  PkGeom pk;
  pk.load(dm);
#pragma unroll 1
  for (int it0 = 0; it0 < FB_ITEMS; it0 += B) {
    int64_t ks[B], kd[B];
    uint32_t us[B], vs[B];
#pragma unroll
    for (int j = 0; j < B; j++) {
      const uint64_t e = base + (uint64_t)(it0 + j) * FB_THREADS + threadIdx.x;
      ks[j] = e < E ? ld_stream(src + e) : HT_EMPTY;
      kd[j] = e < E ? ld_stream(dst + e) : HT_EMPTY;
    }
    if (MODE == 99) {  // timing probe: the streams without the dictionary probes
#pragma unroll
      for (int j = 0; j < B; j++) {
        us[j] = (uint32_t)((uint64_t)ks[j] * DIG_GOLD >> 45);
        vs[j] = (uint32_t)((uint64_t)kd[j] * DIG_GOLD >> 45);
      }
    } else if (MODE == DICT_DIRECT) {
#pragma unroll
      for (int j = 0; j < B; j++) {
        us[j] = direct_lookup(dir, (uint64_t)min_id, ks[j]);
        vs[j] = direct_lookup(dir, (uint64_t)min_id, kd[j]);
      }
    } else if (MODE == DICT_PACKED8) {
      uint64_t hs[B], hd[B], ts[B], td[B];
      uint4 rs[B], rd[B];
#pragma unroll
      for (int j = 0; j < B; j++) {  // first probes of the whole batch issue back to back
        pk.locate(ks[j], &hs[j], &ts[j]);
        pk.locate(kd[j], &hd[j], &td[j]);
        rs[j] = *reinterpret_cast<const uint4 *>(&tab[2 * hs[j]]);
        rd[j] = *reinterpret_cast<const uint4 *>(&tab[2 * hd[j]]);
      }
#pragma unroll
      for (int j = 0; j < B; j++) {
        us[j] = packed_resolve(tab, pk, ks[j] >= min_id && ks[j] <= max_id, hs[j], ts[j], rs[j]);
        vs[j] = packed_resolve(tab, pk, kd[j] >= min_id && kd[j] <= max_id, hd[j], td[j], rd[j]);
      }
    } else {
      uint64_t ss[B], sd[B];
      uint4 rs[B], rd[B];
#pragma unroll
      for (int j = 0; j < B; j++) {
        ss[j] = ht_slot(ks[j], cap);
        sd[j] = ht_slot(kd[j], cap);
        rs[j] = *reinterpret_cast<const uint4 *>(&ht[ss[j]]);
        rd[j] = *reinterpret_cast<const uint4 *>(&ht[sd[j]]);
      }
#pragma unroll
      for (int j = 0; j < B; j++) {
        us[j] = ht_resolve(ht, cap, min_idx, ks[j], ss[j], rs[j]);
        vs[j] = ht_resolve(ht, cap, min_idx, kd[j], sd[j], rd[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < B; j++) {
      const uint64_t e = base + (uint64_t)(it0 + j) * FB_THREADS + threadIdx.x;
      if (e >= E) continue;
      uint32_t u = us[j], v = vs[j];
      if (u == INVALID_U32 || v == INVALID_U32) {
        u = INVALID_U32;  // dropped: an endpoint is not a vertex (inner-join semantics)
        v = INVALID_U32;
      } else {
        const bool fwd = owns(ks[j], part, n_parts), rev = owns(kd[j], part, n_parts);
        if (fwd) atomicAdd(&hist_f[u >> low], 1u);
        if (rev) atomicAdd(&hist_r[v >> low], 1u);
        if (!fwd && !rev) {
          u = INVALID_U32;  // a row of other shards only
          v = INVALID_U32;
        } else {
          u |= fwd ? 0u : FB_SKIP;
          v |= rev ? 0u : FB_SKIP;
        }
      }
      u32x2 pr;
      pr.x = u;
      pr.y = v;
      st_stream(pairs + e, pr);
    }
  }
}

#define GG_FB_DATTR
template <bool PROBE = false>  // PROBE: timing probe without dictionary lookups (its output is overwritten)
__global__ __launch_bounds__(FB_THREADS) GG_FB_DATTR void k_densify_pairs(
    const int64_t *__restrict__ src, const int64_t *__restrict__ dst, uint64_t E, const HtSlot *__restrict__ ht,
    uint64_t cap, const BuildStatus *__restrict__ st, const uint32_t *__restrict__ dir,
    const unsigned long long *__restrict__ tab, const DirectMap *__restrict__ dm, u32x2 *__restrict__ pairs,
    FastGeom g, uint64_t nblocks, uint32_t *__restrict__ counts) {
  __shared__ uint32_t hist[2 << FB_MAX_HB];
  const uint32_t nb = 1u << g.hb;
  for (uint32_t i = threadIdx.x; i < 2 * nb; i += FB_THREADS) hist[i] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * FB_TILE;
  const unsigned long long mode = PROBE ? 99ULL : dm->mode;  // uniform over the grid
  if (PROBE)
    densify_tile<99>(src, dst, E, base, ht, cap, st->min_idx, dir, tab, dm, pairs, g.low, g.part, g.n_parts, hist,
                          hist + nb);
  else if (mode == DICT_DIRECT)
    densify_tile<DICT_DIRECT>(src, dst, E, base, ht, cap, st->min_idx, dir, tab, dm, pairs, g.low, g.part, g.n_parts, hist,
                          hist + nb);
  else if (mode == DICT_PACKED8)
    densify_tile<DICT_PACKED8>(src, dst, E, base, ht, cap, st->min_idx, dir, tab, dm, pairs, g.low, g.part, g.n_parts, hist,
                          hist + nb);
  else
    densify_tile<DICT_WIDE16>(src, dst, E, base, ht, cap, st->min_idx, dir, tab, dm, pairs, g.low, g.part, g.n_parts, hist,
                          hist + nb);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 2 * nb; i += FB_THREADS)
    counts[(uint64_t)blockIdx.x * 2 * nb + i] = hist[i];  // i = dir * nb + bucket
}

// ---- A: two stable bucket partitions from one read of the pairs ---------------------------------------------
// Element order inside a tile: wave w owns rows [w * FB_WTILE, (w + 1) * FB_WTILE), 64 consecutive rows per
// step.  PACK: output word = low(key) << key_bits | payload; otherwise the pair (low(key), payload).
template <bool PACK, bool ROWID, int STOP = 0>  // STOP > 0: timing probes that write nothing (GG_FB_A_PROBE builds)
__global__ __launch_bounds__(FB_THREADS) void k_partition_dual(
    const u32x2 *__restrict__ pairs, uint64_t E, FastGeom g, uint64_t nblocks, const uint32_t *__restrict__ bases,
    const BuildStatus *__restrict__ st, uint32_t *__restrict__ out_f, uint32_t *__restrict__ out_r,
    uint32_t *__restrict__ epos_f) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t nb = 1u << g.hb;
  uint32_t *xw = lds;                                   // staged words (PACK) or low keys
  uint32_t *xp = xw + FB_TILE;                          // staged payloads (!PACK)
  uint32_t *xe = PACK ? xp : xp + FB_TILE;              // staged edge positions (ROWID, forward only)
  uint32_t *hw = ROWID ? xe + FB_TILE : xe;             // [FB_WAVES][nb] per-wave counts -> running cursors
  uint32_t *dbase = hw + FB_WAVES * nb;                 // [nb] first staged slot of each bucket
  uint32_t *gb = dbase + nb;                            // [nb] global position of the bucket's staged slot 0, minus dbase
  uint32_t *misc = gb + nb;                             // [FB_WAVES + 2]
  uint16_t *xd = reinterpret_cast<uint16_t *>(misc + FB_WAVES + 2);  // staged bucket numbers

  // blocks b and b + 8 share an XCD (speed only): give each XCD a contiguous range of tiles, so the short
  // runs that neighbouring tiles append to one bucket meet in one L2 before they are written back
  const uint64_t chunk = (nblocks + 7) / 8;
  const uint64_t tile = (uint64_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tile >= nblocks) return;
  const uint64_t tile_base = tile * FB_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t low_mask = (1u << g.low) - 1u;

  uint32_t u[FB_ITEMS], v[FB_ITEMS];
  const uint64_t wbase = tile_base + (uint64_t)wave * FB_WTILE;
#pragma unroll
  for (int it = 0; it < FB_ITEMS; it++) {
    const uint64_t idx = wbase + (uint64_t)it * 64 + lane;
    u32x2 p;
    p.x = INVALID_U32;
    p.y = INVALID_U32;
    if (idx < E) p = ld_stream(pairs + idx);
    u[it] = p.x;
    v[it] = p.y;
  }
  const uint64_t lane_lt = (1ULL << lane) - 1ULL;
  uint32_t *myh = hw + wave * nb;
  if (STOP == 1) {  // timing probe: loads only
    uint32_t acc = 0;
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++) acc += u[it] ^ v[it];
    if (acc == 0x12345678u && E == 1) out_f[0] = acc;
    return;
  }

#pragma unroll 1
  for (int dir = 0; dir < 2; dir++) {
    for (uint32_t i = threadIdx.x; i < FB_WAVES * nb; i += FB_THREADS) hw[i] = 0;
    __syncthreads();
    // (decided once per wave and direction, on its first 64 rows: the table's order does not change in between)
    const bool runs = wave_has_runs((dir ? v[0] : u[0]) >> g.low,
                                    u[0] != INVALID_U32 && !((dir ? v[0] : u[0]) & FB_SKIP), lane);
    if (runs) {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) {
        const uint32_t key = dir ? v[it] : u[it];
        run_add<false>(myh, key >> g.low, u[it] != INVALID_U32 && !(key & FB_SKIP), lane);
      }
    } else {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) {
        const uint32_t key = dir ? v[it] : u[it];
        if (u[it] != INVALID_U32 && !(key & FB_SKIP)) atomicAdd(&myh[key >> g.low], 1u);
      }
    }
    __syncthreads();
    // per bucket: wave counts -> staged-slot cursors (first staged slot of the bucket + the waves before); the
    // tile's global positions come from the column kernels' bases (tile-major: one coalesced row per direction)
    {
      const uint32_t dpt = (nb + FB_THREADS - 1) / FB_THREADS;  // buckets per thread (1 or 2)
      uint32_t tot[2] = {0, 0};
      for (uint32_t q = 0; q < dpt; q++) {
        const uint32_t d = threadIdx.x * dpt + q;
        if (d < nb) {
          uint32_t t = 0;
#pragma unroll
          for (int w = 0; w < FB_WAVES; w++) {  // counts -> exclusive offsets across the waves
            const uint32_t c = hw[w * nb + d];
            hw[w * nb + d] = t;
            t += c;
          }
          tot[q] = t;
        }
      }
      const uint32_t mine = tot[0] + tot[1];
      const uint32_t incl = wave_scan_incl(mine);  // (DPP moves, gg_internal.h)
      if (lane == 63) misc[wave] = incl;
      __syncthreads();
      uint32_t wb = 0, all = 0;
#pragma unroll
      for (int w = 0; w < FB_WAVES; w++) {
        const uint32_t sv = misc[w];
        if (w < wave) wb += sv;
        all += sv;
      }
      uint32_t ex = wb + incl - mine;
      for (uint32_t q = 0; q < dpt; q++) {
        const uint32_t d = threadIdx.x * dpt + q;
        if (d < nb) {
          gb[d] = bases[tile * 2 * nb + (uint64_t)dir * nb + d] - ex;
#pragma unroll
          for (int w = 0; w < FB_WAVES; w++) hw[w * nb + d] += ex;  // ... + the bucket's first staged slot
          ex += tot[q];
        }
      }
      if (threadIdx.x == 0) misc[FB_WAVES] = all;  // valid rows in this tile
    }
    __syncthreads();

    if (STOP == 2) continue;  // timing probe: counts and scans only
    // rank inside the tile (stable) and stage in LDS in bucket order
    volatile uint32_t *cur = myh;
    auto stage = [&](int it, uint32_t pos, uint32_t key, uint32_t pay, uint32_t d) {
      if (PACK) {
        xw[pos] = ((key & low_mask) << g.key_bits) | pay;
      } else {
        xw[pos] = key & low_mask;
        xp[pos] = pay;
      }
      if (ROWID && dir == 0) xe[pos] = (uint32_t)(wbase + (uint64_t)it * 64 + lane);
      xd[pos] = (uint16_t)d;
    };
    if (g.rank_atomic && !runs) {  // the common case on its own: sixteen adds in flight, no wave-wide tests between
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) {
        const uint32_t key = dir ? v[it] : u[it], pay = (dir ? u[it] : v[it]) & ~FB_SKIP;
        if (u[it] != INVALID_U32 && !(key & FB_SKIP)) {  // (shards: this direction's endpoint is owned)
          const uint32_t d = key >> g.low;
          stage(it, atomicAdd((uint32_t *)&cur[d], 1u), key, pay, d);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) {
        const uint32_t key = dir ? v[it] : u[it], pay = (dir ? u[it] : v[it]) & ~FB_SKIP;
        const bool valid = u[it] != INVALID_U32 && !(key & FB_SKIP);
        const uint32_t d = valid ? key >> g.low : 0u;
        uint64_t m = __ballot(valid);
        uint32_t pos = 0;
        bool counted = false;  // the cursor already moved
        if (g.rank_atomic) {
          // runs: one bucket for the whole wave (a bucket is tens of thousands of entries), lanes in order, one
          // add.  (run_add<true> here costs nine registers, and the workgroup its second slot on the CU.)
          const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, m ? __ffsll((unsigned long long)m) - 1 : 0);
          if (__ballot(valid && d != d0) != 0) {
            if (valid) pos = atomicAdd((uint32_t *)&cur[d], 1u);
            counted = true;
          }
        } else {
          for (uint32_t bit = 0; bit < g.hb; bit++) {  // match mask: same bucket within the wave
            const uint64_t bb = __ballot((d >> bit) & 1u);
            m &= ((d >> bit) & 1u) ? bb : ~bb;
          }
        }
        if (valid) {
          if (!counted) pos = cur[d] + __popcll(m & lane_lt);
          stage(it, pos, key, pay, d);
        }
        if (!counted) {  // uniform
          __builtin_amdgcn_wave_barrier();
          if (valid && (m & lane_lt) == 0) atomicAdd((uint32_t *)&cur[d], (uint32_t)__popcll(m));  // lowest lane of each group
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    __syncthreads();

    if (STOP == 3) continue;  // timing probe: no write-out
    // write out: consecutive staged slots of one bucket -> consecutive global positions
    const uint32_t nvalid = misc[FB_WAVES];
    uint32_t *out = dir ? out_r : out_f;
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++) {
      const uint32_t i = (uint32_t)it * FB_THREADS + threadIdx.x;
      if (i < nvalid) {
        const uint32_t pos = gb[xd[i]] + i;
        if (PACK) {
          out[pos] = xw[i];
        } else {
          reinterpret_cast<uint2 *>(out)[pos] = make_uint2(xw[i], xp[i]);
        }
        if (ROWID && dir == 0) epos_f[pos] = xe[i];
      }
    }
    __syncthreads();
  }
}

// ---- B: rows of one bucket, in three kernels ---------------------------------------------------------------------
// A bucket's entries carry low(key) = (sub, leaf).
//   k_sub_sort    one workgroup per CHUNK (FB_TILE consecutive entries of one bucket): sorts the chunk, stably,
//                 by sub-bucket (<= 64) in LDS and writes it back IN PLACE, plus the chunk's 65 sub-bucket offsets
//   k_sub_totals  per bucket: where each (bucket, sub) leaf starts in the final arrays (sum over the chunks)
//   k_leaf_rows   one WAVE per leaf (<= 64 vertices, ~1 k entries at SF100): gathers the leaf's run out of every
//                 chunk of its bucket (chunk order = rowid order), counts per vertex, writes the row offsets, ranks
//                 the entries stably into a wave-private LDS stage and writes the rows out contiguously
// Earlier forms of this step, one workgroup per bucket streaming it several times, ran at 1.4-2.0 ms at SF100:
// scattering single words to their rows kept tens of MB of half-written lines alive per XCD (2.3x the bytes
// written), and eight waves walking 78 k entries one 2 KB batch at a time were bound by memory latency.
#ifndef GG_FB_CAPW
#define GG_FB_CAPW 1536  // entries of a wave's LDS stage in k_leaf_rows (halved when edge positions ride along)
#endif
#ifndef GG_FB_LEAF_TARGET
#define GG_FB_LEAF_TARGET 1024  // entries per leaf aimed at: between half of this and this
#endif
#ifndef GG_FB_MAX_SUB
#define GG_FB_MAX_SUB 7  // sub-bucket bits: up to 128 sub-buckets per bucket (the chunk sort's lanes then own two each)
#endif
constexpr int FB_MAX_SUB = GG_FB_MAX_SUB;
static_assert(FB_MAX_SUB == 6 || FB_MAX_SUB == 7, "one or two sub-buckets per lane of the chunk sort's scan");
#ifndef GG_FB_LEAF_WAVES
#define GG_FB_LEAF_WAVES 1  // (4 leaves per workgroup hold their LDS until the largest is done: 366 us against 349 at SF100)
#endif
constexpr int LEAF_WAVES = GG_FB_LEAF_WAVES;  // waves (= leaves) per workgroup of k_leaf_rows
#ifndef GG_FB_LEAF_MAXS
#define GG_FB_LEAF_MAXS 24
#endif
constexpr int LEAF_MAXS = GG_FB_LEAF_MAXS;  // 64-entry steps a wave of k_leaf_rows keeps in registers (<= 32)

// ---- column kernels: counters -> global positions ----------------------------------------------------------------
// counts[tile][2 nb] (tile-major) must become bases[tile][c] = start of column c (its direction's bucket start)
// + the counts of the earlier tiles in column c.  Three small kernels over groups of `gsz` tiles:
//   k_col_partial  partial[g][c] = sum of the group's counts in column c
//   k_col_scan     (one workgroup) partial[g][c] <- bucket start + sums of the earlier groups; bucket starts, chunk
//                  table and kept-edge count on the way (every figure the kernels below need)
//   k_col_apply    counts[tile][c] <- partial[g][c] + counts of the group's earlier tiles, in place
// Every access is a coalesced row.  (The first version kept the counters bucket-major for a chained scan: D wrote
// and A read 1024 separate lines per tile, as many bytes again as A's payload.)
//   bstart[dir * (nb + 1) + j]   first position of bucket j in that direction's partitioned array
//   cstart[i], i = dir * nb + j  first chunk of bucket i (cstart[2 nb] = number of chunks)
//   part_of[p]                   chunk p: {first entry, end, bucket i} (one load in k_sub_sort instead of a chain)
__global__ __launch_bounds__(1024) void k_col_partial(const uint32_t *__restrict__ counts, uint64_t nblocks,
                                                     uint32_t ncol, uint32_t gsz, uint32_t *__restrict__ partial,
                                                     uint32_t *__restrict__ coltot /* [ncol], zeroed */) {
  const uint64_t t0 = (uint64_t)blockIdx.x * gsz, t1 = t0 + gsz < nblocks ? t0 + gsz : nblocks;
  for (uint32_t c = threadIdx.x; c < ncol; c += 1024) {
    uint32_t sum = 0;
    for (uint64_t tb = t0; tb < t1; tb += 16) {
      uint32_t v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = tb + q < t1 ? counts[(tb + q) * ncol + c] : 0u;
#pragma unroll
      for (int q = 0; q < 16; q++) sum += v[q];
    }
    partial[(uint64_t)blockIdx.x * ncol + c] = sum;
    if (sum) atomicAdd(&coltot[c], sum);  // integer adds: the total does not depend on their order
  }
}

// one workgroup: bucket starts of both directions (scan of the column totals), chunk table, kept-edge count
__global__ __launch_bounds__(1024) void k_col_scan(uint32_t *__restrict__ coltot /* in: totals, out: starts */,
                                                   uint32_t nb, uint32_t *__restrict__ bstart,
                                                   uint32_t *__restrict__ cstart, uint4 *__restrict__ part_of,
                                                   BuildStatus *__restrict__ st) {
  __shared__ uint32_t s_b[2 * ((1 << FB_MAX_HB) + 1)];  // bucket starts, both directions
  __shared__ uint32_t s_w[2][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // thread j owns bucket j of BOTH directions (nb <= 1024)
  uint32_t tot[2] = {0, 0};
  if (threadIdx.x < nb) {
    tot[0] = coltot[threadIdx.x];
    tot[1] = coltot[nb + threadIdx.x];
  }
  uint32_t incl[2] = {tot[0], tot[1]};
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t a = __shfl_up(incl[0], o, 64), b = __shfl_up(incl[1], o, 64);
    if (lane >= o) {
      incl[0] += a;
      incl[1] += b;
    }
  }
  if (lane == 63) {
    s_w[0][wave] = incl[0];
    s_w[1][wave] = incl[1];
  }
  __syncthreads();
  uint32_t ex[2] = {incl[0] - tot[0], incl[1] - tot[1]}, all[2] = {0, 0};
  for (int w = 0; w < 16; w++) {
    if (w < wave) {
      ex[0] += s_w[0][w];
      ex[1] += s_w[1][w];
    }
    all[0] += s_w[0][w];
    all[1] += s_w[1][w];
  }
  if (threadIdx.x < nb) {
    for (int dir = 0; dir < 2; dir++) {
      s_b[dir * (nb + 1) + threadIdx.x] = ex[dir];
      bstart[dir * (nb + 1) + threadIdx.x] = ex[dir];
      coltot[dir * nb + threadIdx.x] = ex[dir];  // k_col_apply starts every column from its bucket start
    }
  }
  if (threadIdx.x == 0) {
    for (int dir = 0; dir < 2; dir++) {
      s_b[dir * (nb + 1) + nb] = all[dir];
      bstart[dir * (nb + 1) + nb] = all[dir];
    }
    st->kept = all[0];  // both directions count the same rows
    st->kept_rev = all[1];
  }
  __syncthreads();
  // chunks per bucket, exclusive scan over the 2 nb buckets (consecutive buckets per thread), chunk -> bucket table
  const uint32_t per = (2 * nb + 1023) / 1024;
  uint32_t mine = 0;
  for (uint32_t q = 0; q < per; q++) {
    const uint32_t i = threadIdx.x * per + q;
    if (i < 2 * nb) {
      const uint32_t dir = i / nb, j = i % nb;
      const uint32_t len = s_b[dir * (nb + 1) + j + 1] - s_b[dir * (nb + 1) + j];
      mine += (len + FB_TILE - 1) / FB_TILE;
    }
  }
  const uint32_t cincl = wave_scan_incl(mine);  // (DPP moves, gg_internal.h)
  __syncthreads();
  if (lane == 63) s_w[0][wave] = cincl;
  __syncthreads();
  uint32_t cex = cincl - mine;
  for (int w = 0; w < wave; w++) cex += s_w[0][w];
  for (uint32_t q = 0; q < per; q++) {
    const uint32_t i = threadIdx.x * per + q;
    if (i < 2 * nb) {
      const uint32_t dir = i / nb, j = i % nb;
      const uint32_t len = s_b[dir * (nb + 1) + j + 1] - s_b[dir * (nb + 1) + j];
      const uint32_t n = (len + FB_TILE - 1) / FB_TILE;
      cstart[i] = cex;
      const uint32_t b0 = s_b[dir * (nb + 1) + j], b1 = b0 + len;
      for (uint32_t c = 0; c < n; c++) {  // stores are not waited for: cheap even for one huge bucket
        const uint32_t c0 = b0 + c * FB_TILE;
        part_of[cex + c] = make_uint4(c0, c0 + FB_TILE < b1 ? c0 + FB_TILE : b1, i, 0u);
      }
      cex += n;
    }
  }
  if (threadIdx.x == 1023) cstart[2 * nb] = cex;
}

// bases in place: counts[tile][c] <- bucket start + counts of every earlier tile in column c.  A group adds up the
// partial rows of the groups before it (L2-resident: the whole matrix is ~1 MB) instead of waiting for a scan.
__global__ __launch_bounds__(1024) void k_col_apply(uint32_t *__restrict__ counts, uint64_t nblocks, uint32_t ncol,
                                                    uint32_t gsz, const uint32_t *__restrict__ partial,
                                                    const uint32_t *__restrict__ colstart) {
  const uint64_t t0 = (uint64_t)blockIdx.x * gsz, t1 = t0 + gsz < nblocks ? t0 + gsz : nblocks;
  for (uint32_t c = threadIdx.x; c < ncol; c += 1024) {
    uint32_t run = colstart[c];
    uint32_t g = 0;
    for (; g + 8 <= blockIdx.x; g += 8) {  // eight independent row reads in flight
      uint32_t v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = partial[(uint64_t)(g + q) * ncol + c];
#pragma unroll
      for (int q = 0; q < 8; q++) run += v[q];
    }
    for (; g < blockIdx.x; g++) run += partial[(uint64_t)g * ncol + c];
    uint32_t v[16];
    for (uint64_t tb = t0; tb < t1; tb += 16) {
#pragma unroll
      for (int q = 0; q < 16; q++) v[q] = tb + q < t1 ? counts[(tb + q) * ncol + c] : 0u;
#pragma unroll
      for (int q = 0; q < 16; q++) {
        if (tb + q < t1) counts[(tb + q) * ncol + c] = run;
        run += v[q];
      }
    }
  }
}

// Wave 0 of a chunk sort: per-sub-bucket counts of the waves (hw[wave][NSUBP]) -> staged-slot cursors, the chunk's
// row of sub-bucket offsets and its entry count.  Lane l owns sub-buckets l * SPL .. l * SPL + SPL - 1 (SPL = 2
// with 7 sub-bucket bits); sub-buckets the geometry does not use count nothing and come out as the total.
template <int SPL>
__device__ __forceinline__ void sub_cursors(uint32_t *hw, uint32_t *dbase, uint32_t *__restrict__ offs_row, int lane,
                                            uint32_t *s_n) {
  constexpr int NSUBP = 64 * SPL;
  uint32_t cw[SPL][FB_WAVES], tot[SPL], mine = 0;
#pragma unroll
  for (int j = 0; j < SPL; j++) {
    tot[j] = 0;
#pragma unroll
    for (int q = 0; q < FB_WAVES; q++) {
      cw[j][q] = hw[q * NSUBP + lane * SPL + j];
      tot[j] += cw[j][q];
    }
    mine += tot[j];
  }
  const uint32_t incl = wave_scan_incl(mine);  // (DPP moves, gg_internal.h)
  uint32_t run = incl - mine;
#pragma unroll
  for (int j = 0; j < SPL; j++) {
    dbase[lane * SPL + j] = run;
    offs_row[lane * SPL + j] = run;
#pragma unroll
    for (int q = 0; q < FB_WAVES; q++) {
      hw[q * NSUBP + lane * SPL + j] = run;
      run += cw[j][q];
    }
  }
  if (lane == 63) {
    dbase[NSUBP] = incl;
    offs_row[NSUBP] = incl;
    *s_n = incl;
  }
}

// Chunk-local stable sort by sub-bucket, in place.  Element order in a chunk: wave w owns entries
// [w * FB_WTILE, (w + 1) * FB_WTILE), 64 consecutive entries per step.
template <bool PACK, bool ROWID, int SPL, int STOP = 0>  // STOP: timing probes (GG_FB_PROBES)
__global__ __launch_bounds__(FB_THREADS) void k_sub_sort(uint32_t *__restrict__ buf_f, uint32_t *__restrict__ buf_r,
                                                         uint32_t *__restrict__ epos_f,
                                                         const uint32_t *__restrict__ bstart,
                                                         const uint32_t *__restrict__ cstart,
                                                         const uint4 *__restrict__ part_of, FastGeom g,
                                                         uint32_t *__restrict__ offs /* [chunk][g.ss] */) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t s_n;
  constexpr int NSUBP = 64 * SPL;
  const uint32_t nb = 1u << g.hb;
  const uint32_t p = blockIdx.x;
  const uint4 chunk = part_of[p];  // (both loads in flight: entries past the last chunk are allocated, not meaningful)
  if (p >= cstart[2 * nb]) return;
  const uint32_t c0 = chunk.x, c1 = chunk.y, dir = chunk.z / nb;
  uint32_t *__restrict__ buf = dir ? buf_r : buf_f;
  const bool with_pos = ROWID && dir == 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ULL << lane) - 1ULL;
  uint32_t *hw = lds;                         // [FB_WAVES][NSUBP] per-wave counts -> cursors (staged slot)
  uint32_t *dbase = hw + FB_WAVES * NSUBP;    // [NSUBP + 1] first staged slot of each sub-bucket
  uint32_t *xw = dbase + NSUBP + 1;           // staged words (PACK) or low keys
  uint32_t *xp = xw + FB_TILE;                // staged payloads (!PACK)
  uint32_t *xe = PACK ? xp : xp + FB_TILE;    // staged edge positions (ROWID)
  uint32_t *myh = hw + wave * NSUBP;
#pragma unroll
  for (int j = 0; j < SPL; j++) myh[lane + 64 * j] = 0;

  uint32_t k[FB_ITEMS], w[FB_ITEMS], ep[FB_ITEMS];
  const uint32_t wbase = c0 + (uint32_t)wave * FB_WTILE;
#pragma unroll
  for (int it = 0; it < FB_ITEMS; it++) {
    const uint32_t e = wbase + it * 64 + lane;
    k[it] = INVALID_U32;
    w[it] = 0;
    ep[it] = 0;
    if (e < c1) {
      if (PACK) {
        w[it] = buf[e];
        k[it] = w[it] >> g.key_bits;
      } else {
        const uint2 x = reinterpret_cast<const uint2 *>(buf)[e];
        k[it] = x.x;
        w[it] = x.y;
      }
      if (with_pos) ep[it] = epos_f[e];
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (STOP == 1) {  // loads only
    uint32_t acc = 0;
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++) acc += k[it] ^ w[it];
    if (acc == 0x12345678u && c1 == 1) offs[0] = acc;
    return;
  }
  const bool runs = wave_has_runs(k[0] >> g.leaf, k[0] != INVALID_U32, lane);  // (see k_partition_dual)
  if (runs) {
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++) run_add<false>(myh, k[it] >> g.leaf, k[it] != INVALID_U32, lane);
  } else {
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++)
      if (k[it] != INVALID_U32) atomicAdd(&myh[k[it] >> g.leaf], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64) sub_cursors<SPL>(hw, dbase, offs + (uint64_t)p * g.ss, lane, &s_n);  // wave 0
  __syncthreads();
  if (STOP == 2) return;  // + counts and the scan
  volatile uint32_t *cur = myh;
#pragma unroll
  for (int it = 0; it < FB_ITEMS; it++) {
    const bool valid = k[it] != INVALID_U32;
    const uint32_t sb = valid ? k[it] >> g.leaf : 0u;
    uint64_t m = __ballot(valid);
    uint32_t pos = 0;
    bool counted = false;
    if (g.rank_atomic) {
      if (runs) {
        pos = run_add<true>((uint32_t *)cur, sb, valid, lane);
      } else if (valid) {
        pos = atomicAdd((uint32_t *)&cur[sb], 1u);
      }
      counted = true;
    } else {
      for (uint32_t bit = 0; bit < g.sub; bit++) {  // match mask: same sub-bucket within the wave
        const uint64_t bb = __ballot((sb >> bit) & 1u);
        m &= ((sb >> bit) & 1u) ? bb : ~bb;
      }
    }
    if (valid) {
      if (!counted) pos = cur[sb] + __popcll(m & lane_lt);
      if (PACK) {
        xw[pos] = w[it];
      } else {
        xw[pos] = k[it];
        xp[pos] = w[it];
      }
      if (with_pos) xe[pos] = ep[it];
    }
    if (!counted) {  // uniform
      __builtin_amdgcn_wave_barrier();
      if (valid && (m & lane_lt) == 0) atomicAdd((uint32_t *)&cur[sb], (uint32_t)__popcll(m));
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  if (STOP == 3) {  // + ranks and staging, no write-out
    if (xw[threadIdx.x] == 0x12345678u && c1 == 1) offs[0] = 1;
    return;
  }
  const uint32_t n = s_n;
#pragma unroll
  for (int it = 0; it < FB_ITEMS; it++) {
    const uint32_t s = (uint32_t)it * FB_THREADS + threadIdx.x;
    if (s < n) {
      if (PACK) {
        buf[c0 + s] = xw[s];
      } else {
        reinterpret_cast<uint2 *>(buf)[c0 + s] = make_uint2(xw[s], xp[s]);
      }
      if (with_pos) epos_f[c0 + s] = xe[s];
    }
  }
}

// The same sort for packed words without edge positions (the common case), software-pipelined: a workgroup walks
// chunks p, p + G, p + 2G, ... and has the NEXT chunk's entries in flight while it counts, ranks and writes the
// current one.  One chunk per workgroup spends its life in latencies (probe: the loads alone take 105 of the
// kernel's 195 us at SF100, a third of the HBM rate): header -> entries -> four barrier-separated phases, and 32
// resident waves per CU do not cover that.
template <int SPL>
__global__ __launch_bounds__(FB_THREADS) void k_sub_sort_pipe(uint32_t *__restrict__ buf_f, uint32_t *__restrict__ buf_r,
                                                              const uint32_t *__restrict__ cstart,
                                                              const uint4 *__restrict__ part_of, FastGeom g,
                                                              uint32_t *__restrict__ offs /* [chunk][g.ss] */) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t s_n;
  constexpr int NSUBP = 64 * SPL;
  const uint32_t nb = 1u << g.hb, G = gridDim.x;
  uint32_t p = blockIdx.x;
  uint4 chunk = part_of[p];  // (entries past the last chunk are allocated, not meaningful)
  const uint32_t nchunks = cstart[2 * nb];
  if (p >= nchunks) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *hw = lds;                       // [FB_WAVES][NSUBP] per-wave counts -> cursors (staged slot)
  uint32_t *dbase = hw + FB_WAVES * NSUBP;  // [NSUBP + 1] first staged slot of each sub-bucket
  uint32_t *xw = dbase + NSUBP + 1;         // staged words
  uint32_t *myh = hw + wave * NSUBP;
  const uint32_t mine = (uint32_t)wave * FB_WTILE + lane;  // this lane's first entry inside a chunk
  uint32_t w[FB_ITEMS], wn[FB_ITEMS];
  {
    const uint32_t *__restrict__ src = (chunk.z / nb ? buf_r : buf_f) + chunk.x;
    const uint32_t len = chunk.y - chunk.x;
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++) w[it] = mine + it * 64 < len ? ld_stream(src + mine + it * 64) : 0u;
  }
  uint4 chunk_n = p + G < nchunks ? part_of[p + G] : make_uint4(0, 0, 0, 0);
  for (;;) {
    const bool has_next = p + G < nchunks;  // uniform
    uint4 chunk_nn = make_uint4(0, 0, 0, 0);
    if (has_next) {
      const uint32_t *__restrict__ src = (chunk_n.z / nb ? buf_r : buf_f) + chunk_n.x;
      const uint32_t len = chunk_n.y - chunk_n.x;
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) wn[it] = mine + it * 64 < len ? ld_stream(src + mine + it * 64) : 0u;
      if (p + 2 * G < nchunks) chunk_nn = part_of[p + 2 * G];
    }
    uint32_t *__restrict__ buf = (chunk.z / nb ? buf_r : buf_f) + chunk.x;
    const uint32_t len = chunk.y - chunk.x;
#pragma unroll
    for (int j = 0; j < SPL; j++) myh[lane + 64 * j] = 0;
    __builtin_amdgcn_wave_barrier();
    const bool runs = wave_has_runs(w[0] >> (g.key_bits + g.leaf), mine < len, lane);  // (see k_partition_dual)
    if (runs) {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++)
        run_add<false>(myh, w[it] >> (g.key_bits + g.leaf), mine + it * 64 < len, lane);
    } else {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++)
        if (mine + it * 64 < len) atomicAdd(&myh[w[it] >> (g.key_bits + g.leaf)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) sub_cursors<SPL>(hw, dbase, offs + (uint64_t)p * g.ss, lane, &s_n);  // wave 0
    __syncthreads();
    if (runs) {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) {
        const bool valid = mine + it * 64 < len;
        const uint32_t pos = run_add<true>(myh, w[it] >> (g.key_bits + g.leaf), valid, lane);
        if (valid) xw[pos] = w[it];
      }
    } else {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) {  // ranks straight from the wave's cursors (lane-ordered ds_add_rtn)
        if (mine + it * 64 < len) {
          const uint32_t pos = atomicAdd(&myh[w[it] >> (g.key_bits + g.leaf)], 1u);
          xw[pos] = w[it];
        }
      }
    }
    __syncthreads();
    const uint32_t n = s_n;
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++) {
      const uint32_t s = (uint32_t)it * FB_THREADS + threadIdx.x;
      if (s < n) buf[s] = xw[s];
    }
    if (!has_next) break;
    __syncthreads();  // the stage and the cursors are reused
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++) w[it] = wn[it];
    chunk = chunk_n;
    chunk_n = chunk_nn;
    p += G;
  }
}

// substart[i * g.ss + s] = first final position of leaf (bucket i, sub s); [i * g.ss + 64 SPL] = end of the bucket
template <int SPL>
__global__ __launch_bounds__(64) void k_sub_totals(const uint32_t *__restrict__ offs,
                                                   const uint32_t *__restrict__ bstart,
                                                   const uint32_t *__restrict__ cstart, FastGeom g, uint64_t V,
                                                   uint32_t *__restrict__ substart, uint32_t *__restrict__ off,
                                                   uint32_t *__restrict__ roff) {
  const uint32_t nb = 1u << g.hb, i = blockIdx.x, dir = i / nb, j = i % nb;
  const int lane = threadIdx.x;
  const uint32_t p0 = cstart[i], p1 = cstart[i + 1];
  uint32_t tot[SPL], mine = 0;
#pragma unroll
  for (int q = 0; q < SPL; q++) tot[q] = 0;
  for (uint32_t p = p0; p < p1; p++) {
    const uint32_t *row = offs + (uint64_t)p * g.ss + lane * SPL;
#pragma unroll
    for (int q = 0; q < SPL; q++) tot[q] += row[q + 1] - row[q];
  }
#pragma unroll
  for (int q = 0; q < SPL; q++) mine += tot[q];
  const uint32_t incl = wave_scan_incl(mine);  // (DPP moves, gg_internal.h)
  const uint32_t b0 = bstart[dir * (nb + 1) + j];
  uint32_t run = b0 + incl - mine;
#pragma unroll
  for (int q = 0; q < SPL; q++) {
    substart[(uint64_t)i * g.ss + lane * SPL + q] = run;
    run += tot[q];
  }
  if (lane == 63) substart[(uint64_t)i * g.ss + 64 * SPL] = b0 + incl;
  // no leaf holds vertex V when V is a multiple of the bucket width: the last bucket closes the offsets
  if (lane == 0 && j == nb - 1 && ((uint64_t)nb << g.low) == V) (dir ? roff : off)[V] = bstart[dir * (nb + 1) + nb];
}

#define GG_STAMP(k)

template <bool PACK, bool ROWID>
__device__ __forceinline__ void leaf_unit(uint32_t unit, uint32_t *lds, const uint32_t *__restrict__ buf_f,
                              const uint32_t *__restrict__ buf_r, const uint32_t *__restrict__ epos_f,
                              const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ cstart,
                              const uint32_t *__restrict__ offs, const uint32_t *__restrict__ substart, FastGeom g,
                              uint64_t V, uint32_t *__restrict__ off, uint32_t *__restrict__ nbr,
                              uint32_t *__restrict__ epos, uint32_t *__restrict__ roff, uint32_t *__restrict__ rnbr,
                              uint32_t *__restrict__ rrow) {
  const uint32_t nb = 1u << g.hb, nsub = 1u << g.sub, leafW = 1u << g.leaf;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t i = unit / nsub, s = unit % nsub, dir = i / nb, j = i % nb;
  const uint32_t *__restrict__ buf = dir ? buf_r : buf_f;
  uint32_t *__restrict__ o_off = dir ? roff : off;
  uint32_t *__restrict__ o_nbr = dir ? rnbr : nbr;
  const bool with_pos = ROWID && dir == 0;
  constexpr uint32_t CAPW = ROWID ? GG_FB_CAPW / 2 : GG_FB_CAPW;
  uint32_t *lc = lds + (size_t)wave * (64 + GG_FB_CAPW);  // [64] vertex counters -> cursors
  uint32_t *stage = lc + 64;
  uint32_t *stage_e = stage + CAPW;
  const uint64_t lane_lt = (1ULL << lane) - 1ULL;
  const uint32_t leaf_mask = leafW - 1u;
  const uint32_t pay_mask = g.key_bits >= 32 ? 0xFFFFFFFFu : (1u << g.key_bits) - 1u;
  const uint32_t t0 = substart[(uint64_t)i * g.ss + s], t1 = substart[(uint64_t)i * g.ss + s + 1], n = t1 - t0;
  const uint32_t b0 = bstart[dir * (nb + 1) + j];
  const uint32_t p0 = cstart[i], nch = cstart[i + 1] - p0;
  const uint64_t vfirst = ((uint64_t)j << g.low) + ((uint64_t)s << g.leaf);  // first vertex of the leaf
  lc[lane] = 0;
  const bool staged = n <= CAPW;  // wave-uniform
  GG_STAMP(0);  // header loaded

  // The leaf's entries: run `s` of every chunk of the bucket, chunk after chunk.  Chunks are taken 64 at a time
  // (lane c holds chunk c's run); a run is walked in 64-entry steps.  pass 0 counts, pass 1 places; when the
  // whole leaf fits LEAF_MAXS steps the entries stay in registers between the two.
  uint32_t incl_keep = 0;  // lane d: end of vertex d's run (relative), set after pass 0
  uint32_t kw[LEAF_MAXS], pw[PACK ? 1 : LEAF_MAXS], ew[ROWID ? LEAF_MAXS : 1];
  uint32_t have = 0;  // bit q: kw[q] holds an entry
  bool single = false;
  bool runs = false;  // the leaf's entries come in runs of one vertex (decided on its first 64)
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    for (uint32_t cg = 0; cg < nch || cg == 0; cg += 64) {
      const uint32_t c = cg + lane;
      uint32_t so = 0, len = 0;
      if (c < nch) {
        so = offs[(uint64_t)(p0 + c) * g.ss + s];
        len = offs[(uint64_t)(p0 + c) * g.ss + s + 1] - so;
      }
      const uint32_t src = b0 + c * FB_TILE + so;  // position of the run's first entry
      const uint32_t st_c = (len + 63) / 64;       // steps of this run
      const uint32_t sincl = wave_scan_incl(st_c);  // (DPP moves, gg_internal.h)
      const uint32_t sexcl = sincl - st_c;
      const uint32_t T = __shfl(sincl, 63, 64);  // steps in this group of chunks
      if (pass == 0) single = nch <= 64 && T <= LEAF_MAXS;
      if (cg == 0) GG_STAMP(pass == 0 ? 1 : 4);  // run table loaded
      for (uint32_t r = 0; r < T; r += LEAF_MAXS) {
        if (pass == 0 || !single) {
          have = 0;
#pragma unroll
          for (int q = 0; q < LEAF_MAXS; q++) {  // all loads of the round issue back to back
            const uint32_t t = r + q;
            kw[q] = 0;
            if (!PACK) pw[PACK ? 0 : q] = 0;
            if (ROWID) ew[ROWID ? q : 0] = 0;
            if (t < T) {  // uniform
              const int cc = __popcll(__ballot(sincl <= t));  // the run this step belongs to (< 64: t < T)
              // (the run is the same for every lane: v_readlane into scalar registers, no ds_bpermute before the load)
              const uint32_t kidx = (t - (uint32_t)__builtin_amdgcn_readlane((int)sexcl, cc)) * 64 + lane;
              const uint32_t run_src = (uint32_t)__builtin_amdgcn_readlane((int)src, cc);
              const uint32_t run_len = (uint32_t)__builtin_amdgcn_readlane((int)len, cc);
              const uint32_t e = run_src + kidx;
              if (kidx < run_len) {
                have |= 1u << q;
                if (PACK) {
                  // the run as a buffer of its own (scalar 64-bit base: partitions of more than 2^30 entries are
                  // fine, the vector offset stays inside one chunk)
                  kw[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(buf + run_src), 0, (int)(run_len * 4u),
                                                        0x00020000),
                      kidx * 4u, 0, 0);
                } else {
                  const uint2 x = reinterpret_cast<const uint2 *>(buf)[e];
                  kw[q] = x.x;
                  pw[PACK ? 0 : q] = x.y;
                }
                if (with_pos) ew[ROWID ? q : 0] = epos_f[e];
              }
            }
          }
        }
        if (cg == 0 && r == 0) GG_STAMP(pass == 0 ? 2 : 5);  // first round of entries loaded
        if (pass == 0) {
          if (cg == 0 && r == 0)
            runs = wave_has_runs((PACK ? kw[0] >> g.key_bits : kw[0]) & leaf_mask, have & 1u, lane);
          if (runs) {
#pragma unroll
            for (int q = 0; q < LEAF_MAXS; q++)
              if (r + q < T)  // uniform
                run_add<false>(lc, (PACK ? kw[q] >> g.key_bits : kw[q]) & leaf_mask, (have >> q) & 1u, lane);
          } else {
#pragma unroll
            for (int q = 0; q < LEAF_MAXS; q++)
              if ((have >> q) & 1u) atomicAdd(&lc[(PACK ? kw[q] >> g.key_bits : kw[q]) & leaf_mask], 1u);
          }
        } else {
          volatile uint32_t *cur = lc;
#pragma unroll
          for (int q = 0; q < LEAF_MAXS; q++) {
            if (r + q >= T) continue;  // uniform
            const bool valid = (have >> q) & 1u;
            const uint32_t d = valid ? (PACK ? kw[q] >> g.key_bits : kw[q]) & leaf_mask : 0u;
            const uint32_t pay = PACK ? kw[q] & pay_mask : pw[PACK ? 0 : q];
            uint64_t m = __ballot(valid);
            uint32_t rel = 0;
            if (g.rank_atomic) {
              if (runs) {
                rel = run_add<true>((uint32_t *)cur, d, valid, lane);
              } else if (valid) {
                rel = atomicAdd((uint32_t *)&cur[d], 1u);
              }
            } else {
              for (uint32_t bit = 0; bit < g.leaf; bit++) {  // match mask: same vertex within the wave
                const uint64_t bb = __ballot((d >> bit) & 1u);
                m &= ((d >> bit) & 1u) ? bb : ~bb;
              }
              if (valid) rel = cur[d] + __popcll(m & lane_lt);
            }
            if (valid) {
              if (staged) {
                stage[rel] = pay;
                if (with_pos) stage_e[rel] = ew[ROWID ? q : 0];
              } else {
                o_nbr[t0 + rel] = pay;
                if (dir) rrow[t0 + rel] = (uint32_t)vfirst + d;
                if (with_pos) epos[t0 + rel] = ew[ROWID ? q : 0];
              }
            }
            if (!g.rank_atomic) {
              __builtin_amdgcn_wave_barrier();
              if (valid && (m & lane_lt) == 0) atomicAdd((uint32_t *)&cur[d], (uint32_t)__popcll(m));
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
      }
    }
    if (pass == 0) {
      // row offsets of the leaf's vertices (lane d = vertex d); lc becomes the relative cursor
      __builtin_amdgcn_wave_barrier();
      const uint32_t cnt = (uint32_t)lane < leafW ? lc[lane] : 0u;
      const uint32_t incl = wave_scan_incl(cnt);  // (DPP moves, gg_internal.h)
      if ((uint32_t)lane < leafW) {
        const uint64_t vtx = vfirst + (uint32_t)lane;
        if (vtx <= V) o_off[vtx] = t0 + incl - cnt;  // vtx == V: the closing offset
      }
      __builtin_amdgcn_wave_barrier();
      lc[lane] = incl - cnt;
      __builtin_amdgcn_wave_barrier();
      incl_keep = incl;  // (single: the entries are still in registers, pass 1 skips its loads)
      GG_STAMP(3);  // counted, row offsets written
    }
  }
  GG_STAMP(6);  // ranked (and staged or written)
  if (staged) {
    // the vertex of staged slot x is the number of runs that end at or before x
    for (uint32_t x0 = 0; x0 < n; x0 += 64) {
      const uint32_t x = x0 + lane;
      if (dir) {
        uint32_t d = 0;  // binary search: the first vertex whose run ends after x
        for (uint32_t step = leafW >> 1; step; step >>= 1)
          if (x >= (uint32_t)__shfl(incl_keep, (int)(d + step - 1), 64)) d += step;
        if (x < n) rrow[t0 + x] = (uint32_t)vfirst + d;
      }
      if (x < n) {
        o_nbr[t0 + x] = stage[x];
        if (with_pos) epos[t0 + x] = stage_e[x];
      }
    }
  }
  GG_STAMP(7);  // rows written
}

// One wave per leaf (bucket i, sub-bucket s); LEAF_WAVES leaves per workgroup.
template <bool PACK, bool ROWID>
__global__ __launch_bounds__(LEAF_WAVES * 64) void k_leaf_rows(
    const uint32_t *__restrict__ buf_f, const uint32_t *__restrict__ buf_r, const uint32_t *__restrict__ epos_f,
    const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ cstart, const uint32_t *__restrict__ offs,
    const uint32_t *__restrict__ substart, FastGeom g, uint64_t V, uint32_t *__restrict__ off,
    uint32_t *__restrict__ nbr, uint32_t *__restrict__ epos, uint32_t *__restrict__ roff,
    uint32_t *__restrict__ rnbr, uint32_t *__restrict__ rrow) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t unit = blockIdx.x * LEAF_WAVES + (threadIdx.x >> 6);  // waves are independent from here on
  if (unit >= (2u << (g.hb + g.sub))) return;
  leaf_unit<PACK, ROWID>(unit, lds, buf_f, buf_r, epos_f, bstart, cstart, offs, substart, g, V, off, nbr, epos, roff,
                         rnbr, rrow);
}

// ---- B, second form: chunks sorted by VERTEX, rows by address arithmetic -------------------------------------------
// For graphs of up to 2^20 vertices a bucket holds at most 2^10 = 1024 vertices, and a wave's row of LDS cursors can
// have one cursor per vertex: the chunk sort then orders a chunk by the whole bucket-local vertex number (not by a
// 6- or 7-bit sub-bucket), its offset row says where every vertex's entries start inside the chunk, and nothing is
// left to rank afterwards.  The rows step k_vrows therefore has no counting pass and no LDS atomics: an entry's
// final position is  off[v] + (entries of v in the chunks before) + (its index inside the chunk's segment of v),
// all of it known from the offset rows; one pass, one LDS read per entry for the per-(chunk, vertex) term.
//   k_vsort_pipe  persistent workgroups, next chunk's loads in flight (as k_sub_sort_pipe); cursors of up to 1024
//                 keys: every thread owns keys t and t + 512, two scans over the workgroup
//   k_vgroups     per bucket: where the rows of every group of VG vertices start (sums over the chunks' rows)
//   k_vrows       one wave per group of VG consecutive vertices: the group's row offsets (its piece of the CSR's
//                 offset array), then the group's run of every chunk copied to its rows
// Measured at SF100 against the sub-bucket form (k_sub_sort_pipe + k_sub_totals + k_leaf_rows): DESIGN.md §4.1.
#ifndef GG_FB_VLOW
#define GG_FB_VLOW 10  // vertices per bucket (log2) in this form: a wave's cursor row is 4 << GG_FB_VLOW bytes of LDS
#endif
constexpr int FB_VLOW = GG_FB_VLOW;
#ifndef GG_FB_VG
#define GG_FB_VG 16    // vertices per wave of k_vrows (<= 32)
#endif
constexpr int FB_VG = GG_FB_VG;
static_assert(FB_VG <= 32 && (1 << FB_VLOW) / FB_VG <= 64, "k_vgroups: one lane per group of a bucket");
#ifndef GG_FB_VCG
#define GG_FB_VCG 16
#endif
constexpr int FB_VCG = GG_FB_VCG;  // chunks a wave of k_vrows takes at a time (lane c holds chunk c's run; a multiple of 8)
#ifndef GG_FB_VSTEPS
#define GG_FB_VSTEPS 16  // 64-entry loads a wave of k_vrows keeps in flight
#endif
constexpr int FB_VSTEPS = GG_FB_VSTEPS;
#ifndef GG_FB_ALIGNW
#define GG_FB_ALIGNW 1  // k_vsort_pipe / k_vrows write their staged entries from a 128-byte line on
#endif

__global__ __launch_bounds__(FB_THREADS) void k_vsort_pipe(uint32_t *__restrict__ buf_f, uint32_t *__restrict__ buf_r,
                                                           const uint32_t *__restrict__ cstart,
                                                           const uint4 *__restrict__ part_of, FastGeom g,
                                                           uint16_t *__restrict__ offs /* [chunk][nk + 1], <= 8192 */) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t s_n, s_w[2][FB_WAVES];
  const uint32_t nb = 1u << g.hb, nk = 1u << g.low, G = gridDim.x;
  uint32_t p = blockIdx.x;
  uint4 chunk = part_of[p];  // (entries past the last chunk are allocated, not meaningful)
  const uint32_t nchunks = cstart[2 * nb];
  if (p >= nchunks) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *hw = lds;                  // [FB_WAVES][nk] per-wave counts -> cursors (staged slot)
  uint32_t *xw = hw + FB_WAVES * nk;   // staged words
  uint32_t *myh = hw + wave * nk;
  const uint32_t mine = (uint32_t)wave * FB_WTILE + lane;  // this lane's first entry inside a chunk
  uint32_t w[FB_ITEMS], wn[FB_ITEMS];
  {
    const uint32_t *__restrict__ src = (chunk.z / nb ? buf_r : buf_f) + chunk.x;
    const uint32_t len = chunk.y - chunk.x;
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++) w[it] = mine + it * 64 < len ? ld_stream(src + mine + it * 64) : 0u;
  }
  uint4 chunk_n = p + G < nchunks ? part_of[p + G] : make_uint4(0, 0, 0, 0);
  for (;;) {
    const bool has_next = p + G < nchunks;  // uniform
    uint4 chunk_nn = make_uint4(0, 0, 0, 0);
    if (has_next) {
      const uint32_t *__restrict__ src = (chunk_n.z / nb ? buf_r : buf_f) + chunk_n.x;
      const uint32_t len = chunk_n.y - chunk_n.x;
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) wn[it] = mine + it * 64 < len ? ld_stream(src + mine + it * 64) : 0u;
      if (p + 2 * G < nchunks) chunk_nn = part_of[p + 2 * G];
    }
    uint32_t *__restrict__ buf = (chunk.z / nb ? buf_r : buf_f) + chunk.x;
    const uint32_t len = chunk.y - chunk.x;
    for (uint32_t k = lane; k < nk; k += 64) myh[k] = 0;
    __builtin_amdgcn_wave_barrier();
    const bool runs = wave_has_runs(w[0] >> g.key_bits, mine < len, lane);  // (see k_partition_dual)
    if (runs) {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) run_add<false>(myh, w[it] >> g.key_bits, mine + it * 64 < len, lane);
    } else {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++)
        if (mine + it * 64 < len) atomicAdd(&myh[w[it] >> g.key_bits], 1u);
    }
    __syncthreads();
    {
      // cursors: thread t owns keys t and t + FB_THREADS (bank-conflict-free columns of hw); positions are prefix sums
      // in key order, so two scans over the workgroup: keys [0, 512), then [512, 1024) on top of the first total
      uint32_t cw[2][FB_WAVES], tot[2] = {0, 0}, incl[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t k = (uint32_t)h * FB_THREADS + threadIdx.x;
#pragma unroll
        for (int q = 0; q < FB_WAVES; q++) {
          cw[h][q] = k < nk ? hw[q * nk + k] : 0u;
          tot[h] += cw[h][q];
        }
        incl[h] = wave_scan_incl(tot[h]);  // (DPP moves, gg_internal.h)
        if (lane == 63) s_w[h][wave] = incl[h];
      }
      __syncthreads();
      uint32_t before[2] = {0, 0}, all[2] = {0, 0};
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int q = 0; q < FB_WAVES; q++) {
          const uint32_t sv = s_w[h][q];
          if (q < wave) before[h] += sv;
          all[h] += sv;
        }
      uint16_t *__restrict__ row = offs + (uint64_t)p * g.ss;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t k = (uint32_t)h * FB_THREADS + threadIdx.x;
        uint32_t run = (h ? all[0] : 0u) + before[h] + incl[h] - tot[h];
        if (k < nk) {
          row[k] = (uint16_t)run;
#pragma unroll
          for (int q = 0; q < FB_WAVES; q++) {
            hw[q * nk + k] = run;
            run += cw[h][q];
          }
        }
      }
      if (threadIdx.x == 0) {
        row[nk] = (uint16_t)(all[0] + all[1]);
        s_n = all[0] + all[1];
      }
    }
    __syncthreads();
    if (runs) {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) {
        const bool valid = mine + it * 64 < len;
        const uint32_t pos = run_add<true>(myh, w[it] >> g.key_bits, valid, lane);
        if (valid) xw[pos] = w[it];
      }
    } else {
#pragma unroll
      for (int it = 0; it < FB_ITEMS; it++) {  // ranks straight from the wave's cursors (lane-ordered ds_add_rtn)
        if (mine + it * 64 < len) {
          const uint32_t pos = atomicAdd(&myh[w[it] >> g.key_bits], 1u);
          xw[pos] = w[it];
        }
      }
    }
    __syncthreads();
    const uint32_t n = s_n;
    {
      // written back from the 128-byte line the chunk starts in (chunks start wherever their bucket does)
      const uint32_t hd = GG_FB_ALIGNW ? (uint32_t)((reinterpret_cast<uintptr_t>(buf) >> 2) & 31u) : 0u;
#pragma unroll
      for (int it = 0; it <= FB_ITEMS; it++) {
        const uint32_t sidx = (uint32_t)it * FB_THREADS + threadIdx.x - hd;
        if (sidx < n) buf[sidx] = xw[sidx];
      }
    }
    if (!has_next) break;
    __syncthreads();  // the stage and the cursors are reused
#pragma unroll
    for (int it = 0; it < FB_ITEMS; it++) w[it] = wn[it];
    chunk = chunk_n;
    chunk_n = chunk_nn;
    p += G;
  }
}

// where the rows of every group of VG vertices start: one wave per bucket, lane = group (at most 64 groups)
__global__ __launch_bounds__(64) void k_vgroups(const uint16_t *__restrict__ offs, const uint32_t *__restrict__ bstart,
                                                const uint32_t *__restrict__ cstart, FastGeom g,
                                                uint32_t *__restrict__ gstart /* [2 nb][64] */) {
  const uint32_t nb = 1u << g.hb, nk = 1u << g.low, i = blockIdx.x, dir = i / nb, j = i % nb;
  const uint32_t vg = nk < (uint32_t)FB_VG ? nk : (uint32_t)FB_VG, groups = nk / vg;
  const int lane = threadIdx.x;
  const uint32_t p0 = cstart[i], p1 = cstart[i + 1], k0 = (uint32_t)lane * vg;
  uint32_t tot = 0;
  if ((uint32_t)lane < groups)
    for (uint32_t p = p0; p < p1; p++) tot += (uint32_t)offs[(uint64_t)p * g.ss + k0 + vg] - offs[(uint64_t)p * g.ss + k0];
  const uint32_t incl = wave_scan_incl(tot);  // (DPP moves, gg_internal.h)
  gstart[(uint64_t)i * 64 + lane] = bstart[dir * (nb + 1) + j] + incl - tot;
}

// One wave per (direction, bucket, group of VG vertices).  Chunks are taken FB_VCG at a time: for chunk c the lanes
// d <= VG read the chunk's offsets of the group's vertices (one coalesced load), lane d keeps the running row
// position of vertex d, and LDS gets   delta[c][d] = (row position of d's first entry in chunk c) - (index of that
// entry inside the group's run of chunk c),   so an entry at index x of the run goes to delta[c][d] + x.
// The group's rows are one contiguous piece of the output: when it fits the wave's LDS stage the entries are placed
// there (plain LDS stores: scattered 4-byte GLOBAL stores cost a memory-pipeline slot per lane) and written out as
// whole lines; larger groups store straight to their rows.  (Measured at SF100, 240 us as it stands: reading the runs
// of a large group once per stage-sized window instead of storing directly 307 us; four entries per lane and step
// with 16-byte loads — runs start at any 4-byte boundary — 358 us.)
#ifndef GG_FB_VCAP
#define GG_FB_VCAP 1536  // entries of a wave's stage in k_vrows
#endif
__global__ __launch_bounds__(64) void k_vrows(const uint32_t *__restrict__ buf_f, const uint32_t *__restrict__ buf_r,
                                              const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ cstart,
                                              const uint16_t *__restrict__ offs, const uint32_t *__restrict__ gstart,
                                              FastGeom g, uint64_t V, uint32_t *__restrict__ off,
                                              uint32_t *__restrict__ nbr, uint32_t *__restrict__ roff,
                                              uint32_t *__restrict__ rnbr, uint32_t *__restrict__ rrow) {
  __shared__ uint32_t s_delta[FB_VCG * FB_VG];
  __shared__ uint32_t s_run[2 * FB_VCG];  // [c]: first entry of the run inside chunk c, [FB_VCG + c]: its length
  __shared__ uint32_t s_stage[GG_FB_VCAP];
  const uint32_t nb = 1u << g.hb, nk = 1u << g.low;
  const uint32_t vg = nk < (uint32_t)FB_VG ? nk : (uint32_t)FB_VG, groups = nk / vg;  // (powers of two)
  const uint32_t unit = blockIdx.x, i = unit / groups, gi = unit % groups, k0 = gi * vg, dir = i / nb, j = i % nb;
  const int lane = threadIdx.x;
  const uint32_t *__restrict__ buf = dir ? buf_r : buf_f;
  uint32_t *__restrict__ o_nbr = dir ? rnbr : nbr;
  const uint64_t v0 = ((uint64_t)j << g.low) + k0;  // first vertex of the group
  if (v0 >= V) return;                              // (the table ends before this group: uniform)
  const uint32_t b0 = bstart[dir * (nb + 1) + j], p0 = cstart[i], nch = cstart[i + 1] - p0;
  const uint32_t gs = gstart[(uint64_t)i * 64 + gi];
  const uint32_t pay_mask = (1u << g.key_bits) - 1u;  // (this form packs: key_bits + low <= 32)
  const uint32_t keep_mask = (vg << g.key_bits) - 1u;   // payload and the vertex inside the group
  // Pass over the offset rows of ALL the bucket's chunks: the vertices' degrees (sum over the chunks), and with them the
  // row offsets — this wave's piece of the CSR's offset array.  The rows of the first FB_VCG chunks (normally all
  // of them) are parked in the LDS tables as they stand, so the placing pass below does not load them again.
  uint32_t deg = 0;
#pragma unroll 1
  for (uint32_t c0 = 0; c0 < nch; c0 += 8) {
    uint32_t sv[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {  // eight loads in flight
      sv[q] = 0;
      if (c0 + q < nch && (uint32_t)lane <= vg) sv[q] = offs[(uint64_t)(p0 + c0 + q) * g.ss + k0 + lane];
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t c = c0 + q;
      if (c >= nch) continue;  // uniform
      const uint32_t nxt = (uint32_t)__shfl_down((int)sv[q], 1, 64);  // (by the whole wave: lane vg feeds lane vg - 1)
      if ((uint32_t)lane < vg) deg += nxt - sv[q];
      if (c < (uint32_t)FB_VCG) {
        if ((uint32_t)lane < vg) s_delta[c * vg + lane] = sv[q];
        if (lane == 0) s_run[c] = sv[q];
        if ((uint32_t)lane == vg) s_run[FB_VCG + c] = sv[q];  // (the END of the run for now)
      }
    }
  }
  const uint32_t dincl = wave_scan_incl(deg);  // (DPP moves, gg_internal.h)
  uint32_t base = gs + dincl - deg;            // lane d < vg: row position of vertex d's next entry; lane vg: the end
  {
    const uint64_t vv = v0 + (uint32_t)lane;
    if ((uint32_t)lane <= vg && vv <= V) (dir ? roff : off)[vv] = base;  // (lane vg of a group writes the next group's
  }                                                                       // first offset too: the same value)
  const uint32_t t0 = gs;
  const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)dincl, 63);  // entries of the group
  const bool staged = n <= (uint32_t)GG_FB_VCAP;  // uniform
  if (staged) base -= t0;                         // positions inside the stage
#pragma unroll 1
  for (uint32_t cg = 0; cg < nch; cg += FB_VCG) {
    const uint32_t ncg = nch - cg < (uint32_t)FB_VCG ? nch - cg : (uint32_t)FB_VCG;
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (uint32_t c = 0; c < ncg; c++) {  // chunk c of the round, lane d < vg: where vertex d starts in it
      uint32_t sv = 0, first, end;
      if (cg == 0) {  // parked above
        if ((uint32_t)lane < vg) sv = s_delta[c * vg + lane];
        first = s_run[c];
        end = s_run[FB_VCG + c];
      } else {
        if ((uint32_t)lane <= vg) sv = offs[(uint64_t)(p0 + cg + c) * g.ss + k0 + lane];
        first = (uint32_t)__builtin_amdgcn_readlane((int)sv, 0);
        end = (uint32_t)__builtin_amdgcn_readlane((int)sv, (int)vg);
      }
      uint32_t nxt = (uint32_t)__shfl_down((int)sv, 1, 64);
      if ((uint32_t)lane + 1 == vg) nxt = end;
      if ((uint32_t)lane < vg) {
        s_delta[c * vg + lane] = base - (sv - first);
        base += nxt - sv;
      }
      if (lane == 0) {
        s_run[c] = first;
        s_run[FB_VCG + c] = end - first;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // lane c holds chunk c's run; a run is walked in 64-entry steps
    uint32_t so = 0, len = 0;
    if ((uint32_t)lane < ncg) {
      so = s_run[lane];
      len = s_run[FB_VCG + lane];
    }
    const uint32_t src = b0 + (cg + lane) * FB_TILE + so;  // position of the run's first entry
    const uint32_t st_c = (len + 63) / 64;
    const uint32_t sincl = wave_scan_incl(st_c);  // (DPP moves, gg_internal.h)
    const uint32_t sexcl = sincl - st_c;
    const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)sincl, 63);
#pragma unroll 1
    for (uint32_t r = 0; r < T; r += FB_VSTEPS) {
      uint32_t kw[FB_VSTEPS], kx[FB_VSTEPS];
      uint32_t have = 0;
#pragma unroll
      for (int q = 0; q < FB_VSTEPS; q++) {  // all loads of the round issue back to back
        const uint32_t t = r + q;
        kw[q] = 0;
        kx[q] = 0;
        if (t < T) {  // uniform
          const int cc = __popcll(__ballot(sincl <= t));  // the run this step belongs to (< 64: t < T)
          const uint32_t kidx = (t - (uint32_t)__builtin_amdgcn_readlane((int)sexcl, cc)) * 64 + lane;
          const uint32_t run_src = (uint32_t)__builtin_amdgcn_readlane((int)src, cc);
          const uint32_t run_len = (uint32_t)__builtin_amdgcn_readlane((int)len, cc);
          if (kidx < run_len) {
            have |= 1u << q;
            kw[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(buf + run_src), 0, (int)(run_len * 4u),
                                                  0x00020000),
                kidx * 4u, 0, 0);
            kx[q] = kidx | ((uint32_t)cc << 24);  // index inside the run (< 8192), run number
          }
        }
      }
#pragma unroll
      for (int q = 0; q < FB_VSTEPS; q++) {
        if (r + q >= T) continue;  // uniform
        if ((have >> q) & 1u) {
          const uint32_t d = (kw[q] >> g.key_bits) & (vg - 1u);
          const uint32_t pos = s_delta[(kx[q] >> 24) * vg + d] + (kx[q] & 0xFFFFFFu);
          if (staged) {
            s_stage[pos] = kw[q] & keep_mask;
          } else {
            o_nbr[pos] = kw[q] & pay_mask;
            if (dir) rrow[pos] = (uint32_t)v0 + d;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the tables are rewritten by the next round of chunks
  }
  if (staged) {
    // written from the 128-byte line the group's first entry lies in: every store instruction covers whole lines
    // (the first and the last are shared with the neighbouring groups)
    const uint32_t hd = GG_FB_ALIGNW ? (t0 & 31u) : 0u;
    for (uint32_t x0 = 0; x0 < n + hd; x0 += 64) {
      const uint32_t x = x0 + lane - hd;  // (wraps below zero for the lanes before the first entry)
      if (x < n) {
        const uint32_t sw = s_stage[x];
        o_nbr[t0 + x] = sw & pay_mask;
        if (dir) rrow[t0 + x] = (uint32_t)v0 + (sw >> g.key_bits);
      }
    }
  }
}

static int bits_of(uint64_t v) {  // bits needed for values 0..v
  int b = 0;
  while (b < 64 && (v >> b)) b++;
  return b;
}

int csr_build_fast(gg_ctx *ctx, gg_csr *csr, BuildStatus *st, int *taken) {
  *taken = 0;
  const uint64_t V = csr->V, E = csr->E_cap;
  if (E == 0 || V == 0 || E >= 0xFFFFFFFFull) return GG_OK;  // positions are u32
  FastGeom g;
  int kb = bits_of(V - 1);
  if (kb < 1) kb = 1;
  // Geometry: key = (bucket: hb bits)(sub-bucket: sub bits)(leaf vertex: leaf bits).  A leaf is finished by one
  // wave, so it should hold some hundreds of entries: 2^(hb + sub) ~ entries / 512..1024, where a shard expects about
  // half of its local rows per direction.  leaf <= 6 bits (one lane per vertex), sub <= FB_MAX_SUB, hb <= FB_MAX_HB.
  const uint64_t expect = csr->n_parts > 1 ? E / 2 : E;
  int lb = bits_of(expect / GG_FB_LEAF_TARGET);               // log2 of the number of leaves wanted (600-1200 entries each)
  if (lb < kb - 6) lb = kb - 6;                  // leaf <= 6 bits
  if (lb > kb) lb = kb;                          // at most one leaf per vertex
  if (lb < 0) lb = 0;
  int sub = lb < FB_MAX_SUB ? lb : FB_MAX_SUB;
  int hb = lb - sub;
  if (hb > FB_MAX_HB) return GG_OK;              // more than 2^22 vertices: the multi-pass build
  int low = kb - hb;
  GG_TRY(lds_order_ok(ctx, &g.rank_atomic));
  // up to 2^(FB_MAX_HB + FB_VLOW) vertices, packed words, no edge positions, cursor ranks: the chunks are sorted by
  // vertex and the rows step only copies (k_vsort_pipe / k_vtotals / k_vrows)
  const bool want_rowid = ctx->keep_edge_rowid && csr->n_parts <= 1;
  const bool vsort = 1 && g.rank_atomic && !want_rowid && kb <= FB_MAX_HB + FB_VLOW && kb <= 20;
  if (vsort) {
    low = kb < FB_VLOW ? kb : FB_VLOW;
    hb = kb - low;
    sub = low;
  }
  g.key_bits = (uint32_t)kb;
  g.low = (uint32_t)low;
  g.hb = (uint32_t)hb;
  g.pack = (low + kb <= 32) ? 1u : 0u;
  g.sub = (uint32_t)sub;
  g.leaf = (uint32_t)(low - sub);
  g.ss = vsort ? (1u << low) + 1u : sub > 6 ? 129u : 65u;
  g.part = (uint32_t)csr->part;
  g.n_parts = (uint32_t)csr->n_parts;
  *taken = 1;
  const uint32_t nb = 1u << hb;
  const bool rowid = ctx->keep_edge_rowid && csr->n_parts <= 1;  // a shard only serves 2-hop counting and BFS: no rowids
  const uint64_t nblocks64 = (E + FB_TILE - 1) / FB_TILE;
  const unsigned nblocks = (unsigned)nblocks64;

  // ---- dictionary (device-side choice) ---------------------------------------------------------------------
  DirectMap *dm = nullptr;
  uint32_t *dir = nullptr;
  unsigned long long *tab = nullptr;
  const uint32_t idx_bits = (uint32_t)bits_of(V - 1);
  // slot pairs of the packed dictionary: the fewest with a load factor <= GG_FB_LOAD_PCT (not a power of two: the
  // probe rate follows the table's size, gg_dict.h), a multiple of 64, at least 512
  uint64_t npairs = ((V * 100 + 2 * GG_FB_LOAD_PCT - 1) / (2 * GG_FB_LOAD_PCT) + 63) / 64 * 64;
  if (npairs < 512) npairs = 512;
  if (csr->n_parts > 1) {
    // a shard looks up a fraction of the rows and is bound by the latency of its probe chains, not by their rate:
    // the next power of two (load 0.43 instead of 0.5 at SF100) is 22 us faster for one shard of eight (173 against
    // 195 us), while the whole build gains 12 us from the smaller table (profiles/r03_ab_dict_load.txt)
    uint64_t p2 = 512;
    while (p2 < npairs) p2 <<= 1;
    npairs = p2;
  }
  dm = reinterpret_cast<DirectMap *>(reinterpret_cast<unsigned long long *>(st) + 8);  // seeded with st (csr_build_impl)
  GG_TRY(ctx->dev_alloc((void **)&dir, DIRECT_MAX_RANGE * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&tab, 2 * npairs * sizeof(unsigned long long)));
  // column totals of the counter matrix (zeroed by k_dict_init): allocated here, used by the column kernels below
  const uint32_t ncol = 2 * nb;
  uint32_t gsz = 16;  // tiles per group of the column kernels; at most 512 groups for the single-workgroup step
  while ((nblocks64 + gsz - 1) / gsz > 512) gsz *= 2;
  const uint32_t ngroups = (uint32_t)((nblocks64 + gsz - 1) / gsz);
  uint32_t *partial = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&partial, ((uint64_t)ngroups + 1) * ncol * sizeof(uint32_t)));
  uint32_t *coltot = partial + (uint64_t)ngroups * ncol;  // column totals, then column (= bucket) starts
  {
    uint64_t span = csr->ht_cap > 2 * npairs ? csr->ht_cap : 2 * npairs;
    if (span < DIRECT_MAX_RANGE) span = DIRECT_MAX_RANGE;
    if (span < V) span = V;
    GG_LAUNCH(ctx, "dict_init", k_dict_init, dim3((unsigned)((span + 255) / 256)), dim3(256), 0,
              (const int64_t *)ctx->c_vid.dev, V, csr->ht,
              csr->ht_cap, tab, 2 * npairs, dir, dm, st, (uint32_t)csr->part, (uint32_t)csr->n_parts, csr->vid, coltot,
              ncol);
    GG_LAUNCH(ctx, "dict_insert", k_dict_insert, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, csr->vid, V, csr->ht,
              csr->ht_cap, tab, dir, dm, idx_bits, (uint32_t)npairs, st);
    GG_LAUNCH(ctx, "dict_wide", k_dict_wide, dim3((unsigned)((V + 255) / 256 < 512 ? (V + 255) / 256 : 512)), dim3(256), 0,
              csr->vid, V, csr->ht,
              csr->ht_cap, (const DirectMap *)dm, st);
  }

  // ---- D ----------------------------------------------------------------------------------------------------
  u32x2 *pairs = nullptr;
  uint32_t *counts = nullptr, *bstart = nullptr;
  uint64_t *total = nullptr;
  const uint64_t ncount = 2ull * nb * nblocks64;
  GG_TRY(ctx->dev_alloc((void **)&pairs, E * sizeof(u32x2)));
  GG_TRY(ctx->dev_alloc((void **)&counts, ncount * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&bstart, 2 * (nb + 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&total, sizeof(uint64_t)));
  GG_LAUNCH(ctx, "densify_pairs", k_densify_pairs<false>, dim3(nblocks), dim3(FB_THREADS), 0, ctx->c_src.dev, ctx->c_dst.dev, E,
            csr->ht, csr->ht_cap, st, (const uint32_t *)dir, (const unsigned long long *)tab, (const DirectMap *)dm,
            pairs, g, nblocks64, counts);
  const uint64_t pmax = 2 * (nblocks64 + nb);  // chunks: every bucket may end in a partial one
  uint4 *part_of = nullptr;
  uint32_t *cstart = nullptr, *offs = nullptr, *substart = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&cstart, (2 * nb + 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&part_of, pmax * sizeof(uint4)));
  GG_TRY(ctx->dev_alloc((void **)&offs, pmax * g.ss * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&substart, (uint64_t)2 * nb * (g.ss < 64 ? 64 : g.ss) * sizeof(uint32_t)));
  GG_LAUNCH(ctx, "col_partial", k_col_partial, dim3(ngroups), dim3(1024), 0, (const uint32_t *)counts, nblocks64, ncol, gsz,
            partial, coltot);
  GG_LAUNCH(ctx, "col_scan", k_col_scan, dim3(1), dim3(1024), 0, coltot, nb, bstart, cstart, part_of, st);
  if (!rowid) {
    // every status word is final now (duplicate ids: k_dict_insert; dictionary mode: k_dict_wide; kept entries: the
    // column scan) and no kernel below reports an error: the host gets its copy here and waits for THIS, with two
    // thirds of the build still queued — the caller's next launches are then behind them before the device runs dry
    // (the build used to end in a stream synchronisation: ~15 us of idle device per step)
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, st, sizeof(BuildStatus), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipEventRecord(ctx->status_ev, ctx->stream));
    ctx->status_early = true;
  }
  GG_LAUNCH(ctx, "col_apply", k_col_apply, dim3(ngroups), dim3(1024), 0, counts, nblocks64, ncol, gsz,
            (const uint32_t *)partial, (const uint32_t *)coltot);

  // ---- A ----------------------------------------------------------------------------------------------------
  const size_t words = g.pack ? 1 : 2;
  uint32_t *part_f = nullptr, *part_r = nullptr, *epos_f = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&part_f, E * words * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&part_r, E * words * sizeof(uint32_t)));
  if (rowid) GG_TRY(ctx->dev_alloc((void **)&epos_f, E * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->roff, (V + 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->rnbr, E * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->rrow, E * sizeof(uint32_t)));
  const size_t lds_a = ((size_t)FB_TILE * (words + (rowid ? 1 : 0)) + (size_t)(FB_WAVES + 2) * nb + FB_WAVES + 2) *
                           sizeof(uint32_t) +
                       (size_t)FB_TILE * sizeof(uint16_t) + 16 + (0 ? (size_t)FB_WAVES * nb * 8 : 0);
  const unsigned grid_a = (unsigned)(((nblocks64 + 7) / 8) * 8);
#define GG_FB_LAUNCH_A(P, R)                                                                                        \
  GG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_partition_dual<P, R>),                                 \
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));                                \
  GG_LAUNCH(ctx, "partition_dual", (k_partition_dual<P, R>), dim3(grid_a), dim3(FB_THREADS), lds_a,                  \
            (const u32x2 *)pairs, E, g, nblocks64, (const uint32_t *)counts, (const BuildStatus *)st, part_f, part_r, \
            epos_f)
  if (g.pack && rowid) {
    GG_FB_LAUNCH_A(true, true);
  } else if (g.pack) {
    GG_FB_LAUNCH_A(true, false);
  } else if (rowid) {
    GG_FB_LAUNCH_A(false, true);
  } else {
    GG_FB_LAUNCH_A(false, false);
  }
#undef GG_FB_LAUNCH_A

  // ---- B ----------------------------------------------------------------------------------------------------
  const size_t lds_s = (size_t)((FB_WAVES + 1) * (g.ss - 1) + 1 + FB_TILE * (words + (rowid ? 1 : 0))) * sizeof(uint32_t);
  const size_t lds_l = (size_t)LEAF_WAVES * (64 + GG_FB_CAPW) * sizeof(uint32_t);
  const unsigned grid_l = (unsigned)((2ull * nb * (1u << g.sub) + LEAF_WAVES - 1) / LEAF_WAVES);
#define GG_FB_LAUNCH_B(P, R, SPL)                                                                                     \
  GG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sub_sort<P, R, SPL>),                                    \
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));                                  \
  GG_LAUNCH(ctx, "sub_sort", (k_sub_sort<P, R, SPL>), dim3((unsigned)pmax), dim3(FB_THREADS), lds_s, part_f, part_r,    \
            epos_f, (const uint32_t *)bstart, (const uint32_t *)cstart, (const uint4 *)part_of, g, offs);               \
  GG_LAUNCH(ctx, "sub_totals", k_sub_totals<SPL>, dim3(2 * nb), dim3(64), 0, (const uint32_t *)offs,                    \
            (const uint32_t *)bstart, (const uint32_t *)cstart, g, V, substart, csr->off, csr->roff);                   \
  GG_LAUNCH(ctx, "leaf_rows", (k_leaf_rows<P, R>), dim3(grid_l), dim3(LEAF_WAVES * 64), lds_l,                          \
            (const uint32_t *)part_f, (const uint32_t *)part_r, (const uint32_t *)epos_f, (const uint32_t *)bstart,     \
            (const uint32_t *)cstart, (const uint32_t *)offs, (const uint32_t *)substart, g, V, csr->off, csr->nbr,     \
            csr->epos, csr->roff, csr->rnbr, csr->rrow)
#define GG_FB_LAUNCH_B2(P, R)      \
  do {                             \
    if (g.sub > 6) {               \
      GG_FB_LAUNCH_B(P, R, 2);     \
    } else {                       \
      GG_FB_LAUNCH_B(P, R, 1);     \
    }                              \
  } while (0)
  if (vsort) {
    const size_t lds_v = ((size_t)FB_WAVES * (1u << low) + FB_TILE) * sizeof(uint32_t);
    const uint64_t resident = (uint64_t)ctx->num_cus * 2;
    const unsigned grid_p = (unsigned)(pmax < resident ? pmax : resident);
    GG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_vsort_pipe), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds_v));
    uint16_t *offs16 = reinterpret_cast<uint16_t *>(offs);  // (rows of nk + 1 16-bit offsets inside the same block)
    uint32_t *gstart = substart;                            // [2 nb][64]
    GG_LAUNCH(ctx, "sub_sort", k_vsort_pipe, dim3(grid_p), dim3(FB_THREADS), lds_v, part_f, part_r,
              (const uint32_t *)cstart, (const uint4 *)part_of, g, offs16);
    GG_LAUNCH(ctx, "sub_totals", k_vgroups, dim3(2 * nb), dim3(64), 0, (const uint16_t *)offs16,
              (const uint32_t *)bstart, (const uint32_t *)cstart, g, gstart);
    const uint32_t nk = 1u << low, vg = nk < (uint32_t)FB_VG ? nk : (uint32_t)FB_VG;
    GG_LAUNCH(ctx, "leaf_rows", k_vrows, dim3((unsigned)(2ull * nb * (nk / vg))), dim3(64), 0, (const uint32_t *)part_f,
              (const uint32_t *)part_r, (const uint32_t *)bstart, (const uint32_t *)cstart, (const uint16_t *)offs16,
              (const uint32_t *)gstart, g, V, csr->off, csr->nbr, csr->roff, csr->rnbr, csr->rrow);
  } else if (g.pack && rowid) {
    GG_FB_LAUNCH_B2(true, true);
  } else if (g.pack && g.rank_atomic && GG_FB_SUBPIPE) {
    const uint64_t resident = (uint64_t)ctx->num_cus * GG_FB_SUBPIPE;
    const unsigned grid_p = (unsigned)(pmax < resident ? pmax : resident);
    if (g.sub > 6) {
      GG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sub_sort_pipe<2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
      GG_LAUNCH(ctx, "sub_sort", k_sub_sort_pipe<2>, dim3(grid_p), dim3(FB_THREADS), lds_s, part_f, part_r,
                (const uint32_t *)cstart, (const uint4 *)part_of, g, offs);
      GG_LAUNCH(ctx, "sub_totals", k_sub_totals<2>, dim3(2 * nb), dim3(64), 0, (const uint32_t *)offs,
                (const uint32_t *)bstart, (const uint32_t *)cstart, g, V, substart, csr->off, csr->roff);
    } else {
      GG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sub_sort_pipe<1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
      GG_LAUNCH(ctx, "sub_sort", k_sub_sort_pipe<1>, dim3(grid_p), dim3(FB_THREADS), lds_s, part_f, part_r,
                (const uint32_t *)cstart, (const uint4 *)part_of, g, offs);
      GG_LAUNCH(ctx, "sub_totals", k_sub_totals<1>, dim3(2 * nb), dim3(64), 0, (const uint32_t *)offs,
                (const uint32_t *)bstart, (const uint32_t *)cstart, g, V, substart, csr->off, csr->roff);
    }
    GG_LAUNCH(ctx, "leaf_rows", (k_leaf_rows<true, false>), dim3(grid_l), dim3(LEAF_WAVES * 64), lds_l,
              (const uint32_t *)part_f, (const uint32_t *)part_r, (const uint32_t *)epos_f, (const uint32_t *)bstart,
              (const uint32_t *)cstart, (const uint32_t *)offs, (const uint32_t *)substart, g, V, csr->off, csr->nbr,
              csr->epos, csr->roff, csr->rnbr, csr->rrow);
  } else if (g.pack) {
    GG_FB_LAUNCH_B2(true, false);
  } else if (rowid) {
    GG_FB_LAUNCH_B2(false, true);
  } else {
    GG_FB_LAUNCH_B2(false, false);
  }
#undef GG_FB_LAUNCH_B2
#undef GG_FB_LAUNCH_B

  for (void *p : {(void *)dir, (void *)tab, (void *)pairs, (void *)counts, (void *)bstart, (void *)total,
                  (void *)part_f, (void *)part_r, (void *)epos_f, (void *)cstart, (void *)part_of, (void *)offs,
                  (void *)substart, (void *)partial})
    ctx->dev_free(p);
  return GG_OK;
}

}  // namespace gg
