// gg_dict.h — id -> dense index dictionaries of the CSR build (device code shared by gg_csr.hip and
// gg_csr_fast.hip; every kernel is `static`, each translation unit carries its own copy).
//
// The reference switches its join to a perfect hash table — an array indexed by key - min — when the build
// keys are integers spanning at most 1 000 000 values (CheckForPerfectJoinOpt, src/execution/
// physical_plan/plan_comparison_join.cpp:35-107; PerfectHashJoinExecutor::BuildPerfectHashTable,
// src/execution/operator/join/perfect_hash_join_executor.cpp:24-65).  Same idea for the id -> dense index
// dictionary the edge densification probes twice per edge row, in three forms chosen ON THE DEVICE from the
// vertex ids' min/max (no host synchronisation):
//   DICT_DIRECT   ids span < 2^20 values: a uint32 array indexed by id - min (<= 4 MiB, stays in an XCD's L2)
//   DICT_PACKED8  ids spanning up to ~2^54: open addressing with 8-byte slots that hold the quotient of a
//                 bijective hash of id - min (exact), the pair displacement and the dense index — half the
//                 footprint of the general table, two slots per 16-byte probe
//   DICT_WIDE16   anything else: 16-byte slots {id, dense index} (gg_internal.h ht_lookup)
#pragma once
#include "gg_internal.h"

namespace gg {

static __global__ __launch_bounds__(256) void k_id_minmax(const int64_t *__restrict__ vid, uint64_t V,
                                                          DirectMap *__restrict__ dm) {
  long long lo = INT64_MAX, hi = INT64_MIN;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (uint64_t)gridDim.x * blockDim.x) {
    const long long x = vid[i];
    lo = x < lo ? x : lo;
    hi = x > hi ? x : hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  __shared__ long long s_lo[4], s_hi[4];
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // one pair of atomics per workgroup: same-address 64-bit atomics serialise
    for (int w = 1; w < 4; w++) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
    }
    atomicMin(&dm->min_id, lo);
    atomicMax(&dm->max_id, hi);
  }
}

// idx_bits = bits of V - 1; pairs = 16-byte slot pairs of the packed table (0: the caller has no packed table)
static __global__ void k_dict_decide(DirectMap *__restrict__ dm, uint64_t V, uint32_t idx_bits, uint32_t pairs) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const uint64_t span = (uint64_t)dm->max_id - (uint64_t)dm->min_id;  // exact in unsigned arithmetic
    const bool any = V > 0 && dm->max_id >= dm->min_id;
    const uint32_t span_bits = span ? 64u - (uint32_t)__clzll((long long)span) : 0u;
    unsigned long long mode = DICT_WIDE16;
    if (any && span < DIRECT_MAX_RANGE)
      mode = DICT_DIRECT;
    else if (any && pairs > 12u /* PK_MAX_DISP */ && span_bits > 31u - (uint32_t)__clz((int)pairs) && span_bits <= 57u &&
             span_bits - (31u - (uint32_t)__clz((int)pairs)) + 6u /* PK_DISP_BITS */ + idx_bits <= 63u)
      mode = DICT_PACKED8;
    dm->mode = mode;
    dm->enabled = mode == DICT_DIRECT ? 1ULL : 0ULL;
    dm->idx_bits = idx_bits;
    dm->span_bits = span_bits;
    dm->pairs = pairs;
  }
}

static __global__ __launch_bounds__(256) void k_direct_init(uint32_t *__restrict__ dir,
                                                            const DirectMap *__restrict__ dm) {
  if (!dm->enabled) return;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < DIRECT_MAX_RANGE) dir[i] = INVALID_U32;
}

static __global__ __launch_bounds__(256) void k_direct_fill(const int64_t *__restrict__ vid, uint64_t V,
                                                            uint32_t *__restrict__ dir,
                                                            const DirectMap *__restrict__ dm) {
  if (!dm->enabled) return;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < V) dir[(uint64_t)vid[i] - (uint64_t)dm->min_id] = (uint32_t)i;  // duplicate ids: the hash insert reports them
}

__device__ __forceinline__ uint32_t direct_lookup(const uint32_t *__restrict__ dir, uint64_t min_id, int64_t key) {
  const uint64_t off = (uint64_t)key - min_id;
  return off < DIRECT_MAX_RANGE ? dir[off] : INVALID_U32;
}

// ---- packed 8-byte slots --------------------------------------------------------------------------------
// Exact with 8 bytes per vertex for ids spanning up to ~2^54: quotienting.  rel = id - min has S = span_bits
// significant bits; pk_mix is a BIJECTION on S-bit values (odd multiplications mod 2^S and xor-shifts), so
// h = pk_mix(rel) identifies the id.  The table has G slot pairs (one aligned 16-byte granule each: a probe
// loads both slots with one dwordx4), G ANY number — the probe rate of the edge densification falls steadily
// with the table's size between one XCD's L2 (4 MB) and twice that (scripts/ubench_gather_sizes.hip: 546 us
// at 4 MB, 591 at 5.33, 650 at 6.4, 785 at 8), so the table is as small as the load factor allows, not the
// next power of two.  The home pair is floor(h G / 2^S), taken from the top 32 bits t of h as umulhi(t, G);
// the h that share a home differ in fewer than 2^(S - qf) places, qf = floor(log2 G), so the low S - qf bits
// of h (rem) tell them apart and only those are stored, next to the pair displacement from home
// (PK_DISP_BITS, linear probing over pairs) and the dense index:
//   slot = ((rem << PK_DISP_BITS | disp) << idx_bits) | idx,   bit 63 clear, empty = all ones.
// A lookup scans pairs from home and stops at the first empty slot.  If some key would be displaced further
// than the field allows, the insert kernel flips the mode back to DICT_WIDE16 (nothing has read it yet).
constexpr unsigned long long PK_EMPTY = ~0ULL;
constexpr uint32_t PK_DISP_BITS = 6;
#ifndef GG_PK_MAX_DISP
#define GG_PK_MAX_DISP 12
#endif
constexpr uint64_t PK_MAX_DISP = GG_PK_MAX_DISP;  // pairs a key may sit away from home before the table is given up

#ifndef GG_PK_MIX
#define GG_PK_MIX 1
#endif
__device__ __forceinline__ uint64_t pk_mix(uint64_t rel, uint32_t S) {  // 21 <= S <= 57
  const uint64_t mask = (1ULL << S) - 1ULL;
#if GG_PK_MIX == 1
  // Fibonacci hashing on S bits: ids that are (pieces of) arithmetic progressions — the usual shape of
  // database keys — spread almost evenly over the home pairs, far better than a random function would at a
  // load factor near 0.9; k_packed_insert gives the table up if some key is displaced more than PK_MAX_DISP
  return (rel * 0x9E3779B97F4A7C15ULL) & mask;
#else
  const uint32_t r = S >> 1;
  uint64_t x = rel;
  x ^= x >> r;
  x = (x * 0x9E3779B97F4A7C15ULL) & mask;
  x ^= x >> r;
  x = (x * 0xD6E8FEB86659FD93ULL) & mask;
  x ^= x >> r;
  return x;
#endif
}

struct PkGeom {  // uniform values of a packed table, read once per kernel
  uint64_t min_id, max_id;
  uint32_t S, G, qf, b;
  __device__ __forceinline__ void set(uint64_t min_id_, uint64_t max_id_, uint32_t S_, uint32_t G_, uint32_t b_) {
    min_id = min_id_;
    max_id = max_id_;
    S = S_;
    G = G_;
    qf = G_ ? 31u - (uint32_t)__clz((int)G_) : 0u;
    b = b_;
  }
  __device__ __forceinline__ void load(const DirectMap *dm) {
    set((uint64_t)dm->min_id, (uint64_t)dm->max_id, (uint32_t)dm->span_bits, (uint32_t)dm->pairs, (uint32_t)dm->idx_bits);
  }
  // home pair and the tag of displacement 0 for a key inside [min, max] (any other key: some pair of the table)
  __device__ __forceinline__ void locate(int64_t key, uint64_t *home, uint64_t *tag0) const {
    const uint64_t h = pk_mix((uint64_t)key - min_id, S);
    const uint32_t t = S >= 32u ? (uint32_t)(h >> (S - 32u)) : (uint32_t)(h << (32u - S));
    *home = __umulhi(t, G);
    *tag0 = (h & ((1ULL << (S - qf)) - 1ULL)) << PK_DISP_BITS;
  }
  __device__ __forceinline__ uint64_t pair_at(uint64_t home, uint64_t disp) const {  // (disp <= PK_MAX_DISP < G)
    const uint64_t g = home + disp;
    return g >= G ? g - G : g;
  }
};

// finish a packed lookup whose first probe (pair `home`, raw) is already loaded
__device__ __forceinline__ uint32_t packed_resolve(const unsigned long long *__restrict__ tab, const PkGeom &pk,
                                                   bool in_range, uint64_t home, uint64_t tag0, uint4 raw) {
  if (!in_range) return INVALID_U32;
  const unsigned long long idx_mask = (1ULL << pk.b) - 1ULL;
  for (uint64_t disp = 0;;) {
    const unsigned long long e0 = ((unsigned long long)raw.y << 32) | raw.x;
    const unsigned long long e1 = ((unsigned long long)raw.w << 32) | raw.z;
    if ((e0 >> pk.b) == (tag0 | disp)) return (uint32_t)(e0 & idx_mask);
    if (e0 == PK_EMPTY) return INVALID_U32;
    if ((e1 >> pk.b) == (tag0 | disp)) return (uint32_t)(e1 & idx_mask);
    if (e1 == PK_EMPTY) return INVALID_U32;
    if (++disp > PK_MAX_DISP) return INVALID_U32;
    raw = *reinterpret_cast<const uint4 *>(&tab[2 * pk.pair_at(home, disp)]);
  }
}

// ---- the bucketed build's dictionary in two launches (+ one that normally does nothing) -------------------------
// k_dict_init    clears all three candidate tables and takes the ids' min / max in the same pass
// k_dict_insert  every thread derives the mode from min / max (thread 0 publishes it), inserts its vertex into the
//                table of that mode, reports duplicate ids
// k_dict_wide    only if a packed insert had to give up (mode flipped to DICT_WIDE16): fills the 16-byte table
__device__ __forceinline__ bool packed_fits(uint32_t span_bits, uint32_t pairs, uint32_t idx_bits) {
  if (pairs <= PK_MAX_DISP) return false;  // (also: no packed table)
  const uint32_t qf = 31u - (uint32_t)__clz((int)pairs);
  return span_bits > qf && span_bits <= 57u && span_bits - qf + PK_DISP_BITS + idx_bits <= 63u;
}

__device__ __forceinline__ unsigned long long dict_mode_of(long long min_id, long long max_id, uint64_t V,
                                                           uint32_t idx_bits, uint32_t pairs, uint32_t *span_bits_out) {
  const uint64_t span = (uint64_t)max_id - (uint64_t)min_id;
  const bool any = V > 0 && max_id >= min_id;
  const uint32_t span_bits = span ? 64u - (uint32_t)__clzll((long long)span) : 0u;
  *span_bits_out = span_bits;
  if (any && span < DIRECT_MAX_RANGE) return DICT_DIRECT;
  if (any && packed_fits(span_bits, pairs, idx_bits)) return DICT_PACKED8;
  return DICT_WIDE16;
}

static __global__ __launch_bounds__(256) void k_dict_init(const int64_t *__restrict__ vid, uint64_t V,
                                                          HtSlot *__restrict__ ht, uint64_t cap,
                                                          unsigned long long *__restrict__ tab, uint64_t nslots,
                                                          uint32_t *__restrict__ dir, DirectMap *__restrict__ dm,
                                                          BuildStatus *__restrict__ st, uint32_t part,
                                                          uint32_t n_parts, int64_t *__restrict__ vid_copy,
                                                          uint32_t *__restrict__ zero_u32, uint32_t n_zero) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (small jobs that would otherwise be commands of their own on the stream, ~5 us each: the CSR's copy of the
  // staged vertex ids, the column totals of the counter matrix)
  if (i < V) vid_copy[i] = vid[i];
  if (i < n_zero) zero_u32[i] = 0;
  if (i < cap) {
    uint4 e;
    e.x = 0u;
    e.y = 0x80000000u;  // key = INT64_MIN
    e.z = INVALID_U32;
    e.w = 0u;
    *reinterpret_cast<uint4 *>(&ht[i]) = e;
  }
  if (i < nslots) tab[i] = PK_EMPTY;
  if (i < DIRECT_MAX_RANGE) dir[i] = INVALID_U32;
  if (blockIdx.x < 64) {  // min / max (and a shard's owned-vertex count): 64 workgroups stride over the ids
    long long lo = INT64_MAX, hi = INT64_MIN;  // (two or three same-address atomics each, not per block of the grid)
    uint32_t owned = 0;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += 64ull * blockDim.x) {
      const long long x = vid[v];
      lo = x < lo ? x : lo;
      hi = x > hi ? x : hi;
      if (n_parts > 1 && owns(x, part, n_parts)) owned++;
    }
    if (n_parts > 1) {  // (whole builds own everything: set on the host)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) owned += __shfl_xor(owned, o, 64);
      if ((threadIdx.x & 63) == 0 && owned) atomicAdd(&st->owned, (unsigned long long)owned);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
    __shared__ long long s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) {
      s_lo[threadIdx.x >> 6] = lo;
      s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; w++) {
        lo = s_lo[w] < lo ? s_lo[w] : lo;
        hi = s_hi[w] > hi ? s_hi[w] : hi;
      }
      if (lo <= hi) {
        atomicMin(&dm->min_id, lo);
        atomicMax(&dm->max_id, hi);
      }
    }
  }
}

// 16-byte-table insert of vertex i (shared by k_dict_insert, k_dict_wide and the legacy k_ht_insert)
__device__ __forceinline__ void ht_insert_one(const int64_t *__restrict__ vid, uint64_t i, HtSlot *__restrict__ ht,
                                              uint64_t cap, BuildStatus *__restrict__ st) {
  const int64_t key = vid[i];
  if (key == HT_EMPTY) {  // the sentinel value itself is a legal id: keep it outside the table
    const long long prev = (long long)atomicCAS((unsigned long long *)&st->min_idx, (unsigned long long)-1LL,
                                                (unsigned long long)i);
    if (prev != -1LL) atomicOr(&st->dup_vertex, 1ULL);
    return;
  }
  uint64_t slot = ht_slot(key, cap);
  while (true) {
    const unsigned long long prev =
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

static __global__ __launch_bounds__(256) void k_dict_insert(const int64_t *__restrict__ vid, uint64_t V,
                                                            HtSlot *__restrict__ ht, uint64_t cap,
                                                            unsigned long long *__restrict__ tab,
                                                            uint32_t *__restrict__ dir, DirectMap *__restrict__ dm,
                                                            uint32_t idx_bits, uint32_t pairs,
                                                            BuildStatus *__restrict__ st) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const long long min_id = dm->min_id, max_id = dm->max_id;  // complete: k_dict_init has finished
  uint32_t span_bits;
  const unsigned long long mode = dict_mode_of(min_id, max_id, V, idx_bits, pairs, &span_bits);
  if (i == 0) {  // for the kernels after this one (nobody in this launch reads these fields)
    dm->decided = mode;
    dm->enabled = mode == DICT_DIRECT ? 1ULL : 0ULL;
    dm->idx_bits = idx_bits;
    dm->span_bits = span_bits;
    dm->pairs = pairs;
    if (mode != DICT_PACKED8) dm->mode = mode;  // packed: starts as DICT_PACKED8 (host), a failed insert flips it
  }
  if (i >= V) return;
  if (mode == DICT_DIRECT) {
    const uint32_t prev = atomicCAS(&dir[(uint64_t)vid[i] - (uint64_t)min_id], INVALID_U32, (uint32_t)i);
    if (prev != INVALID_U32) atomicOr(&st->dup_vertex, 1ULL);
  } else if (mode == DICT_PACKED8) {
    PkGeom pk;
    pk.set((uint64_t)min_id, (uint64_t)max_id, span_bits, pairs, idx_bits);
    uint64_t home, tag0;
    pk.locate(vid[i], &home, &tag0);
    for (uint64_t disp = 0; disp <= PK_MAX_DISP; disp++) {
      const uint64_t g = pk.pair_at(home, disp);
      const unsigned long long entry = ((tag0 | disp) << idx_bits) | i;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const unsigned long long prev = atomicCAS(&tab[2 * g + s], PK_EMPTY, entry);
        if (prev == PK_EMPTY) return;
        if ((prev >> idx_bits) == (tag0 | disp)) {
          atomicOr(&st->dup_vertex, 1ULL);
          return;
        }
      }
    }
    dm->mode = DICT_WIDE16;  // displaced too far: k_dict_wide fills the 16-byte table, the densification uses it
  } else {
    ht_insert_one(vid, i, ht, cap, st);
  }
}

static __global__ __launch_bounds__(256) void k_dict_wide(const int64_t *__restrict__ vid, uint64_t V,
                                                          HtSlot *__restrict__ ht, uint64_t cap,
                                                          const DirectMap *__restrict__ dm,
                                                          BuildStatus *__restrict__ st) {
  const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 == 0) st->dict_mode = dm->mode;  // final: k_dict_insert has finished
  if (dm->mode != DICT_WIDE16 || dm->decided == DICT_WIDE16) return;  // normally: nothing to do
  for (uint64_t i = i0; i < V; i += (uint64_t)gridDim.x * blockDim.x) ht_insert_one(vid, i, ht, cap, st);
}

}  // namespace gg
