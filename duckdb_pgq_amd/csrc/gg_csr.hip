// gg_csr.hip — CSR construction from the staged vertex/edge base-table columns (Finalize side).
//
// Replaces the reference's adjacency index = hash-join build side:
//   PhysicalHashJoin::Finalize -> JoinHashTable::Finalize/InsertHashes
//   (src/execution/operator/join/physical_hash_join.cpp:165-185, src/execution/join_hashtable.cpp:240-302),
// which inserts every build row single-threaded into a chained pointer table.  Here:
//   1. k_ht_insert      vertex ids -> open-addressing id hash table (dense index = table position)
//   2. k_edge_densify   (src,dst) ids -> dense (u,v); drops edges with a non-vertex endpoint;
//                       degree histogram by atomics on the (L2-resident) degree array
//   3. scan             degree -> row offsets
//   4. LSD radix passes stable radix-bucket scatter of (u, v, rowid) by u: per-wave digit
//                       histograms in LDS, one global prefix scan over (digit, wave) counters, then
//                       a wave-ordered scatter whose in-wave ranks come from __ballot match masks.
//      Stable => inside a CSR row neighbours keep ascending edge-rowid order; the build is
//      bit-reproducible run to run (no atomics decide a position).
// All integer work, HBM-bound: algorithmic bytes 40E + 16V (with rowid).
#include "gg_internal.h"

using namespace gg;

namespace gg {

struct BuildStatus {  // device-side status word block, copied back once per build
  unsigned long long dup_vertex;   // !=0: duplicate vertex id seen
  long long min_idx;               // dense index of the vertex with id == HT_EMPTY, or -1
  unsigned long long kept;         // edges kept (both endpoints are vertices)
};

__global__ __launch_bounds__(256) void k_ht_init(int64_t *__restrict__ keys, uint64_t cap) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) keys[i] = HT_EMPTY;
}

__global__ __launch_bounds__(256) void k_ht_insert(const int64_t *__restrict__ vid, uint64_t V,
                                                   int64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                   uint32_t shift, uint64_t mask, BuildStatus *__restrict__ st) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V) return;
  int64_t key = vid[i];
  if (key == HT_EMPTY) {  // the sentinel value itself is a legal id: keep it outside the table
    long long prev = (long long)atomicCAS((unsigned long long *)&st->min_idx, (unsigned long long)-1LL,
                                          (unsigned long long)i);
    if (prev != -1LL) atomicOr(&st->dup_vertex, 1ULL);
    return;
  }
  uint64_t slot = ((uint64_t)key * DIG_GOLD) >> shift;
  while (true) {
    unsigned long long prev =
        atomicCAS((unsigned long long *)&keys[slot], (unsigned long long)HT_EMPTY, (unsigned long long)key);
    if (prev == (unsigned long long)HT_EMPTY) {
      vals[slot] = (uint32_t)i;
      return;
    }
    if (prev == (unsigned long long)key) {
      atomicOr(&st->dup_vertex, 1ULL);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

// One thread per edge row: two id lookups (hash table is V-sized, L2/MALL resident), degree count.
__global__ __launch_bounds__(256) void k_edge_densify(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                      uint64_t E, const int64_t *__restrict__ keys,
                                                      const uint32_t *__restrict__ vals, uint32_t shift, uint64_t mask,
                                                      const BuildStatus *__restrict__ st, uint32_t *__restrict__ su,
                                                      uint32_t *__restrict__ dv, uint32_t *__restrict__ deg) {
  uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t min_idx = st->min_idx;
  uint32_t u = ht_lookup(keys, vals, shift, mask, min_idx, src[e]);
  uint32_t v = ht_lookup(keys, vals, shift, mask, min_idx, dst[e]);
  if (u == INVALID_U32 || v == INVALID_U32) {
    su[e] = INVALID_U32;
    dv[e] = INVALID_U32;
    return;
  }
  su[e] = u;
  dv[e] = v;
  atomicAdd(&deg[u], 1u);
}

// ---- stable LSD radix pass -------------------------------------------------------------------
// A wave owns a contiguous sub-tile of RS_WTILE elements and walks it in order, 64 at a time.
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_WTILE = 4096;          // elements per wave
constexpr int RS_MAX_BITS = 11;         // digits <= 2048 -> 4 x 2048 x 4 B = 32 KiB LDS per block

// counts[digit * nwaves + wave_global] = number of valid elements of that wave with that digit
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint32_t *__restrict__ key, uint64_t n,
                                                           uint32_t lo_bit, uint32_t bits, uint64_t nwaves,
                                                           uint32_t *__restrict__ counts) {
  extern __shared__ uint32_t lds_hist[];  // RS_WAVES << bits
  const uint32_t ndig = 1u << bits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < RS_WAVES * ndig; i += RS_THREADS) lds_hist[i] = 0;
  __syncthreads();
  const uint64_t wg = (uint64_t)blockIdx.x * RS_WAVES + wave;
  uint32_t *h = lds_hist + wave * ndig;
  if (wg < nwaves) {
    const uint64_t base = wg * RS_WTILE;
#pragma unroll 4
    for (int it = 0; it < RS_WTILE / 64; it++) {
      uint64_t idx = base + (uint64_t)it * 64 + lane;
      if (idx < n) {
        uint32_t k = key[idx];
        if (k != INVALID_U32) atomicAdd(&h[(k >> lo_bit) & (ndig - 1)], 1u);
      }
    }
  }
  __syncthreads();
  if (wg < nwaves)
    for (uint32_t d = lane; d < ndig; d += 64) counts[(uint64_t)d * nwaves + wg] = h[d];
}

// Scatter.  `bases` is the exclusive scan of `counts` (same layout).  LAST pass writes only the
// payload (neighbour + rowid) to the CSR arrays; earlier passes also carry the key.
template <bool LAST>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(
    const uint32_t *__restrict__ key_in, const uint32_t *__restrict__ val_in, const int64_t *__restrict__ eid_in,
    uint64_t n, uint32_t lo_bit, uint32_t bits, uint64_t nwaves, const uint32_t *__restrict__ bases,
    uint32_t *__restrict__ key_out, uint32_t *__restrict__ val_out, int64_t *__restrict__ eid_out) {
  extern __shared__ uint32_t lds_cur[];  // RS_WAVES << bits : per-wave running cursor per digit
  const uint32_t ndig = 1u << bits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t wg = (uint64_t)blockIdx.x * RS_WAVES + wave;
  if (wg >= nwaves) return;  // no block-level barrier below: waves are independent
  volatile uint32_t *cur = lds_cur + wave * ndig;
  for (uint32_t d = lane; d < ndig; d += 64) cur[d] = bases[(uint64_t)d * nwaves + wg];
  __builtin_amdgcn_wave_barrier();
  const uint64_t base = wg * RS_WTILE;
  const uint64_t lane_lt = (1ULL << lane) - 1ULL;
  for (int it = 0; it < RS_WTILE / 64; it++) {
    const uint64_t idx = base + (uint64_t)it * 64 + lane;
    if (base + (uint64_t)it * 64 >= n) break;  // wave-uniform
    uint32_t k = INVALID_U32, v = 0;
    int64_t r = 0;
    if (idx < n) {
      k = key_in[idx];
      v = val_in[idx];
      r = eid_in[idx];
    }
    const bool valid = (k != INVALID_U32);
    const uint32_t d = (k >> lo_bit) & (ndig - 1);
    // match mask: lanes of this wave holding the same digit (ballot per digit bit)
    uint64_t m = __ballot(valid);
    for (uint32_t b = 0; b < bits; b++) {
      uint64_t bb = __ballot((d >> b) & 1u);
      m &= ((d >> b) & 1u) ? bb : ~bb;
    }
    if (valid) {
      const uint32_t rank = __popcll(m & lane_lt);
      const uint32_t pos = cur[d] + rank;  // all peers read the cursor before the leader bumps it
      if (!LAST) key_out[pos] = k;
      val_out[pos] = v;
      eid_out[pos] = r;
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && (m & lane_lt) == 0) cur[d] += __popcll(m);  // lowest lane of each digit group
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void k_set_last_offset(uint32_t *__restrict__ off, uint64_t V,
                                                         const uint64_t *__restrict__ total,
                                                         BuildStatus *__restrict__ st) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    off[V] = (uint32_t)*total;
    st->kept = *total;
  }
}

}  // namespace gg

static int ceil_log2_u64(uint64_t v) {
  int b = 0;
  while ((1ULL << b) < v) b++;
  return b;
}

extern "C" void gg_csr_destroy(gg_csr *csr) {
  if (!csr) return;
  gg_ctx *ctx = csr->ctx;
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->dev_free(csr->off);
    ctx->dev_free(csr->nbr);
    ctx->dev_free(csr->eid);
    ctx->dev_free(csr->vid);
    ctx->dev_free(csr->ht_keys);
    ctx->dev_free(csr->ht_vals);
  }
  delete csr;
}

extern "C" int gg_csr_build(gg_ctx *ctx, gg_csr **out) {
  if (!ctx || !out) return GG_ERR_INVALID_ARG;
  *out = nullptr;
  GG_TRY(gg_staging_sync(ctx));
  GG_HIP(hipSetDevice(ctx->device));
  const uint64_t V = ctx->n_vertices, E = ctx->n_edges;
  hipStream_t s = ctx->stream;

  gg_csr *csr = new gg_csr();
  csr->ctx = ctx;
  csr->V = V;
  struct Guard {  // frees the half-built CSR on any early return
    gg_csr *c;
    bool armed = true;
    ~Guard() {
      if (armed) gg_csr_destroy(c);
    }
  } guard{csr};

  // ---- id hash table --------------------------------------------------------------------
  int lg = ceil_log2_u64(V * 2 < 1024 ? 1024 : V * 2);
  csr->ht_cap = 1ULL << lg;
  csr->ht_shift = 64 - lg;
  GG_TRY(ctx->dev_alloc((void **)&csr->ht_keys, csr->ht_cap * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->ht_vals, csr->ht_cap * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->vid, (V ? V : 1) * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->off, (V + 1) * sizeof(uint32_t)));
  BuildStatus *st = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&st, sizeof(BuildStatus)));
  BuildStatus init{0ULL, -1LL, 0ULL};
  memcpy(ctx->pin_scratch, &init, sizeof(init));
  GG_HIP(hipMemcpyAsync(st, ctx->pin_scratch, sizeof(init), hipMemcpyHostToDevice, s));
  if (V) GG_HIP(hipMemcpyAsync(csr->vid, ctx->c_vid.dev, V * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
  GG_LAUNCH(ctx, "ht_init", k_ht_init, dim3((unsigned)((csr->ht_cap + 255) / 256)), dim3(256), 0, csr->ht_keys,
            csr->ht_cap);
  if (V)
    GG_LAUNCH(ctx, "ht_insert", k_ht_insert, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, csr->vid, V,
              csr->ht_keys, csr->ht_vals, csr->ht_shift, csr->ht_cap - 1, st);

  // ---- densify + degree histogram ----------------------------------------------------------
  uint32_t *su = nullptr, *dv = nullptr, *deg = nullptr;
  uint64_t *total = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&su, (E ? E : 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&dv, (E ? E : 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&deg, (V + 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&total, sizeof(uint64_t)));
  GG_HIP(hipMemsetAsync(deg, 0, (V + 1) * sizeof(uint32_t), s));
  if (E)
    GG_LAUNCH(ctx, "edge_densify", k_edge_densify, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, ctx->c_src.dev,
              ctx->c_dst.dev, E, csr->ht_keys, csr->ht_vals, csr->ht_shift, csr->ht_cap - 1, st, su, dv, deg);

  // ---- offsets ------------------------------------------------------------------------------
  GG_TRY(scan_exclusive_u32(ctx, deg, csr->off, V, total));
  GG_LAUNCH(ctx, "set_last_offset", k_set_last_offset, dim3(1), dim3(64), 0, csr->off, V, total, st);

  // status back to the host: duplicate check, sentinel vertex, kept-edge count
  GG_HIP(hipMemcpyAsync(ctx->pin_scratch, st, sizeof(BuildStatus), hipMemcpyDeviceToHost, s));
  GG_HIP(hipStreamSynchronize(s));
  BuildStatus hs;
  memcpy(&hs, ctx->pin_scratch, sizeof(hs));
  ctx->dev_free(st);
  ctx->dev_free(deg);
  ctx->dev_free(total);
  if (hs.dup_vertex) {
    ctx->dev_free(su);
    ctx->dev_free(dv);
    set_error("vertex key column is not unique (duplicate vertex id)");
    return GG_ERR_DUPLICATE_VERTEX;
  }
  csr->ht_min_idx = hs.min_idx;
  csr->E = hs.kept;
  csr->dropped = E - hs.kept;

  GG_TRY(ctx->dev_alloc((void **)&csr->nbr, (csr->E ? csr->E : 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&csr->eid, (csr->E ? csr->E : 1) * sizeof(int64_t)));

  // ---- stable LSD radix scatter by source ------------------------------------------------------
  if (E && csr->E) {
    int key_bits = ceil_log2_u64(V < 2 ? 2 : V);
    int passes = (key_bits + RS_MAX_BITS - 1) / RS_MAX_BITS;
    int bits_per = (key_bits + passes - 1) / passes;
    uint32_t *kbuf[2] = {nullptr, nullptr}, *vbuf[2] = {nullptr, nullptr};
    int64_t *ebuf[2] = {nullptr, nullptr};
    if (passes > 1) {
      for (int i = 0; i < (passes > 2 ? 2 : 1); i++) {
        GG_TRY(ctx->dev_alloc((void **)&kbuf[i], csr->E * sizeof(uint32_t)));
        GG_TRY(ctx->dev_alloc((void **)&vbuf[i], csr->E * sizeof(uint32_t)));
        GG_TRY(ctx->dev_alloc((void **)&ebuf[i], csr->E * sizeof(int64_t)));
      }
    }
    const uint32_t *kin = su, *vin = dv;
    const int64_t *ein = ctx->c_rowid.dev;
    uint64_t n_in = E;  // pass 0 reads all staged rows (and drops invalid ones); later passes read E_kept
    for (int p = 0; p < passes; p++) {
      const bool last = (p == passes - 1);
      const uint32_t lo_bit = (uint32_t)(p * bits_per);
      const uint32_t bits = (uint32_t)((key_bits - (int)lo_bit) < bits_per ? (key_bits - (int)lo_bit) : bits_per);
      const uint64_t nwaves = (n_in + RS_WTILE - 1) / RS_WTILE;
      const unsigned nblocks = (unsigned)((nwaves + RS_WAVES - 1) / RS_WAVES);
      const size_t lds = (size_t)RS_WAVES * (1u << bits) * sizeof(uint32_t);
      const uint64_t ncount = (uint64_t)(1u << bits) * nwaves;
      uint32_t *counts = nullptr;
      GG_TRY(ctx->dev_alloc((void **)&counts, ncount * sizeof(uint32_t)));
      GG_LAUNCH(ctx, "radix_hist", k_radix_hist, dim3(nblocks), dim3(RS_THREADS), lds, kin, n_in, lo_bit, bits,
                nwaves, counts);
      GG_TRY(scan_exclusive_u32(ctx, counts, counts, ncount, nullptr));
      uint32_t *kout = last ? nullptr : kbuf[p & 1];
      uint32_t *vout = last ? csr->nbr : vbuf[p & 1];
      int64_t *eout = last ? csr->eid : ebuf[p & 1];
      if (last)
        GG_LAUNCH(ctx, "radix_scatter", (k_radix_scatter<true>), dim3(nblocks), dim3(RS_THREADS), lds, kin, vin, ein,
                  n_in, lo_bit, bits, nwaves, counts, kout, vout, eout);
      else
        GG_LAUNCH(ctx, "radix_scatter", (k_radix_scatter<false>), dim3(nblocks), dim3(RS_THREADS), lds, kin, vin, ein,
                  n_in, lo_bit, bits, nwaves, counts, kout, vout, eout);
      ctx->dev_free(counts);
      kin = kout;
      vin = vout;
      ein = eout;
      n_in = csr->E;
    }
    for (int i = 0; i < 2; i++) {
      ctx->dev_free(kbuf[i]);
      ctx->dev_free(vbuf[i]);
      ctx->dev_free(ebuf[i]);
    }
  }
  ctx->dev_free(su);
  ctx->dev_free(dv);
  GG_HIP(hipStreamSynchronize(s));
  guard.armed = false;
  *out = csr;
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
  if (eid && csr->E) GG_HIP(hipMemcpy(eid, csr->eid, csr->E * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (vid && csr->V) GG_HIP(hipMemcpy(vid, csr->vid, csr->V * sizeof(int64_t), hipMemcpyDeviceToHost));
  return GG_OK;
}
