// gg_khop.hip — fixed-length path expansion (k-hop MATCH) over the device CSR.
//
// Replaces the probe side of the reference's hash-join chain:
//   PhysicalHashJoin::Execute -> JoinHashTable::Probe + ScanStructure::NextInnerJoin/AdvancePointers
//   (src/execution/operator/join/physical_hash_join.cpp:217-254, src/execution/join_hashtable.cpp:304-476)
// whose cost is a dependent pointer chase per matching build row.  Here a path prefix ending in
// vertex v expands into CSR row v, read as a contiguous, coalesced run.
//
// Work decomposition ("flattened frontier"): a frontier is a list of path prefixes; entry i ends
// in vertex fv[i] and owns deg(fv[i]) children; foff[] is the exclusive prefix sum of those
// degrees, so child positions 0..M-1 enumerate every (prefix, neighbour) pair.  A workgroup takes
// a tile of 256 consecutive child positions (k_tile_partition finds the entry each tile starts
// in), so work per workgroup is balanced on EDGES, not vertices.
//
//   k_expand_fused2  last two hops fused: phase 1 resolves the tile's 256 children x (one
//                    coalesced nbr read + two offset reads each) and stages {row start, row
//                    length, hash state} in LDS; phase 2: each wavefront streams whole CSR rows
//                    (64 lanes x 4 B coalesced) and folds every entry into the row digest.
//   k_expand_step    one hop, materialising the next frontier (only needed for k_max >= 3)
//   k_expand_last1   single last hop (k_max == 1)
// Count-only mode never writes rows: algorithmic bytes 8*TE + 16*frontier entries.
#include "gg_internal.h"

using namespace gg;

namespace gg {

constexpr int XT = 256;  // child positions per tile == threads per workgroup
// The runtime takes a grid as gridDim.x * blockDim.x threads in 32 bits: at most this many XT-thread workgroups per
// launch (about 1.3e10 3-hop tile rows; SF100's 12.8 G 2-hop rows are 16.6 M tiles of 768, just below it).
constexpr uint64_t MAX_GRID_TILES = 0xFFFFFFFFull / XT;
// second-to-last frontiers of at least this many entries take the product form of the last hop (pairs + sort by last
// vertex + fold); smaller ones are over before three sort passes have been launched
constexpr uint64_t FRONT_PRODUCT_MIN = 1u << 16;
// last levels of at least this many rows are materialised in the product form (k_mat_front)
constexpr uint64_t MAT_FRONT_MIN = 1u << 20;

// tile t starts in entry  upper_bound(foff, foff[0] + t*XT) - 1
template <typename OffT>
__global__ __launch_bounds__(256) void k_tile_partition(const OffT *__restrict__ foff, uint64_t n_entries,
                                                        uint64_t n_tiles, uint32_t *__restrict__ tile_entry) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  const uint64_t target = (uint64_t)foff[0] + t * XT;
  uint64_t lo = 0, hi = n_entries;  // first idx in [0,n_entries] with foff[idx] > target
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if ((uint64_t)foff[mid] <= target)
      lo = mid + 1;
    else
      hi = mid;
  }
  tile_entry[t] = (uint32_t)(lo - 1);
}

// Locate the frontier entry that owns flattened child position p.  s_foff holds foff[i0..i0+XT]
// (UINT64_MAX past the end).  Returns entry index and writes the position inside the entry.
template <typename OffT>
__device__ __forceinline__ uint64_t locate_entry(const uint64_t *s_foff, const OffT *__restrict__ foff,
                                                 uint64_t n_entries, uint64_t i0, uint64_t p, uint64_t *k) {
  uint32_t lo = 0, hi = XT + 1;  // first idx in [0, XT+1) with s_foff[idx] > p
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (s_foff[mid] <= p)
      lo = mid + 1;
    else
      hi = mid;
  }
  uint64_t idx = i0 + lo - 1;
  uint64_t start = s_foff[lo - 1];
  if (lo == XT + 1) {  // window exhausted by zero-degree entries: finish the search in global memory
    uint64_t glo = i0 + XT, ghi = n_entries;
    while (glo < ghi) {
      uint64_t mid = (glo + ghi) >> 1;
      if ((uint64_t)foff[mid] <= p)
        glo = mid + 1;
      else
        ghi = mid;
    }
    idx = glo - 1;
    start = (uint64_t)foff[idx];
  }
  *k = p - start;
  return idx;
}

template <typename OffT>
__device__ __forceinline__ void load_window(uint64_t *s_foff, const OffT *__restrict__ foff, uint64_t n_entries,
                                            uint64_t i0) {
  for (uint32_t t = threadIdx.x; t <= XT; t += XT) {
    uint64_t gi = i0 + t;
    s_foff[t] = gi <= n_entries ? (uint64_t)foff[gi] : UINT64_MAX;
  }
}

struct Frontier {
  const uint32_t *fv;   // last vertex per entry (nullptr: identity, vertex = ident_base + i)
  const uint64_t *fq;   // hash state q_j per entry (nullptr: q_0 computed from the vertex)
  uint64_t n_entries;
  uint32_t ident_base;
  int j;                // hops already in the prefix
};

__device__ __forceinline__ void entry_vertex_q(const Frontier &f, uint64_t i, uint32_t *v, uint64_t *q) {
  uint32_t vv = f.fv ? f.fv[i] : (uint32_t)(f.ident_base + i);
  *v = vv;
  *q = f.fq ? f.fq[i] : dig_q((uint64_t)vv, 0);
}

// block reduce -> partial[blockIdx.x*4 + 0..2]; a, b are digest sums (lane-wise 2 x u32), c a plain count
__device__ __forceinline__ void block_store_partials(uint64_t a, uint64_t b, uint64_t c, uint64_t *s_red /*12*/,
                                                     unsigned long long *__restrict__ partial) {
  a = wave_reduce_dsum(a);
  b = wave_reduce_dsum(b);
  c = wave_total_u50(c);  // (a lane's rows: far below 2^50)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_red[wave * 3 + 0] = a;
    s_red[wave * 3 + 1] = b;
    s_red[wave * 3 + 2] = c;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const uint64_t v0 = s_red[threadIdx.x], v1 = s_red[3 + threadIdx.x], v2 = s_red[6 + threadIdx.x],
                   v3 = s_red[9 + threadIdx.x];
    const uint64_t s = threadIdx.x < 2 ? dsum_add(dsum_add(v0, v1), dsum_add(v2, v3)) : v0 + v1 + v2 + v3;
    partial[(uint64_t)blockIdx.x * 4 + threadIdx.x] = s;
  }
}

// ---- fused last two hops -----------------------------------------------------------------------
template <typename OffT>
__global__ __launch_bounds__(XT) void k_expand_fused2(const uint32_t *__restrict__ off,
                                                      const uint32_t *__restrict__ nbr, const OffT *__restrict__ foff,
                                                      Frontier f, uint64_t M, const uint32_t *__restrict__ tile_entry,
                                                      int emit_mid, unsigned long long *__restrict__ partial) {
  __shared__ uint64_t s_foff[XT + 1];
  __shared__ uint64_t s_q[XT];
  __shared__ uint32_t s_start[XT];
  __shared__ uint32_t s_len[XT];
  __shared__ uint64_t s_red[12];

  const uint64_t fbase = (uint64_t)foff[0];
  const uint64_t p = fbase + (uint64_t)blockIdx.x * XT + threadIdx.x;
  const uint64_t i0 = tile_entry[blockIdx.x];
  load_window(s_foff, foff, f.n_entries, i0);
  __syncthreads();

  uint64_t mid_sum = 0, rows_last = 0;
  uint32_t my_len = 0;
  if (p < fbase + M) {
    uint64_t k;
    uint64_t i = locate_entry(s_foff, foff, f.n_entries, i0, p, &k);
    uint32_t v;
    uint64_t q;
    entry_vertex_q(f, i, &v, &q);
    const uint32_t x = nbr[(uint64_t)off[v] + k];
    const uint64_t P = dig_leaf(q, x);
    if (emit_mid) mid_sum = P;
    const uint32_t st = off[x];
    my_len = off[x + 1] - st;
    s_start[threadIdx.x] = st;
    s_q[threadIdx.x] = dig_q(P, f.j + 1);
    rows_last = my_len;
  }
  s_len[threadIdx.x] = my_len;
  __syncthreads();

  // phase 2: a wavefront per staged row; lanes stride the row, 4 independent loads in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t acc = 0;
  for (int r = wave; r < XT; r += XT / 64) {
    const uint32_t len = s_len[r];
    if (len == 0) continue;
    const uint32_t st = s_start[r];
    const uint64_t q2 = s_q[r];
    const uint32_t *__restrict__ row = nbr + st;
    uint32_t jj = lane;
    for (; jj + 192 < len; jj += 256) {
      uint32_t w0 = row[jj], w1 = row[jj + 64], w2 = row[jj + 128], w3 = row[jj + 192];
      acc = dsum_add(acc, dig_leaf(q2, w0));
      acc = dsum_add(acc, dig_leaf(q2, w1));
      acc = dsum_add(acc, dig_leaf(q2, w2));
      acc = dsum_add(acc, dig_leaf(q2, w3));
    }
    for (; jj < len; jj += 64) acc = dsum_add(acc, dig_leaf(q2, row[jj]));
  }
  block_store_partials(mid_sum, acc, rows_last, s_red, partial);
}

// ---- one hop, materialising the next frontier (fv', fq', deg') -----------------------------------
template <typename OffT>
__global__ __launch_bounds__(XT) void k_expand_step(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                    const OffT *__restrict__ foff, Frontier f, uint64_t M,
                                                    const uint32_t *__restrict__ tile_entry, int emit,
                                                    uint32_t *__restrict__ nv, uint64_t *__restrict__ nq,
                                                    uint64_t *__restrict__ ndeg,
                                                    unsigned long long *__restrict__ partial) {
  __shared__ uint64_t s_foff[XT + 1];
  __shared__ uint64_t s_red[12];
  const uint64_t fbase = (uint64_t)foff[0];
  const uint64_t p = fbase + (uint64_t)blockIdx.x * XT + threadIdx.x;
  const uint64_t i0 = tile_entry[blockIdx.x];
  load_window(s_foff, foff, f.n_entries, i0);
  __syncthreads();
  uint64_t sum = 0;
  if (p < fbase + M) {
    uint64_t k;
    uint64_t i = locate_entry(s_foff, foff, f.n_entries, i0, p, &k);
    uint32_t v;
    uint64_t q;
    entry_vertex_q(f, i, &v, &q);
    const uint32_t x = nbr[(uint64_t)off[v] + k];
    const uint64_t P = dig_leaf(q, x);
    if (emit) sum = P;
    const uint64_t o = p - fbase;
    nv[o] = x;
    nq[o] = dig_q(P, f.j + 1);
    ndeg[o] = (uint64_t)(off[x + 1] - off[x]);
  }
  block_store_partials(sum, 0, 0, s_red, partial);
}

// ---- single last hop -----------------------------------------------------------------------------
template <typename OffT>
__global__ __launch_bounds__(XT) void k_expand_last1(const uint32_t *__restrict__ off,
                                                     const uint32_t *__restrict__ nbr, const OffT *__restrict__ foff,
                                                     Frontier f, uint64_t M, const uint32_t *__restrict__ tile_entry,
                                                     unsigned long long *__restrict__ partial) {
  __shared__ uint64_t s_foff[XT + 1];
  __shared__ uint64_t s_red[12];
  const uint64_t fbase = (uint64_t)foff[0];
  const uint64_t p = fbase + (uint64_t)blockIdx.x * XT + threadIdx.x;
  const uint64_t i0 = tile_entry[blockIdx.x];
  load_window(s_foff, foff, f.n_entries, i0);
  __syncthreads();
  uint64_t sum = 0;
  if (p < fbase + M) {
    uint64_t k;
    uint64_t i = locate_entry(s_foff, foff, f.n_entries, i0, p, &k);
    uint32_t v;
    uint64_t q;
    entry_vertex_q(f, i, &v, &q);
    sum = dig_leaf(q, nbr[(uint64_t)off[v] + k]);
  }
  block_store_partials(0, sum, 0, s_red, partial);
}


// ---- 2-hop as a join product around the middle vertex ---------------------------------------------
// Every 2-hop walk u -> x -> w is one pair (in-edge of x, out-edge of x): the walks through x are the
// cartesian product in(x) x out(x) — exactly what the reference's second hash join emits when a probe
// key matches a chain of build rows (join_hashtable.cpp:442-476).  Reading CSR row x once per walk (the
// frontier formulation above) therefore re-reads it |in(x)| times.  Here a tile of MT (768) consecutive
// REVERSE-CSR entries (= 1-hop rows u->x, grouped by x) is staged in LDS as hash states; for each run
// of equal x the out-row is loaded ONCE into registers (64 lanes x MID_R values, coalesced, through a buffer
// descriptor) and every staged state of the run is folded against it (one 16-byte LDS broadcast read per four
// states, ONE v_xad_u32 per walk).  The kernel is VALU-bound; HBM/L2 traffic drops to one pass over both CSRs.
#ifndef GG_MID_ISPLIT
#define GG_MID_ISPLIT 128  // runs of at least this many states are split over the workgroup's waves
#endif
constexpr int MID_R = 8;  // out-row values per lane per J-block: a J-block covers 64*MID_R = 512 leaves

#define GG_XAD(acc, q, t) asm("v_xad_u32 %0, %1, %2, %0" : "+v"(acc) : "v"(q), "v"(t))

// acc = (q ^ t) + acc in ONE instruction per walk; spelled out because the compiler otherwise splits the
// unrolled loop into v_xor_b32 + v_add3_u32.
template <int NREG>
__device__ __forceinline__ void mid_fold(uint32_t q, const uint32_t (&t)[MID_R], uint32_t (&acc)[MID_R]) {
#pragma unroll
  for (int r = 0; r < NREG; r++) GG_XAD(acc[r], q, t[r]);
}

template <int NREG>
__device__ __forceinline__ void mid_accumulate(const uint32_t *s_q, int ia, int ib, const uint32_t (&t)[MID_R],
                                               uint32_t (&acc)[MID_R]) {
  // the slice bounds are the same in every lane: scalar loop control; states are read four at a time with one
  // 16-byte LDS broadcast read, the next four already in flight while the current ones are folded
  ia = __builtin_amdgcn_readfirstlane(ia);
  ib = __builtin_amdgcn_readfirstlane(ib);
  int i = ia;
  for (; i < ib && (i & 3); i++) mid_fold<NREG>(s_q[i], t, acc);
  if (i + 4 <= ib) {
    // two groups of four per trip with fixed roles (no register copies between trips)
    uint4 c0 = *reinterpret_cast<const uint4 *>(s_q + i);
    for (; i + 12 <= ib; i += 8) {
      const uint4 c1 = *reinterpret_cast<const uint4 *>(s_q + i + 4);
      mid_fold<NREG>(c0.x, t, acc);
      mid_fold<NREG>(c0.y, t, acc);
      mid_fold<NREG>(c0.z, t, acc);
      mid_fold<NREG>(c0.w, t, acc);
      c0 = *reinterpret_cast<const uint4 *>(s_q + i + 8);
      mid_fold<NREG>(c1.x, t, acc);
      mid_fold<NREG>(c1.y, t, acc);
      mid_fold<NREG>(c1.z, t, acc);
      mid_fold<NREG>(c1.w, t, acc);
    }
    if (i + 8 <= ib) {
      const uint4 c1 = *reinterpret_cast<const uint4 *>(s_q + i + 4);
      mid_fold<NREG>(c0.x, t, acc);
      mid_fold<NREG>(c0.y, t, acc);
      mid_fold<NREG>(c0.z, t, acc);
      mid_fold<NREG>(c0.w, t, acc);
      c0 = c1;
      i += 4;
    }
    mid_fold<NREG>(c0.x, t, acc);
    mid_fold<NREG>(c0.y, t, acc);
    mid_fold<NREG>(c0.z, t, acc);
    mid_fold<NREG>(c0.w, t, acc);
    i += 4;
  }
  for (; i < ib; i++) mid_fold<NREG>(s_q[i], t, acc);
}

#ifndef GG_MID_EPT
#define GG_MID_EPT 3
#endif
constexpr int MID_EPT = GG_MID_EPT;    // reverse-CSR entries per thread
constexpr int MT = XT * MID_EPT;       // entries per tile (workgroup): 768 (512: 797 us, 1024: 813 us, 768: 757 us at SF100)
constexpr int MID_SEG = MID_EPT * (XT / 64);  // (entry slot, wave) segments of a tile, in position order

struct alignas(16) MidShared {  // LDS image of one tile
  uint32_t q[MT];          // low half of the hash state q_1 of each 1-hop row of the tile
  uint32_t pq[MT + 1];     // prefix sums (mod 2^32) of q: pq[i] = sum of q[0..i)
  uint32_t x[MT];          // middle vertex of each row
  uint32_t run[MT + 1];    // tile position where each run of equal x starts (+ end sentinel)
  uint32_t rst[MT];        // per run: start of out-row x in nbr
  uint32_t rdout[MT];      // per run: out-degree of x
  uint32_t wcnt[MID_SEG];
  uint32_t wsum[MID_SEG];
  uint64_t red[12];
};

struct MidRows {  // a thread's MID_EPT rows of one tile, as loaded from the reverse CSR
  uint32_t x[MID_EPT], u[MID_EPT];
  bool valid[MID_EPT];
};
struct MidPrep {  // the same rows after hashing and the out-row lookups
  uint64_t q[MID_EPT];
  uint32_t st[MID_EPT], dout[MID_EPT];
};

// stage 1: reverse-CSR entry p is the 1-hop row u -> x with x = rrow[p] (COO view), u = rnbr[p]:
// two coalesced loads, no search.  Tile position of thread t's e-th entry: e*XT + t.
__device__ __forceinline__ void mid_load_rows(const uint32_t *__restrict__ rrow, const uint32_t *__restrict__ rnbr,
                                              uint64_t fbase, uint64_t M, uint64_t tile, MidRows &r) {
#pragma unroll
  for (int e = 0; e < MID_EPT; e++) {
    const uint64_t i = tile * MT + (uint64_t)(e * XT) + threadIdx.x;
    r.valid[e] = i < M;
    r.x[e] = INVALID_U32;
    r.u[e] = 0;
    if (r.valid[e]) {
      r.x[e] = rrow[fbase + i];
      r.u[e] = rnbr[fbase + i];
    }
  }
}

// stage 2: row hashes (two fmix64 each) and the out-row extents of the middle vertices (gathers)
__device__ __forceinline__ void mid_prepare(const uint32_t *__restrict__ off, const MidRows &r, int emit_mid,
                                            MidPrep &p, uint64_t &mid_sum, uint64_t &rows_last) {
#pragma unroll
  for (int e = 0; e < MID_EPT; e++) {
    p.q[e] = 0;
    p.st[e] = p.dout[e] = 0;
    if (r.valid[e]) {
      p.st[e] = off[r.x[e]];
      p.dout[e] = off[r.x[e] + 1] - p.st[e];
      const uint64_t P = dig_leaf(dig_q((uint64_t)r.u[e], 0), r.x[e]);
      if (emit_mid) mid_sum = dsum_add(mid_sum, P);
      p.q[e] = dig_q(P, 1);
      rows_last += (uint64_t)p.dout[e];
    }
  }
}

// stage 3: LDS image of the tile (states, their prefix sums, runs of equal middle vertex).  Returns the
// number of runs.  Contains three workgroup barriers; the caller must have finished reading the previous image.
__device__ __forceinline__ uint32_t mid_stage(MidShared &sm, const MidRows &r, const MidPrep &p, uint64_t M,
                                              uint64_t tile) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t plo[MID_EPT];
#pragma unroll
  for (int e = 0; e < MID_EPT; e++) {
    const uint32_t idx = e * XT + threadIdx.x;
    sm.q[idx] = (uint32_t)p.q[e];
    sm.x[idx] = r.x[e];
    plo[e] = wave_scan_incl((uint32_t)p.q[e]);  // (mod 2^32)
    if (lane == 63) sm.wsum[e * (XT / 64) + wave] = plo[e];
  }
  __syncthreads();
  // sums of the segments before each segment: lane s holds segment s, one scan over the first row of 16 lanes
  auto seg_scan = [&](const uint32_t *per_seg, uint32_t &incl) {
    const uint32_t v = lane < MID_SEG ? per_seg[lane] : 0u;
    incl = row_scan_incl(v);
    return incl - v;
  };
  static_assert(MID_SEG <= 16, "segment scan covers 16 lanes");
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  uint32_t s_incl;
  const uint32_t s_excl = seg_scan(sm.wsum, s_incl);
  bool head[MID_EPT];
  uint64_t hm[MID_EPT];
#pragma unroll
  for (int e = 0; e < MID_EPT; e++) {
    const uint32_t idx = e * XT + threadIdx.x;
    head[e] = r.valid[e] && (idx == 0 || sm.x[idx - 1] != r.x[e]);
    hm[e] = __ballot(head[e]);
    const int seg = e * (XT / 64) + uwave;
    if (lane == 0) sm.wcnt[seg] = (uint32_t)__popcll(hm[e]);
    sm.pq[idx + 1] = (uint32_t)__builtin_amdgcn_readlane((int)s_excl, seg) + plo[e];
  }
  if (threadIdx.x == 0) sm.pq[0] = 0;
  __syncthreads();
  uint32_t c_incl;
  const uint32_t c_excl = seg_scan(sm.wcnt, c_incl);
  const uint32_t nruns = (uint32_t)__builtin_amdgcn_readlane((int)c_incl, MID_SEG - 1);
#pragma unroll
  for (int e = 0; e < MID_EPT; e++) {
    const uint32_t hbase = (uint32_t)__builtin_amdgcn_readlane((int)c_excl, e * (XT / 64) + uwave);
    if (head[e]) {
      const uint32_t rr = hbase + __popcll(hm[e] & ((1ULL << lane) - 1ULL));
      sm.run[rr] = e * XT + threadIdx.x;
      sm.rst[rr] = p.st[e];
      sm.rdout[rr] = p.dout[e];
    }
  }
  const uint64_t remaining = M - tile * MT;
  if (threadIdx.x == 0) sm.run[nruns] = remaining < MT ? (uint32_t)remaining : MT;
  __syncthreads();
  return nruns;
}

// One J-block: NREG out-row values per lane (loads issue back to back: a lane past the row's end reads the last
// entry and zeroes it), every state of the slice folded against them.  A lane without a leaf added q ^ 0 = q for
// every state of the slice: `corr` takes that back.
template <int NREG>
__device__ __forceinline__ void mid_block(const uint32_t *s_q, const uint32_t *__restrict__ row, uint32_t j0,
                                          uint32_t dout, int ia, int ib, uint32_t sq, uint32_t (&acc)[MID_R],
                                          uint32_t &corr) {
  uint32_t t[MID_R];
  uint32_t ninv = 0;
  // the out-row as a buffer of dout words (descriptor in scalar registers, the run is the same in every lane): a
  // lane past the row's end reads 0 without a compare or a clamped index, the NREG offsets are immediates
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(row), 0, (int)(dout * 4u), 0x00020000);
  const uint32_t voff = j0 * 4u;
#pragma unroll
  for (int r = 0; r < NREG; r++)
    t[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + r * 256u, 0, 0) * DIG_K32;
  const uint32_t left = dout > j0 ? (dout - j0 + 63u) >> 6 : 0u;  // registers of this lane that hold a leaf
  ninv = NREG - (left < (uint32_t)NREG ? left : (uint32_t)NREG);
  mid_accumulate<NREG>(s_q, ia, ib, t, acc);
  corr += ninv * sq;
}

// stage 4: fold every staged state against the out-row of its run's middle vertex
__device__ __forceinline__ void mid_fold_tile(const MidShared &sm, uint32_t nruns, const uint32_t *__restrict__ nbr,
                                              uint32_t (&acc)[MID_R], uint32_t &corr) {
  const int lane = threadIdx.x & 63;
  // Run descriptors 64 at a time, one per lane; the wave then walks only the runs it owns (ballot), with the
  // descriptor in scalar registers.  (Every wave stepping through every run, three quarters of them only to skip
  // them, was a seventh of the kernel's VALU instructions.)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (uint32_t r0 = 0; r0 < nruns; r0 += 64) {
    const uint32_t rl = r0 + (uint32_t)lane;
    int la = 0, llen = 0;
    uint32_t ldout = 0, lst = 0;
    if (rl < nruns) {
      la = (int)sm.run[rl];
      llen = (int)sm.run[rl + 1] - la;
      ldout = sm.rdout[rl];
      lst = sm.rst[rl];
    }
    // long runs: every wave takes a slice of the run (i-split) and walks all J-blocks;
    // short runs: the whole run belongs to ONE wave (dealt round-robin)
    uint64_t todo = __ballot(rl < nruns && ldout != 0 &&
                             (llen >= GG_MID_ISPLIT || (rl & (XT / 64 - 1)) == (uint32_t)wave));
    while (todo) {
      const int l = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const int a = __builtin_amdgcn_readlane(la, l), len = __builtin_amdgcn_readlane(llen, l);
      const uint32_t dout = (uint32_t)__builtin_amdgcn_readlane((int)ldout, l);
      const uint32_t *__restrict__ row = nbr + (uint32_t)__builtin_amdgcn_readlane((int)lst, l);
      const bool isplit = len >= GG_MID_ISPLIT;
      const int ia = isplit ? a + (len * wave) / (XT / 64) : a;
      const int ib = isplit ? a + (len * (wave + 1)) / (XT / 64) : a + len;
      if (ia >= ib) continue;
      const uint32_t sq = sm.pq[ib] - sm.pq[ia];  // sum of the slice's hash states
      // J-blocks of equal size (multiple of 64 leaves, at most 64*MID_R): avoids a nearly empty tail block
      const uint32_t nJ = (dout + 64 * MID_R - 1) / (64 * MID_R);
      const uint32_t jsz = nJ == 1 ? (dout + 63) & ~63u : (((dout + nJ - 1) / nJ) + 63) & ~63u;  // (no division for <= 512)
      const int nreg = (int)(jsz >> 6);
      for (uint32_t jb = 0; jb < nJ; jb++) {
        const uint32_t base = jb * jsz + (uint32_t)lane;
        switch (nreg) {
        case 1: mid_block<1>(sm.q, row, base, dout, ia, ib, sq, acc, corr); break;
        case 2: mid_block<2>(sm.q, row, base, dout, ia, ib, sq, acc, corr); break;
        case 3: mid_block<3>(sm.q, row, base, dout, ia, ib, sq, acc, corr); break;
        case 4: mid_block<4>(sm.q, row, base, dout, ia, ib, sq, acc, corr); break;
        case 5: mid_block<5>(sm.q, row, base, dout, ia, ib, sq, acc, corr); break;
        case 6: mid_block<6>(sm.q, row, base, dout, ia, ib, sq, acc, corr); break;
        case 7: mid_block<7>(sm.q, row, base, dout, ia, ib, sq, acc, corr); break;
        default: mid_block<8>(sm.q, row, base, dout, ia, ib, sq, acc, corr); break;
        }
      }
    }
  }
}

// One tile per workgroup.
__global__ __launch_bounds__(XT) void k_expand_mid2(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                    const uint32_t *__restrict__ rrow, const uint32_t *__restrict__ rnbr,
                                                    uint64_t fbase, uint64_t M, int emit_mid,
                                                    unsigned long long *__restrict__ partial,
                                                    unsigned long long *__restrict__ zero3) {
  __shared__ MidShared sm;
  if (blockIdx.x == 0 && threadIdx.x < 3) zero3[threadIdx.x] = 0;  // k_reduce_partials' accumulators (next kernel)
  uint64_t mid_sum = 0, rows_last = 0;
  uint32_t corr = 0;
  uint32_t acc[MID_R];
#pragma unroll
  for (int r = 0; r < MID_R; r++) acc[r] = 0;
  MidRows rows;
  MidPrep prep;
  mid_load_rows(rrow, rnbr, fbase, M, blockIdx.x, rows);
  mid_prepare(off, rows, emit_mid, prep, mid_sum, rows_last);
  const uint32_t nruns = mid_stage(sm, rows, prep, M, blockIdx.x);
  mid_fold_tile(sm, nruns, nbr, acc, corr);
  uint32_t tsum = 0;
#pragma unroll
  for (int r = 0; r < MID_R; r++) tsum += acc[r];
  const uint64_t total = (uint64_t)(uint32_t)(tsum - corr);
  block_store_partials(mid_sum, total, rows_last, sm.red, partial);
}

// ---- the last TWO hops of an explicit frontier through k_expand_mid3 --------------------------------------------------
// identity frontier [base, base + n) as explicit arrays
__global__ __launch_bounds__(256) void k_front_ident(uint32_t base, uint64_t n, uint32_t *__restrict__ fv,
                                                     uint32_t *__restrict__ qlo, uint32_t *__restrict__ qhi) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t v = base + (uint32_t)i;
    const uint64_t q = dig_q((uint64_t)v, 0);
    fv[i] = v;
    qlo[i] = (uint32_t)q;
    qhi[i] = (uint32_t)(q >> 32);
  }
}
__global__ __launch_bounds__(256) void k_front_split(const uint64_t *__restrict__ fq, uint64_t n, uint32_t *__restrict__ qlo,
                                                     uint32_t *__restrict__ qhi) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    qlo[i] = (uint32_t)fq[i];
    qhi[i] = (uint32_t)(fq[i] >> 32);
  }
}
// sorted last vertices -> where each vertex's entries start (froff[0..V]), and the states joined again
__global__ __launch_bounds__(256) void k_front_offsets(const uint32_t *__restrict__ sv, const uint32_t *__restrict__ qlo,
                                                       const uint32_t *__restrict__ qhi, uint64_t n, uint64_t V,
                                                       uint32_t *__restrict__ froff, uint64_t *__restrict__ fq) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const uint64_t cur = i < n ? (uint64_t)sv[i] : V, prev = i ? (uint64_t)sv[i - 1] + 1 : 0;
  for (uint64_t v = prev; v <= cur; v++) froff[v] = (uint32_t)i;  // (vertices without an entry: empty ranges)
  if (i < n) fq[i] = ((uint64_t)qhi[i] << 32) | (uint64_t)qlo[i];
}
// children of reverse entry e = (a -> b): the frontier entries that end in a
__global__ __launch_bounds__(256) void k_front3_prepare(const uint32_t *__restrict__ froff, const uint32_t *__restrict__ rnbr,
                                                        uint64_t n, uint64_t *__restrict__ foff2) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) {
    const uint32_t a = rnbr[e];
    foff2[e] = (uint64_t)(froff[a + 1] - froff[a]);
  }
}

// ---- the last hop of an EXPLICIT frontier as a product (source lists, source ranges, k >= 4) -------------------------
// The walks of the last hop are {prefixes that end in x} x out(x) for every vertex x, whatever the prefixes are: the
// fused kernel above hashes every walk of the last two hops (a full dig_leaf per walk, out(x) re-read per prefix);
// here the second-to-last hop is written out as PAIRS (last vertex x, low half of the prefix's hash state) —
// k_expand_pairs —, the pairs are sorted by x (sort_pairs_by_key: the frontier grouped by its last vertex), and the
// last hop is k_expand_mid2's fold with the states loaded instead of computed: out(x) once per run of equal x, one
// v_xad_u32 per walk (k_expand_front).  Only the low 32 bits of a state reach the 32-bit digest of the last hop.
template <typename OffT>
__global__ __launch_bounds__(XT) void k_expand_pairs(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                     const OffT *__restrict__ foff, Frontier f, uint64_t M,
                                                     const uint32_t *__restrict__ tile_entry, int emit,
                                                     uint32_t *__restrict__ nv, uint32_t *__restrict__ nq32,
                                                     unsigned long long *__restrict__ partial) {
  __shared__ uint64_t s_foff[XT + 1];
  __shared__ uint64_t s_red[12];
  const uint64_t fbase = (uint64_t)foff[0];
  const uint64_t p = fbase + (uint64_t)blockIdx.x * XT + threadIdx.x;
  const uint64_t i0 = tile_entry[blockIdx.x];
  load_window(s_foff, foff, f.n_entries, i0);
  __syncthreads();
  uint64_t sum = 0, rows_last = 0;
  if (p < fbase + M) {
    uint64_t k;
    uint64_t i = locate_entry(s_foff, foff, f.n_entries, i0, p, &k);
    uint32_t v;
    uint64_t q;
    entry_vertex_q(f, i, &v, &q);
    const uint32_t x = nbr[(uint64_t)off[v] + k];
    const uint64_t P = dig_leaf(q, x);
    if (emit) sum = P;
    const uint64_t o = p - fbase;
    nv[o] = x;
    nq32[o] = (uint32_t)dig_q(P, f.j + 1);
    rows_last = (uint64_t)(off[x + 1] - off[x]);
  }
  block_store_partials(sum, 0, rows_last, s_red, partial);
}

__global__ __launch_bounds__(XT) void k_expand_front(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                     const uint32_t *__restrict__ fv, const uint32_t *__restrict__ fq32,
                                                     uint64_t M, unsigned long long *__restrict__ partial) {
  __shared__ MidShared sm;
  uint32_t corr = 0;
  uint32_t acc[MID_R];
#pragma unroll
  for (int r = 0; r < MID_R; r++) acc[r] = 0;
  MidRows rows;
  MidPrep prep;
#pragma unroll
  for (int e = 0; e < MID_EPT; e++) {
    const uint64_t i = (uint64_t)blockIdx.x * MT + (uint64_t)(e * XT) + threadIdx.x;
    rows.valid[e] = i < M;
    rows.x[e] = INVALID_U32;
    rows.u[e] = 0;
    prep.q[e] = 0;
    prep.st[e] = prep.dout[e] = 0;
    if (rows.valid[e]) {
      rows.x[e] = fv[i];
      prep.q[e] = (uint64_t)fq32[i];
      prep.st[e] = off[rows.x[e]];
      prep.dout[e] = off[rows.x[e] + 1] - prep.st[e];
    }
  }
  const uint32_t nruns = mid_stage(sm, rows, prep, M, blockIdx.x);
  mid_fold_tile(sm, nruns, nbr, acc, corr);
  uint32_t tsum = 0;
#pragma unroll
  for (int r = 0; r < MID_R; r++) tsum += acc[r];
  const uint64_t total = (uint64_t)(uint32_t)(tsum - corr);
  block_store_partials(0, total, 0, sm.red, partial);
}

// ---- 3-hop as a product around the LAST inner vertex ----------------------------------------------------------
// Every 3-hop walk u -> a -> b -> w from every vertex is one pair (2-hop row u -> a -> b, out-edge of b).  The
// 2-hop rows that end in b are, for every reverse-CSR entry e = (a -> b) of b, the reverse row of a: a two-level
// flattened index over the reverse entries (foff2[e] = 2-hop rows before entry e), already grouped by b.  A tile
// of MT consecutive positions is hashed into level-2 states (three fmix64 each) and folded against out(b) exactly
// like k_expand_mid2's tiles; the frontier kernels instead hash every 3-hop walk once (k_expand_fused2).
// Runs are long here (sum of in-degrees over in(b), ~8 k states at SF100), so the per-run set-up vanishes.
__global__ __launch_bounds__(256) void k_mid3_prepare(const uint32_t *__restrict__ roff, const uint32_t *__restrict__ rrow,
                                                      const uint32_t *__restrict__ rnbr, uint64_t n,
                                                      uint64_t *__restrict__ foff2,
                                                      unsigned long long *__restrict__ partial) {
  __shared__ uint64_t s_red[12];
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t d1 = 0;
  if (e < n) {
    const uint32_t a = rnbr[e];
    foff2[e] = (uint64_t)(roff[a + 1] - roff[a]);
    d1 = dig_leaf(dig_q((uint64_t)a, 0), rrow[e]);  // the 1-hop row a -> b
  }
  block_store_partials(d1, 0, 0, s_red, partial);
}

__global__ __launch_bounds__(256) void k_tile_partition_mt(const uint64_t *__restrict__ foff, uint64_t n_entries,
                                                           uint64_t n_tiles, uint32_t *__restrict__ tile_entry) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  const uint64_t target = t * MT;
  uint64_t lo = 0, hi = n_entries;  // first idx in [0, n_entries] with foff[idx] > target
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (foff[mid] <= target)
      lo = mid + 1;
    else
      hi = mid;
  }
  tile_entry[t] = (uint32_t)(lo - 1);
}

__global__ __launch_bounds__(XT) void k_expand_mid3(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                    const uint32_t *__restrict__ roff, const uint32_t *__restrict__ rrow,
                                                    const uint32_t *__restrict__ rnbr, const uint64_t *__restrict__ foff2,
                                                    uint64_t n_entries, uint64_t M2,
                                                    const uint32_t *__restrict__ tile_entry, int emit_mid,
                                                    unsigned long long *__restrict__ partial /* of tile0 */,
                                                    uint64_t tile0 /* first tile of this launch */,
                                                    const uint64_t *__restrict__ fq /* explicit frontier, or nullptr */,
                                                    int level) {
  // fq != nullptr: the last TWO hops of an EXPLICIT frontier F_level (source lists, source ranges, k >= 4) in the same
  // product form — `roff` then holds, per vertex a, where the frontier entries that END in a start in `fq` (the
  // frontier sorted by its last vertex: their hash states q_level), so the prefixes one hop longer that end in b are,
  // for every reverse entry (a -> b), those entries; their states are dig_q(dig_leaf(q, b), level + 1).
  __shared__ MidShared sm;
  __shared__ uint64_t s_foff[MT + 1];
  uint64_t mid_sum = 0, rows_last = 0;
  uint32_t acc[MID_R], corr = 0;
#pragma unroll
  for (int r = 0; r < MID_R; r++) acc[r] = 0;
  const uint64_t tile = tile0 + blockIdx.x, i0 = tile_entry[tile];
  for (uint32_t t = threadIdx.x; t <= MT; t += XT) {
    const uint64_t gi = i0 + t;
    s_foff[t] = gi <= n_entries ? foff2[gi] : UINT64_MAX;
  }
  __syncthreads();
  MidRows rows;
  MidPrep prep;
#pragma unroll
  for (int e = 0; e < MID_EPT; e++) {
    const uint64_t p = tile * MT + (uint64_t)(e * XT) + threadIdx.x;
    rows.valid[e] = p < M2;
    rows.x[e] = INVALID_U32;
    rows.u[e] = 0;
    prep.q[e] = 0;
    prep.st[e] = prep.dout[e] = 0;
    if (rows.valid[e]) {
      uint32_t lo = 0, hi = MT + 1;  // first idx in the window with s_foff[idx] > p
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_foff[mid] <= p)
          lo = mid + 1;
        else
          hi = mid;
      }
      uint64_t ent = i0 + lo - 1, start = s_foff[lo - 1];
      if (lo == MT + 1) {  // window exhausted by entries without children: finish in global memory
        uint64_t glo = i0 + MT, ghi = n_entries;
        while (glo < ghi) {
          const uint64_t mid = (glo + ghi) >> 1;
          if (foff2[mid] <= p)
            glo = mid + 1;
          else
            ghi = mid;
        }
        ent = glo - 1;
        start = foff2[ent];
      }
      const uint32_t a = rnbr[ent], b = rrow[ent];
      uint64_t q1;
      if (fq) {
        q1 = fq[(uint64_t)roff[a] + (p - start)];
      } else {
        const uint32_t u = rnbr[(uint64_t)roff[a] + (p - start)];
        q1 = dig_q(dig_leaf(dig_q((uint64_t)u, 0), a), 1);
      }
      const uint64_t P2 = dig_leaf(q1, b);
      if (emit_mid) mid_sum = dsum_add(mid_sum, P2);
      rows.x[e] = b;
      prep.q[e] = dig_q(P2, fq ? level + 1 : 2);
      prep.st[e] = off[b];
      prep.dout[e] = off[b + 1] - prep.st[e];
      rows_last += (uint64_t)prep.dout[e];
    }
  }
  const uint32_t nruns = mid_stage(sm, rows, prep, M2, tile);
  mid_fold_tile(sm, nruns, nbr, acc, corr);
  uint32_t tsum = 0;
#pragma unroll
  for (int r = 0; r < MID_R; r++) tsum += acc[r];
  const uint64_t total = (uint64_t)(uint32_t)(tsum - corr);
  block_store_partials(mid_sum, total, rows_last, sm.red, partial);
}

// per-vertex work of the product kernel: in-degree x (1 + out-degree)
__global__ __launch_bounds__(256) void k_mid_work(const uint32_t *__restrict__ off, const uint32_t *__restrict__ roff,
                                                  uint64_t V, uint64_t *__restrict__ work) {
  const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x < V) work[x] = 1 + (uint64_t)(roff[x + 1] - roff[x]) * (1 + (uint64_t)(off[x + 1] - off[x]));
}

// sum partial[b*4 + c] over b -> out[c]  (c < 3); c = 0,1 are digest sums (lane-wise), c = 2 a count
__global__ __launch_bounds__(256) void k_reduce_partials(const unsigned long long *__restrict__ partial,
                                                         uint64_t nblocks, unsigned long long *__restrict__ out) {
  __shared__ uint64_t s_red[12];
  uint64_t a = 0, b = 0, c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nblocks; i += (uint64_t)gridDim.x * blockDim.x) {
    a = dsum_add(a, partial[i * 4]);
    b = dsum_add(b, partial[i * 4 + 1]);
    c += partial[i * 4 + 2];
  }
  a = wave_reduce_dsum(a);
  b = wave_reduce_dsum(b);
  c = wave_reduce_add_u64(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_red[wave * 3] = a;
    s_red[wave * 3 + 1] = b;
    s_red[wave * 3 + 2] = c;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const uint64_t v0 = s_red[threadIdx.x], v1 = s_red[3 + threadIdx.x], v2 = s_red[6 + threadIdx.x],
                   v3 = s_red[9 + threadIdx.x];
    if (threadIdx.x < 2) {  // digests are u32 sums kept in the low half of a u64 slot: 32-bit atomic, no carry
      const uint64_t sv = dsum_add(dsum_add(v0, v1), dsum_add(v2, v3));
      atomicAdd(reinterpret_cast<unsigned int *>(&out[threadIdx.x]), (unsigned int)sv);
    } else {
      atomicAdd(&out[2], v0 + v1 + v2 + v3);
    }
  }
}

// ---- source list -> frontier 0 -------------------------------------------------------------------
// ballot/popcount compaction of the valid (found) sources; order inside the list is irrelevant
// for a multiset result, so a wave-granular atomic cursor is fine.
__global__ __launch_bounds__(256) void k_compact_sources(const uint32_t *__restrict__ dense, uint64_t n,
                                                         uint32_t *__restrict__ fv, uint64_t *__restrict__ fq,
                                                         uint64_t *__restrict__ fdeg, const uint32_t *__restrict__ off,
                                                         unsigned long long *__restrict__ cursor) {
  // a block takes 1024 consecutive sources (four per thread) and ONE returning atomic on the cursor: one per
  // wavefront was 9 000 same-address round trips for the 577 k Segment seeds of ConnectedSegments SF1024 (112 us)
  __shared__ uint32_t s_cnt[16];
  __shared__ unsigned long long s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t d[4];
  uint64_t m[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + (uint64_t)k * 256 + threadIdx.x;
    d[k] = i < n ? dense[i] : INVALID_U32;
    m[k] = __ballot(d[k] != INVALID_U32);
    if (lane == 0) s_cnt[k * 4 + wave] = (uint32_t)__popcll(m[k]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int q = 0; q < 16; q++) tot += s_cnt[q];
    s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ULL;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (d[k] == INVALID_U32) continue;
    uint64_t o = s_base;
    for (int q = 0; q < k * 4 + wave; q++) o += s_cnt[q];
    o += (uint64_t)__popcll(m[k] & ((1ULL << lane) - 1ULL));
    fv[o] = d[k];
    fq[o] = dig_q((uint64_t)d[k], 0);
    fdeg[o] = (uint64_t)(off[d[k] + 1] - off[d[k]]);
  }
}

// per-vertex 2-hop work estimate for gg_khop_partition: sum over v in adj(u) of (1 + deg(v))
__global__ __launch_bounds__(256) void k_twohop_work(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                     uint64_t V, uint64_t *__restrict__ work) {
  const uint64_t u = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // a wavefront per vertex
  const int lane = threadIdx.x & 63;
  if (u >= V) return;
  uint64_t s = 0;
  for (uint32_t i = off[u] + lane; i < off[u + 1]; i += 64) {
    uint32_t v = nbr[i];
    s += 1 + (uint64_t)(off[v + 1] - off[v]);
  }
  s = wave_reduce_add_u64(s);
  if (lane == 0) work[u] = s;
}

// ---- materialisation -------------------------------------------------------------------------------
// parent table: cols_in[c][i] (c <= j), child row p: copies the parent's columns and appends the child.
template <typename OffT>
__global__ __launch_bounds__(XT) void k_mat_fill(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                 const OffT *__restrict__ foff, uint64_t n_entries, uint64_t M,
                                                 const uint32_t *__restrict__ tile_entry, int j,
                                                 const uint32_t *const *__restrict__ cols_in,
                                                 uint32_t *const *__restrict__ cols_out, uint64_t *__restrict__ ndeg,
                                                 const uint32_t *const *__restrict__ epos_in /* nullable: CSR */,
                                                 uint32_t *const *__restrict__ epos_out /* positions of the edges taken */) {
  __shared__ uint64_t s_foff[XT + 1];
  const uint64_t fbase = (uint64_t)foff[0];
  const uint64_t p = fbase + (uint64_t)blockIdx.x * XT + threadIdx.x;
  const uint64_t i0 = tile_entry[blockIdx.x];
  load_window(s_foff, foff, n_entries, i0);
  __syncthreads();
  if (p >= fbase + M) return;
  uint64_t k;
  uint64_t i = locate_entry(s_foff, foff, n_entries, i0, p, &k);
  const uint32_t v = cols_in[j][i];
  const uint64_t at = (uint64_t)off[v] + k;
  const uint32_t x = nbr[at];
  const uint64_t o = p - fbase;
  for (int c = 0; c <= j; c++) cols_out[c][o] = cols_in[c][i];
  cols_out[j + 1][o] = x;
  if (epos_out) {  // walks with their edges: hop j + 1 took CSR entry `at`
    for (int c = 0; c < j; c++) epos_out[c][o] = epos_in[c][i];
    epos_out[j][o] = (uint32_t)at;
  }
  if (ndeg) ndeg[o] = (uint64_t)(off[x + 1] - off[x]);
}

// Last level of a materialisation WITH edge columns (gg_expand_khop_edges): one thread per output row, vertex ids and
// edge rowids gathered per column.  (The plain path's k_mat_last / k_mat_mid2 are the tuned ones: rows with edge
// columns feed a late join with the edge table's payload columns, typically from a handful of sources.)
struct EdgeCols {
  int64_t *v[GG_MAX_HOPS + 1];
  int64_t *e[GG_MAX_HOPS + 1];
};
template <typename OffT>
__global__ __launch_bounds__(XT) void k_mat_rows_edges(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                       const int64_t *__restrict__ vid, const int64_t *__restrict__ eid,
                                                       const uint32_t *__restrict__ epos, const OffT *__restrict__ foff,
                                                       uint64_t n_entries, uint64_t M,
                                                       const uint32_t *__restrict__ tile_entry, int j,
                                                       const uint32_t *const *__restrict__ cols_in,
                                                       const uint32_t *const *__restrict__ epos_in, EdgeCols out) {
  __shared__ uint64_t s_foff[XT + 1];
  const uint64_t fbase = (uint64_t)foff[0];
  const uint64_t p = fbase + (uint64_t)blockIdx.x * XT + threadIdx.x;
  const uint64_t i0 = tile_entry[blockIdx.x];
  load_window(s_foff, foff, n_entries, i0);
  __syncthreads();
  if (p >= fbase + M) return;
  uint64_t k;
  uint64_t i = locate_entry(s_foff, foff, n_entries, i0, p, &k);
  const uint64_t at = (uint64_t)off[cols_in[j][i]] + k, o = p - fbase;
  auto rowid_of = [&](uint64_t pos) { return eid ? eid[pos] : (int64_t)epos[pos]; };  // explicit rowid, or append position
  for (int c = 0; c <= j; c++) out.v[c][o] = vid[cols_in[c][i]];
  out.v[j + 1][o] = vid[nbr[at]];
  for (int c = 0; c < j; c++) out.e[c][o] = rowid_of(epos_in[c][i]);
  out.e[j][o] = rowid_of(at);
}

// Last level of a materialised expansion, written directly as int64 ids: a tile of XT parent rows is
// staged in LDS (their ids, output offset, out-row extent), then each wavefront takes whole parents and
// streams their children: the child column is a coalesced CSR row read + id gather, the parent columns
// are broadcasts; every store instruction writes 64 consecutive rows (512 contiguous bytes per column).
struct IdCols {
  int64_t *c[GG_MAX_HOPS + 1];
};

__global__ __launch_bounds__(XT) void k_mat_last(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                 const int64_t *__restrict__ vid, const uint64_t *__restrict__ foff,
                                                 uint64_t n_parents, int j /* columns of a parent = j + 1 */,
                                                 const uint32_t *const *__restrict__ cols_in, IdCols out) {
  __shared__ int64_t s_id[GG_MAX_HOPS][XT];
  __shared__ uint64_t s_base[XT];
  __shared__ uint32_t s_start[XT], s_len[XT];
  const uint64_t i = (uint64_t)blockIdx.x * XT + threadIdx.x;
  uint32_t len = 0;
  if (i < n_parents) {
    for (int c = 0; c <= j; c++) s_id[c][threadIdx.x] = vid[cols_in[c][i]];
    const uint32_t v = cols_in[j][i];
    s_start[threadIdx.x] = off[v];
    len = off[v + 1] - off[v];
    s_base[threadIdx.x] = foff[i] - foff[0];
  }
  s_len[threadIdx.x] = len;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < XT; r += XT / 64) {
    const uint32_t n = s_len[r];
    if (n == 0) continue;
    const uint32_t *__restrict__ row = nbr + s_start[r];
    const uint64_t base = s_base[r];
    // Stores are 16 bytes per lane (two consecutive rows of a column), and lane 0 of every store instruction sits on
    // a 128-byte line: the pairs are counted from the line the parent's first row lies in, so an instruction covers
    // eight whole lines except at the two ends of the parent's block (nontemporal stores that start anywhere ran at
    // 4.45 instead of 6.1 TB/s: scripts/ubench_fill.hip).  The first and the last pair may hold one row only.  All
    // loads and id gathers of a trip come first, then its stores: the output is written once, never re-read here,
    // and stores must not serialise the loads.
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    const uint64_t a0 = base & ~15ULL;
    const uint32_t head = (uint32_t)(base - a0), total = head + n;  // row k of the parent is slot head + k behind a0
    for (uint32_t p0 = 0; 2 * p0 < total; p0 += 128) {
      ll2 idv[2];
      bool lov[2], hiv[2];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const uint32_t s0 = 2 * (p0 + q * 64 + lane);
        lov[q] = s0 >= head && s0 < total;
        hiv[q] = s0 + 1 >= head && s0 + 1 < total;
        idv[q].x = lov[q] ? vid[row[s0 - head]] : 0;
        idv[q].y = hiv[q] ? vid[row[s0 + 1 - head]] : 0;
      }
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const uint64_t o = a0 + 2 * (uint64_t)(p0 + q * 64 + lane);  // even: 16-byte aligned
        if (lov[q] && hiv[q]) {
          __builtin_nontemporal_store(idv[q], reinterpret_cast<ll2 *>(&out.c[j + 1][o]));
          for (int c = 0; c <= j; c++) {
            ll2 two;
            two.x = two.y = s_id[c][r];
            __builtin_nontemporal_store(two, reinterpret_cast<ll2 *>(&out.c[c][o]));
          }
        } else if (lov[q] || hiv[q]) {  // an end of the block: one row
          const uint64_t o1 = lov[q] ? o : o + 1;
          out.c[j + 1][o1] = lov[q] ? idv[q].x : idv[q].y;
          for (int c = 0; c <= j; c++) out.c[c][o1] = s_id[c][r];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_gather_ids(const uint32_t *__restrict__ dense, const int64_t *__restrict__ vid,
                                                    uint64_t n, int64_t *__restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = vid[dense[i]];
}

__global__ __launch_bounds__(256) void k_iota_deg(uint32_t base, uint64_t n, const uint32_t *__restrict__ off,
                                                  uint32_t *__restrict__ fv, uint64_t *__restrict__ fdeg) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint32_t v = base + (uint32_t)i;
    fv[i] = v;
    fdeg[i] = (uint64_t)(off[v + 1] - off[v]);
  }
}

}  // namespace gg

// ---------------------------------------------------------------------------------------------------
namespace {

struct DevFrontier {  // explicit frontier living in pool memory
  uint32_t *fv = nullptr;
  uint64_t *fq = nullptr;
  uint64_t *foff = nullptr;  // n+1
  uint64_t n = 0;
  uint64_t M = 0;  // flattened children
};

void free_frontier(gg_ctx *ctx, DevFrontier &f) {
  ctx->dev_free(f.fv);
  ctx->dev_free(f.fq);
  ctx->dev_free(f.foff);
  f = DevFrontier();
}

// read one u64 from device memory (synchronises the stream)
int read_u64(gg_ctx *ctx, const uint64_t *dev, uint64_t *host) {
  GG_HIP(hipMemcpyAsync(ctx->pin_scratch, dev, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  *host = ctx->pin_scratch[0];
  return GG_OK;
}

// degrees (u64, n entries) -> exclusive offsets (n+1 entries), returns total
int offsets_from_deg(gg_ctx *ctx, uint64_t *deg_then_off /* n+1 */, uint64_t n, uint64_t *total_host) {
  uint64_t *tot = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&tot, sizeof(uint64_t)));
  GG_TRY(scan_exclusive_u64(ctx, deg_then_off, deg_then_off, n, tot));
  GG_HIP(hipMemcpyAsync(deg_then_off + n, tot, sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
  GG_TRY(scan_error_fetch(ctx));
  GG_TRY(read_u64(ctx, tot, total_host));
  GG_TRY(scan_error_test(ctx));
  ctx->dev_free(tot);
  return GG_OK;
}

template <typename OffT>
int make_tiles(gg_ctx *ctx, const OffT *foff, uint64_t n_entries, uint64_t M, uint32_t **tile_entry,
               uint64_t *n_tiles) {
  *n_tiles = (M + XT - 1) / XT;
  if (*n_tiles > MAX_GRID_TILES) {  // (callers hand the runtime n_tiles workgroups of XT threads: see MAX_GRID_TILES)
    set_error("expansion of %llu children exceeds one launch (%llu tiles)", (unsigned long long)M,
              (unsigned long long)MAX_GRID_TILES);
    return GG_ERR_TOO_LARGE;
  }
  GG_TRY(ctx->dev_alloc((void **)tile_entry, (*n_tiles ? *n_tiles : 1) * sizeof(uint32_t)));
  if (*n_tiles)
    GG_LAUNCH(ctx, "tile_partition", (k_tile_partition<OffT>), dim3((unsigned)((*n_tiles + 255) / 256)), dim3(256), 0,
              foff, n_entries, *n_tiles, *tile_entry);
  return GG_OK;
}

struct Sums {  // device: [0]=mid digest [1]=last digest [2]=rows_last
  unsigned long long *dev = nullptr;
};

int reduce_partials(gg_ctx *ctx, unsigned long long *partial, uint64_t nblocks, unsigned long long *out3,
                    bool zeroed = false /* the producer kernel cleared out3 */) {
  if (!zeroed) GG_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(unsigned long long), ctx->stream));
  if (nblocks) {
    unsigned grid = (unsigned)((nblocks + 255) / 256);
    if (grid > 64) grid = 64;
    GG_LAUNCH(ctx, "reduce_partials", k_reduce_partials, dim3(grid), dim3(256), 0, partial, nblocks, out3);
  }
  return GG_OK;
}

// count + digest for walks of length k_min..k_max from frontier 0 given either as an identity
// range (csr offsets, u32) or as an explicit frontier (u64 offsets).
int khop_count(gg_ctx *ctx, const gg_csr *csr, bool ident, uint32_t lo, uint64_t n0, uint64_t M1, DevFrontier f0,
               int k_min, int k_max, gg_khop_stats *st) {
  memset(st, 0, sizeof(*st));
  uint64_t walks[GG_MAX_HOPS + 1] = {0};
  uint64_t digests[GG_MAX_HOPS + 1] = {0};
  walks[0] = n0;
  walks[1] = M1;

  unsigned long long *sums = nullptr;  // 3 words per level
  GG_TRY(ctx->dev_alloc((void **)&sums, (GG_MAX_HOPS + 1) * 3 * sizeof(unsigned long long)));
  GG_HIP(hipMemsetAsync(sums, 0, (GG_MAX_HOPS + 1) * 3 * sizeof(unsigned long long), ctx->stream));

  DevFrontier cur = f0;  // j-hop frontier (explicit) — unused while `ident` at j == 0
  bool cur_ident = ident;
  bool own_cur = false;
  int j = 0;
  uint64_t M = M1;  // children of the current frontier = walks[j+1]

  // materialise frontiers until two hops remain
  while (k_max - j > 2 && M > 0) {
    uint32_t *tile_entry = nullptr;
    uint64_t n_tiles = 0;
    DevFrontier nx;
    nx.n = M;
    GG_TRY(ctx->dev_alloc((void **)&nx.fv, M * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&nx.fq, M * sizeof(uint64_t)));
    GG_TRY(ctx->dev_alloc((void **)&nx.foff, (M + 1) * sizeof(uint64_t)));
    unsigned long long *partial = nullptr;
    Frontier fr;
    fr.j = j;
    if (cur_ident) {
      fr.fv = nullptr;
      fr.fq = nullptr;
      fr.n_entries = n0;
      fr.ident_base = lo;
      GG_TRY(make_tiles<uint32_t>(ctx, csr->off + lo, n0, M, &tile_entry, &n_tiles));
      GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
      GG_LAUNCH(ctx, "expand_step", (k_expand_step<uint32_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off,
                csr->nbr, csr->off + lo, fr, M, tile_entry, (int)(j + 1 >= k_min), nx.fv, nx.fq, nx.foff, partial);
    } else {
      fr.fv = cur.fv;
      fr.fq = cur.fq;
      fr.n_entries = cur.n;
      fr.ident_base = 0;
      GG_TRY(make_tiles<uint64_t>(ctx, cur.foff, cur.n, M, &tile_entry, &n_tiles));
      GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
      GG_LAUNCH(ctx, "expand_step", (k_expand_step<uint64_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off,
                csr->nbr, cur.foff, fr, M, tile_entry, (int)(j + 1 >= k_min), nx.fv, nx.fq, nx.foff, partial);
    }
    GG_TRY(reduce_partials(ctx, partial, n_tiles, sums + (j + 1) * 3));
    ctx->dev_free(partial);
    ctx->dev_free(tile_entry);
    uint64_t Mn = 0;
    GG_TRY(offsets_from_deg(ctx, nx.foff, M, &Mn));
    if (own_cur) free_frontier(ctx, cur);
    cur = nx;
    own_cur = true;
    cur_ident = false;
    j++;
    walks[j + 1] = Mn;
    M = Mn;
  }

  // last one or two hops
  // Two hops left and a frontier whose children are a good part of the edge table: the frontier itself (small: one
  // level below its children) is sorted by last vertex, and the last two hops run through k_expand_mid3's tiles with
  // the frontier's entries in place of the reverse rows.  (One pass over all reverse entries to size the tiles: not for
  // small frontiers.)
  if (k_max - j == 2 && ctx->force_frontier != 1 && csr->n_parts <= 1 && csr->V > 1 &&
      ((ctx->force_frontier == 0 && M >= csr->E / 8 + 1) || ctx->force_frontier == 3) && M > 0 &&
      (cur_ident ? n0 : cur.n) < 0xFFFFFFFFull) {
    gg_csr *mcsr = const_cast<gg_csr *>(csr);
    GG_TRY(ensure_reverse(ctx, mcsr));
    const uint64_t nf = cur_ident ? n0 : cur.n, E = csr->E_rev, V = csr->V;
    uint32_t *fv = nullptr, *qlo = nullptr, *qhi = nullptr, *sv = nullptr, *slo = nullptr, *shi = nullptr, *froff = nullptr;
    uint64_t *fqs = nullptr, *foff2 = nullptr;
    for (uint32_t **b : {&qlo, &qhi, &sv, &slo, &shi}) GG_TRY(ctx->dev_alloc((void **)b, nf * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&froff, (V + 1) * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&fqs, nf * sizeof(uint64_t)));
    const unsigned fgrid = (unsigned)((nf + 255) / 256);
    const uint32_t *keys = cur.fv;
    if (cur_ident) {
      GG_TRY(ctx->dev_alloc((void **)&fv, nf * sizeof(uint32_t)));
      hipLaunchKernelGGL(k_front_ident, dim3(fgrid), dim3(256), 0, ctx->stream, lo, nf, fv, qlo, qhi);
      keys = fv;
    } else {
      hipLaunchKernelGGL(k_front_split, dim3(fgrid), dim3(256), 0, ctx->stream, (const uint64_t *)cur.fq, nf, qlo, qhi);
    }
    int key_bits = 1;
    while ((1ull << key_bits) < V) key_bits++;
    GG_TRY(sort_triples_by_key(ctx, keys, qlo, qhi, nf, key_bits, sv, slo, shi));
    hipLaunchKernelGGL(k_front_offsets, dim3((unsigned)((nf + 256) / 256)), dim3(256), 0, ctx->stream, (const uint32_t *)sv,
                       (const uint32_t *)slo, (const uint32_t *)shi, nf, V, froff, fqs);
    GG_TRY(ctx->dev_alloc((void **)&foff2, (E + 1) * sizeof(uint64_t)));
    if (E)
      hipLaunchKernelGGL(k_front3_prepare, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, ctx->stream,
                         (const uint32_t *)froff, (const uint32_t *)csr->rnbr, E, foff2);
    uint64_t M2 = 0;
    GG_TRY(offsets_from_deg(ctx, foff2, E, &M2));
    if (M2 != M) {
      set_error("frontier product: %llu prefixes located through the reverse rows, %llu expected",
                (unsigned long long)M2, (unsigned long long)M);
      return GG_ERR_HIP;
    }
    const uint64_t n_tiles = (M2 + MT - 1) / MT;
    if (n_tiles > 0xFFFFFFFFull) {
      set_error("expansion over %llu prefixes exceeds 2^32 tiles", (unsigned long long)M2);
      return GG_ERR_TOO_LARGE;
    }
    uint32_t *tile_entry = nullptr;
    unsigned long long *partial = nullptr, *tmp = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&tile_entry, n_tiles * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
    GG_TRY(ctx->dev_alloc((void **)&tmp, 3 * sizeof(unsigned long long)));
    GG_LAUNCH(ctx, "tile_partition", k_tile_partition_mt, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0,
              (const uint64_t *)foff2, E, n_tiles, tile_entry);
    const uint64_t per_launch =
        ctx->max_grid_tiles && ctx->max_grid_tiles < MAX_GRID_TILES ? ctx->max_grid_tiles : MAX_GRID_TILES;
    for (uint64_t t0 = 0; t0 < n_tiles; t0 += per_launch) {
      const uint64_t nt = n_tiles - t0 < per_launch ? n_tiles - t0 : per_launch;
      GG_LAUNCH(ctx, "expand_front3", k_expand_mid3, dim3((unsigned)nt), dim3(XT), 0, csr->off, csr->nbr,
                (const uint32_t *)froff, csr->rrow, csr->rnbr, (const uint64_t *)foff2, E, M2, (const uint32_t *)tile_entry,
                (int)(j + 1 >= k_min), partial + t0 * 4, t0, (const uint64_t *)fqs, j);
    }
    GG_TRY(reduce_partials(ctx, partial, n_tiles, tmp));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, tmp, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    digests[j + 1] = ctx->pin_scratch[0];
    digests[j + 2] = ctx->pin_scratch[1];
    walks[j + 2] = ctx->pin_scratch[2];
    for (void *b : {(void *)fv, (void *)qlo, (void *)qhi, (void *)sv, (void *)slo, (void *)shi, (void *)froff, (void *)fqs,
                    (void *)foff2, (void *)tile_entry, (void *)partial, (void *)tmp})
      ctx->dev_free(b);
  } else
  // (two hops left and a frontier worth the three sort passes: the product form — pairs, sort by last vertex, fold)
  if (((ctx->force_frontier == 0 && M >= FRONT_PRODUCT_MIN) || (ctx->force_frontier == 2 && M > 0)) && k_max - j == 2 &&
      csr->V > 1) {
    uint32_t *tile_entry = nullptr, *pv = nullptr, *pq = nullptr, *sv = nullptr, *sq = nullptr;
    uint64_t n_tiles = 0;
    unsigned long long *partial = nullptr, *tmp = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&pv, M * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&pq, M * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&tmp, 6 * sizeof(unsigned long long)));
    Frontier fr;
    fr.j = j;
    if (cur_ident) {
      fr.fv = nullptr;
      fr.fq = nullptr;
      fr.n_entries = n0;
      fr.ident_base = lo;
      GG_TRY(make_tiles<uint32_t>(ctx, csr->off + lo, n0, M, &tile_entry, &n_tiles));
      GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
      GG_LAUNCH(ctx, "expand_pairs", (k_expand_pairs<uint32_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off, csr->nbr,
                csr->off + lo, fr, M, tile_entry, (int)(j + 1 >= k_min), pv, pq, partial);
    } else {
      fr.fv = cur.fv;
      fr.fq = cur.fq;
      fr.n_entries = cur.n;
      fr.ident_base = 0;
      GG_TRY(make_tiles<uint64_t>(ctx, cur.foff, cur.n, M, &tile_entry, &n_tiles));
      GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
      GG_LAUNCH(ctx, "expand_pairs", (k_expand_pairs<uint64_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off, csr->nbr,
                cur.foff, fr, M, tile_entry, (int)(j + 1 >= k_min), pv, pq, partial);
    }
    GG_TRY(reduce_partials(ctx, partial, n_tiles, tmp));  // [0] digest of hop j + 1, [2] rows of the last hop
    ctx->dev_free(partial);
    ctx->dev_free(tile_entry);
    GG_TRY(ctx->dev_alloc((void **)&sv, M * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&sq, M * sizeof(uint32_t)));
    int key_bits = 1;
    while ((1ull << key_bits) < csr->V) key_bits++;
    GG_TRY(sort_pairs_by_key(ctx, pv, pq, M, key_bits, sv, sq));
    const uint64_t f_tiles = (M + MT - 1) / MT;
    GG_TRY(ctx->dev_alloc((void **)&partial, f_tiles * 4 * sizeof(unsigned long long)));
    GG_LAUNCH(ctx, "expand_front", k_expand_front, dim3((unsigned)f_tiles), dim3(XT), 0, csr->off, csr->nbr,
              (const uint32_t *)sv, (const uint32_t *)sq, M, partial);
    GG_TRY(reduce_partials(ctx, partial, f_tiles, tmp + 3));  // [1] digest of the last hop
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, tmp, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    digests[j + 1] = ctx->pin_scratch[0];
    walks[j + 2] = ctx->pin_scratch[2];
    digests[j + 2] = ctx->pin_scratch[4];
    for (void *b : {(void *)partial, (void *)tmp, (void *)pv, (void *)pq, (void *)sv, (void *)sq}) ctx->dev_free(b);
  } else if (M > 0) {
    uint32_t *tile_entry = nullptr;
    uint64_t n_tiles = 0;
    unsigned long long *partial = nullptr;
    Frontier fr;
    fr.j = j;
    const int remaining = k_max - j;  // 1 or 2
    if (cur_ident) {
      fr.fv = nullptr;
      fr.fq = nullptr;
      fr.n_entries = n0;
      fr.ident_base = lo;
      GG_TRY(make_tiles<uint32_t>(ctx, csr->off + lo, n0, M, &tile_entry, &n_tiles));
      GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
      if (remaining == 2)
        GG_LAUNCH(ctx, "expand_fused2", (k_expand_fused2<uint32_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off,
                  csr->nbr, csr->off + lo, fr, M, tile_entry, (int)(j + 1 >= k_min), partial);
      else
        GG_LAUNCH(ctx, "expand_last1", (k_expand_last1<uint32_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off,
                  csr->nbr, csr->off + lo, fr, M, tile_entry, partial);
    } else {
      fr.fv = cur.fv;
      fr.fq = cur.fq;
      fr.n_entries = cur.n;
      fr.ident_base = 0;
      GG_TRY(make_tiles<uint64_t>(ctx, cur.foff, cur.n, M, &tile_entry, &n_tiles));
      GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
      if (remaining == 2)
        GG_LAUNCH(ctx, "expand_fused2", (k_expand_fused2<uint64_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off,
                  csr->nbr, cur.foff, fr, M, tile_entry, (int)(j + 1 >= k_min), partial);
      else
        GG_LAUNCH(ctx, "expand_last1", (k_expand_last1<uint64_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off,
                  csr->nbr, cur.foff, fr, M, tile_entry, partial);
    }
    // partial layout: [0]=digest of hop j+1 (fused2 mid) [1]=digest of the last hop [2]=rows of last hop
    unsigned long long *tmp = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&tmp, 3 * sizeof(unsigned long long)));
    GG_TRY(reduce_partials(ctx, partial, n_tiles, tmp));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, tmp, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    if (remaining == 2) {
      digests[j + 1] = ctx->pin_scratch[0];
      digests[j + 2] = ctx->pin_scratch[1];
      walks[j + 2] = ctx->pin_scratch[2];
    } else {
      digests[j + 1] = ctx->pin_scratch[1];
    }
    ctx->dev_free(tmp);
    ctx->dev_free(partial);
    ctx->dev_free(tile_entry);
  }
  // digests of the materialised intermediate hops
  if (j > 0) {
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, sums, (GG_MAX_HOPS + 1) * 3 * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    for (int h = 1; h <= j; h++) digests[h] = ctx->pin_scratch[h * 3];
  }
  ctx->dev_free(sums);
  if (own_cur) free_frontier(ctx, cur);

  for (int h = 1; h <= k_max; h++) {
    st->traversed_edges += walks[h];
    if (h >= k_min) {
      st->rows[h] = walks[h];
      st->digest[h] = digests[h];
    }
  }
  for (int h = 0; h < k_max; h++) st->frontier_entries += walks[h];
  return GG_OK;
}


// 2-hop count + digest through the product kernel for middle vertices [mid_lo, mid_hi)
// (rows1, rows2, digest1, digest2, traversed edges, frontier entries) of a counting expansion as six u64 words in
// device memory — the vector the ranks of a sharded query add up (duckdb_pgq_amd/sharding.py: FIELDS)
__global__ void k_pack_stats(const unsigned long long *__restrict__ tmp3, unsigned long long M, unsigned long long fe,
                             int with_one_hop, unsigned long long *__restrict__ out6) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned long long dig1 = tmp3 ? tmp3[0] : 0, dig2 = tmp3 ? tmp3[1] : 0, rows2 = tmp3 ? tmp3[2] : 0;
    out6[0] = with_one_hop ? M : 0;
    out6[1] = rows2;
    out6[2] = with_one_hop ? dig1 : 0;
    out6[3] = dig2;
    out6[4] = M + rows2;
    out6[5] = fe;
  }
}

// st == nullptr: nothing comes back to the host and nothing waits — the six words are left in dev_out6
int khop_count_mid(gg_ctx *ctx, gg_csr *csr, uint64_t mid_lo, uint64_t mid_hi, int k_min, gg_khop_stats *st,
                   unsigned long long *dev_out6 = nullptr) {
  if (st) memset(st, 0, sizeof(*st));
  GG_TRY(ensure_reverse(ctx, csr));
  const uint64_t n_mid = mid_hi - mid_lo;
  uint64_t M = csr->E_rev, fbase = 0;
  if (!(mid_lo == 0 && mid_hi == csr->V)) {
    uint32_t ends[2] = {0, 0};
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, csr->roff + mid_lo, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch + 1, csr->roff + mid_hi, sizeof(uint32_t), hipMemcpyDeviceToHost,
                          ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(&ends[0], ctx->pin_scratch, sizeof(uint32_t));
    memcpy(&ends[1], ctx->pin_scratch + 1, sizeof(uint32_t));
    M = (uint64_t)ends[1] - ends[0];
    fbase = ends[0];
  }
  uint64_t rows2 = 0, dig1 = 0, dig2 = 0;
  const uint64_t frontier_entries = (csr->n_parts > 1 ? csr->owned_vertices : n_mid) + M;
  if (dev_out6) {  // device-side result: (the pool keeps freed blocks alive in stream order)
    unsigned long long *partial = nullptr, *tmp = nullptr;
    if (M) {
      const uint64_t n_tiles = (M + MT - 1) / MT;
      GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
      GG_TRY(ctx->dev_alloc((void **)&tmp, 3 * sizeof(unsigned long long)));
      GG_LAUNCH(ctx, "expand_mid2", k_expand_mid2, dim3((unsigned)n_tiles), dim3(XT), 0, csr->off, csr->nbr, csr->rrow,
                csr->rnbr, fbase, M, (int)(k_min <= 1), partial, tmp);
      GG_TRY(reduce_partials(ctx, partial, n_tiles, tmp, true));
    }
    hipLaunchKernelGGL(k_pack_stats, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long *)tmp,
                       (unsigned long long)M, (unsigned long long)frontier_entries, (int)(k_min <= 1), dev_out6);
    ctx->dev_free(tmp);
    ctx->dev_free(partial);
    return GG_OK;
  }
  if (M) {
    const uint64_t n_tiles = (M + MT - 1) / MT;
    unsigned long long *partial = nullptr, *tmp = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
    GG_TRY(ctx->dev_alloc((void **)&tmp, 3 * sizeof(unsigned long long)));
    GG_LAUNCH(ctx, "expand_mid2", k_expand_mid2, dim3((unsigned)n_tiles), dim3(XT), 0, csr->off, csr->nbr, csr->rrow,
              csr->rnbr, fbase, M, (int)(k_min <= 1), partial, tmp);
    GG_TRY(reduce_partials(ctx, partial, n_tiles, tmp, true));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, tmp, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    dig1 = ctx->pin_scratch[0];
    dig2 = ctx->pin_scratch[1];
    rows2 = ctx->pin_scratch[2];
    ctx->dev_free(tmp);
    ctx->dev_free(partial);
  }
  if (k_min <= 1) {
    st->rows[1] = M;
    st->digest[1] = dig1;
  }
  st->rows[2] = rows2;
  st->digest[2] = dig2;
  st->traversed_edges = M + rows2;
  st->frontier_entries = frontier_entries;
  return GG_OK;
}

// 1..3-hop count + digest from every vertex through the product kernels (whole graphs)
int khop_count_mid3(gg_ctx *ctx, gg_csr *csr, int k_min, gg_khop_stats *st) {
  memset(st, 0, sizeof(*st));
  GG_TRY(ensure_reverse(ctx, csr));
  const uint64_t E = csr->E_rev;
  uint64_t M2 = 0, rows3 = 0, dig1 = 0, dig2 = 0, dig3 = 0;
  if (E) {
    uint64_t *foff2 = nullptr;
    unsigned long long *partial = nullptr, *tmp = nullptr;
    const uint64_t nb1 = (E + 255) / 256;
    GG_TRY(ctx->dev_alloc((void **)&foff2, (E + 1) * sizeof(uint64_t)));
    GG_TRY(ctx->dev_alloc((void **)&partial, nb1 * 4 * sizeof(unsigned long long)));
    GG_TRY(ctx->dev_alloc((void **)&tmp, 3 * sizeof(unsigned long long)));
    GG_LAUNCH(ctx, "mid3_prepare", k_mid3_prepare, dim3((unsigned)nb1), dim3(256), 0, csr->roff, csr->rrow, csr->rnbr,
              E, foff2, partial);
    GG_TRY(reduce_partials(ctx, partial, nb1, tmp));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch + 8, tmp, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    GG_TRY(offsets_from_deg(ctx, foff2, E, &M2));  // (synchronises: pin_scratch[8] is the 1-hop digest)
    dig1 = ctx->pin_scratch[8];
    ctx->dev_free(partial);
    partial = nullptr;
    if (M2) {
      const uint64_t n_tiles = (M2 + MT - 1) / MT;
      if (n_tiles > 0xFFFFFFFFull) {  // tile_entry's tile numbers and the tile-partition grid
        set_error("3-hop expansion over %llu 2-hop rows exceeds 2^32 tiles", (unsigned long long)M2);
        return GG_ERR_TOO_LARGE;
      }
      uint32_t *tile_entry = nullptr;
      GG_TRY(ctx->dev_alloc((void **)&tile_entry, n_tiles * sizeof(uint32_t)));
      GG_TRY(ctx->dev_alloc((void **)&partial, n_tiles * 4 * sizeof(unsigned long long)));
      GG_LAUNCH(ctx, "tile_partition", k_tile_partition_mt, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0,
                (const uint64_t *)foff2, E, n_tiles, tile_entry);
      const uint64_t per_launch =
          ctx->max_grid_tiles && ctx->max_grid_tiles < MAX_GRID_TILES ? ctx->max_grid_tiles : MAX_GRID_TILES;
      for (uint64_t t0 = 0; t0 < n_tiles; t0 += per_launch) {  // (one launch up to 1.3e10 2-hop rows)
        const uint64_t nt = n_tiles - t0 < per_launch ? n_tiles - t0 : per_launch;
        GG_LAUNCH(ctx, "expand_mid3", k_expand_mid3, dim3((unsigned)nt), dim3(XT), 0, csr->off, csr->nbr, csr->roff,
                  csr->rrow, csr->rnbr, (const uint64_t *)foff2, E, M2, (const uint32_t *)tile_entry,
                  (int)(k_min <= 2), partial + t0 * 4, t0, (const uint64_t *)nullptr, 0);
      }
      GG_TRY(reduce_partials(ctx, partial, n_tiles, tmp));
      GG_HIP(hipMemcpyAsync(ctx->pin_scratch, tmp, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
      GG_HIP(hipStreamSynchronize(ctx->stream));
      dig2 = ctx->pin_scratch[0];
      dig3 = ctx->pin_scratch[1];
      rows3 = ctx->pin_scratch[2];
      ctx->dev_free(tile_entry);
      ctx->dev_free(partial);
    }
    ctx->dev_free(tmp);
    ctx->dev_free(foff2);
  }
  if (k_min <= 1) {
    st->rows[1] = E;
    st->digest[1] = dig1;
  }
  if (k_min <= 2) {
    st->rows[2] = M2;
    st->digest[2] = dig2;
  }
  st->rows[3] = rows3;
  st->digest[3] = dig3;
  st->traversed_edges = E + M2 + rows3;
  st->frontier_entries = csr->V + E + M2;
  return GG_OK;
}

// ---- 2-hop rows as a join product, materialised ----------------------------------------------------------------
// The materialising counterpart of k_expand_mid2 for walks from EVERY vertex: the rows u -> x -> w through a middle
// vertex x are in(x) x out(x), and the reverse CSR lists the 1-hop rows u -> x grouped by x.  k_mat_last, one parent
// row at a time in source order, translates every child to its id — a random 8-byte gather per OUTPUT row (1.06 G at
// SF10, more time than the stores).  Here the ids of out(x) are gathered once per run of equal x (E gathers in all)
// into LDS, and the rest is stores.  Row order of a materialised result is unspecified (gg.h), so grouping the rows
// by middle vertex is the caller's right.
// The rows of a run's entries i0 .. i1 - 1 are (i1 - i0) blocks of dout rows back to back, the same rows in all three
// columns: row r of that piece belongs to entry r / dout and leaf r % dout (multiply-high by a per-run reciprocal).
// The workgroup walks the piece flat from the 128-byte line its first row lies in, two rows (16 bytes) per lane, so
// EVERY store instruction covers eight whole lines whatever dout and the block boundaries are; only the first and
// last line of a piece are written in part.  That matters more than anything else in this kernel:
// scripts/ubench_fill.hip (profiles/r03_ubench_fill.txt) writes the same three arrays at 6.1 TB/s with line-aligned
// nontemporal 16-byte stores and at 4.45 TB/s when every instruction starts 16 bytes past a line (its end lines
// shared with the neighbours); the earlier form — one store per entry and 128 leaves, starting wherever the entry's
// block starts — ran at 4.8.  Out-rows longer than MAT_CAP leaves go through LDS in chunks, one entry's segment of a
// chunk at a time (>= 16 KB per column: the part-written lines no longer count).
// A workgroup owns MAT_ROWS consecutive OUTPUT rows, not a fixed number of entries (k_mat_tile_entries finds the entry
// each tile starts in): tiles of equal bytes, no heavy tile left over at the end of a launch.
#ifndef GG_MAT_CAP
#define GG_MAT_CAP 2048
#endif
constexpr uint32_t MAT_CAP = GG_MAT_CAP;  // ids of out(x) staged in LDS at a time
#ifndef GG_MAT_ALIGN
#define GG_MAT_ALIGN 16
#endif
constexpr uint32_t MAT_ALIGN = GG_MAT_ALIGN;  // rows

__global__ __launch_bounds__(256) void k_mat_mid2_prepare(const uint32_t *__restrict__ off, const uint32_t *__restrict__ rrow,
                                                          uint64_t e0, uint64_t n, uint64_t *__restrict__ foff) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t x = rrow[e0 + i];
    foff[i] = (uint64_t)(off[x + 1] - off[x]);
  }
}

typedef long long mat_ll2 __attribute__((ext_vector_type(2)));

#ifndef GG_MAT_ROWS
#define GG_MAT_ROWS 32768
#endif
constexpr uint64_t MAT_ROWS = GG_MAT_ROWS;  // output rows per workgroup of k_mat_mid2 (x 24 bytes)

// first output row of tile t (tiles start on 128-byte lines: MAT_ROWS is a multiple of 16 rows)
__device__ __forceinline__ uint64_t mat_tile_start(uint64_t t, uint64_t n_tiles, uint64_t M2) {
  const uint64_t b = t * MAT_ROWS;
  return t < n_tiles && b < M2 ? b : M2;
}
static_assert(MAT_ROWS % 16 == 0, "tiles of k_mat_mid2 start on 128-byte lines");

// tile t of k_mat_mid2 starts in the entry that holds its first output row: upper_bound(foff, start) - 1
__global__ __launch_bounds__(256) void k_mat_tile_entries(const uint64_t *__restrict__ foff, uint64_t n_entries,
                                                          uint64_t n_tiles, uint64_t M2,
                                                          uint32_t *__restrict__ tile_entry) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n_tiles) return;
  const uint64_t target = mat_tile_start(t, n_tiles, M2);
  if (target >= M2) {  // (an empty tile, or the end)
    tile_entry[t] = (uint32_t)n_entries;
    return;
  }
  uint64_t lo = 0, hi = n_entries;  // first idx in [0, n_entries] with foff[idx] > target
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (foff[mid] <= target)
      lo = mid + 1;
    else
      hi = mid;
  }
  tile_entry[t] = (uint32_t)(lo - 1);
}

// rows S + r, r in [r_lo, r_hi), of the three columns: row r = (entry ib + r / d, middle id, leaf r % d); d >= 1,
// r_hi <= 2^32 / d.  The whole workgroup; uid / oid are the batch's entry ids and the staged leaf ids in LDS.
__device__ __forceinline__ void mat_write_rows(uint64_t S, uint32_t r_lo, uint32_t r_hi, uint32_t d, uint32_t ib,
                                               long long xid, const int64_t *uid, const int64_t *oid,
                                               int64_t *__restrict__ c0, int64_t *__restrict__ c1,
                                               int64_t *__restrict__ c2) {
  // first row of the 128-byte line the first row lies in (the columns are 256-byte aligned)
  const uint64_t A0 = (S + r_lo) & ~(uint64_t)(MAT_ALIGN - 1);
  const uint32_t head = (uint32_t)(S + r_lo - A0), total = head + (r_hi - r_lo);
  const uint32_t m = d >= 2 ? 0xFFFFFFFFu / d + 1u : 0u;  // r / d = umulhi(r, m) for r < 2^32 / d
  mat_ll2 xx;
  xx.x = xx.y = xid;
  for (uint32_t q = threadIdx.x; 2 * q < total; q += 256) {
    const uint32_t hi = 2 * q + 1;  // the lane's pair: rows A0 + 2q (low half) and A0 + hi (high half)
    if (hi < head) continue;        // both before the first row
    const uint32_t rh = r_lo + (hi - head);
    const bool lo_ok = hi > head, hi_ok = rh < r_hi;
    const uint32_t ih = d >= 2 ? __umulhi(rh, m) : rh, jh = rh - ih * d;
    const uint32_t il = jh ? ih : ih - 1u, jl = jh ? jh - 1u : d - 1u;
    mat_ll2 uu, w;
    uu.y = uid[ib + ih];  // (rh == r_hi may read one element past the piece: the arrays are padded, nothing is stored)
    w.y = oid[jh];
    uu.x = lo_ok ? uid[ib + il] : 0;
    w.x = lo_ok ? oid[jl] : 0;
    const uint64_t o = A0 + 2 * (uint64_t)q;
    if (lo_ok && hi_ok) {
      __builtin_nontemporal_store(uu, reinterpret_cast<mat_ll2 *>(c0 + o));
      __builtin_nontemporal_store(xx, reinterpret_cast<mat_ll2 *>(c1 + o));
      __builtin_nontemporal_store(w, reinterpret_cast<mat_ll2 *>(c2 + o));
    } else if (lo_ok) {  // last row, alone
      c0[o] = uu.x;
      c1[o] = xid;
      c2[o] = w.x;
    } else if (hi_ok) {  // first row, alone
      c0[o + 1] = uu.y;
      c1[o + 1] = xid;
      c2[o + 1] = w.y;
    }
  }
}

// One workgroup per MAT_ROWS consecutive OUTPUT rows (tiles of equal bytes: a tile of 256 reverse entries is anything
// between nothing and 256 x the largest out-degree rows, and the few heaviest were the kernel's tail); the entries
// that overlap the tile are taken 256 at a time, their pieces clipped to the tile.
__global__ __launch_bounds__(256) void k_mat_mid2(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                  const uint32_t *__restrict__ rrow, const uint32_t *__restrict__ rnbr,
                                                  const int64_t *__restrict__ vid, const uint64_t *__restrict__ foff,
                                                  const uint32_t *__restrict__ tile_entry, uint64_t e0, uint64_t n,
                                                  uint64_t M2, int64_t *__restrict__ c0, int64_t *__restrict__ c1,
                                                  int64_t *__restrict__ c2, uint32_t n_tiles) {
  __shared__ int64_t s_uid[256 + 1];
  __shared__ int64_t s_oid[MAT_CAP + 1];
  __shared__ uint64_t s_base[256];
  __shared__ uint32_t s_x[256], s_run[257], s_wcnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (tiles in launch order: a scattered order — the resident workgroups writing at 16 or 256 places spread over every
  // column instead of one moving window — changed nothing; what matters is which memory ranks the columns lie in, §4.2)
  const uint32_t tile = blockIdx.x;
  const uint64_t t_lo = mat_tile_start(tile, n_tiles, M2), t_hi = mat_tile_start(tile + 1ull, n_tiles, M2);
  const uint64_t e_lo = tile_entry[tile];
  uint64_t e_hi = (uint64_t)tile_entry[tile + 1] + 1;  // (that entry holds the next tile's first row, and maybe ours)
  e_hi = e_hi < n ? e_hi : n;
  for (uint64_t eb = e_lo; eb < e_hi; eb += 256) {  // (everything about the control flow is uniform over the workgroup)
    const uint64_t p = eb + threadIdx.x;
    const bool valid = p < e_hi;
    uint32_t x = INVALID_U32;
    __syncthreads();  // the batch before is done with the arrays
    if (valid) {
      x = rrow[e0 + p];
      s_uid[threadIdx.x] = vid[rnbr[e0 + p]];
      s_base[threadIdx.x] = foff[p];
    }
    s_x[threadIdx.x] = x;
    __syncthreads();
    const bool head = valid && (threadIdx.x == 0 || s_x[threadIdx.x - 1] != x);
    const uint64_t hm = __ballot(head);
    if (lane == 0) s_wcnt[wave] = (uint32_t)__popcll(hm);
    __syncthreads();
    uint32_t before = 0, nruns = 0;
    for (int q = 0; q < 4; q++) {
      if (q < wave) before += s_wcnt[q];
      nruns += s_wcnt[q];
    }
    if (head) s_run[before + __popcll(hm & ((1ULL << lane) - 1ULL))] = threadIdx.x;
    if (threadIdx.x == 0) s_run[nruns] = e_hi - eb < 256 ? (uint32_t)(e_hi - eb) : 256u;
    __syncthreads();
    for (uint32_t r = 0; r < nruns; r++) {
      const uint32_t i0 = s_run[r], i1 = s_run[r + 1];
      const uint32_t xr = s_x[i0], st = off[xr], dout = off[xr + 1] - st;
      if (dout == 0) continue;
      // the piece: (i1 - i0) blocks of dout rows from Bs on; [lo, hi) is its part inside the tile
      const uint64_t Bs = s_base[i0], Be = Bs + (uint64_t)(i1 - i0) * dout;
      const uint64_t lo = Bs > t_lo ? Bs : t_lo, hi = Be < t_hi ? Be : t_hi;
      if (lo >= hi) continue;
      const long long xid = vid[xr];
      for (uint32_t jc = 0; jc < dout; jc += MAT_CAP) {
        const uint32_t clen = dout - jc < MAT_CAP ? dout - jc : MAT_CAP;
        __syncthreads();  // the readers of the chunk before are done
        for (uint32_t j = threadIdx.x; j < clen; j += 256) s_oid[j] = vid[nbr[st + jc + j]];
        __syncthreads();
        if (dout <= MAT_CAP) {  // the whole piece in one go (256 entries x MAT_CAP leaves <= 2^32 / dout rows)
          mat_write_rows(Bs, (uint32_t)(lo - Bs), (uint32_t)(hi - Bs), dout, i0, xid, s_uid, s_oid, c0, c1, c2);
        } else {  // long out-rows: the chunk's segment of one entry at a time
          for (uint32_t i = i0; i < i1; i++) {
            const uint64_t S = s_base[i] + jc, slo = S > t_lo ? S : t_lo, shi = S + clen < t_hi ? S + clen : t_hi;
            if (slo < shi)
              mat_write_rows(S, (uint32_t)(slo - S), (uint32_t)(shi - S), clen, i, xid, s_uid, s_oid, c0, c1, c2);
          }
        }
      }
    }
  }
}

// ---- the LAST level of any materialised walk as a product (source lists, k >= 3) -----------------------------------
// k_mat_last translates every child to its id with a gather per OUTPUT row.  The rows of the last level are
// {rows of level k - 1 that end in x} x out(x): with the level k - 1 table sorted by its last column (a permutation:
// sort_pairs_by_key of (last vertex, row number)) the ids of out(x) are gathered once per run of equal x, the P prefix
// ids once per level-(k-1) row, and the rest is k_mat_mid2's line-aligned stores — P + 2 columns instead of 3.
struct MatFrontArgs {
  const uint32_t *pcol[GG_MAX_HOPS];  // the P prefix columns of level k - 1 (dense indices), in walk order
  int64_t *out[GG_MAX_HOPS + 1];      // the P + 2 id columns of level k
};
constexpr int MAT_FRONT_MAX_P = 3;    // walks of up to 4 hops (P = k - 1); longer ones keep k_mat_last

template <int P>
__device__ __forceinline__ void mat_write_rows_p(uint64_t S, uint32_t r_lo, uint32_t r_hi, uint32_t d, uint32_t ib,
                                                 long long xid, const int64_t (*uid)[256 + 1], const int64_t *oid,
                                                 const MatFrontArgs &a) {
  const uint64_t A0 = (S + r_lo) & ~(uint64_t)(MAT_ALIGN - 1);
  const uint32_t head = (uint32_t)(S + r_lo - A0), total = head + (r_hi - r_lo);
  const uint32_t m = d >= 2 ? 0xFFFFFFFFu / d + 1u : 0u;  // r / d = umulhi(r, m) for r < 2^32 / d
  mat_ll2 xx;
  xx.x = xx.y = xid;
  for (uint32_t q = threadIdx.x; 2 * q < total; q += 256) {
    const uint32_t hi = 2 * q + 1;
    if (hi < head) continue;
    const uint32_t rh = r_lo + (hi - head);
    const bool lo_ok = hi > head, hi_ok = rh < r_hi;
    const uint32_t ih = d >= 2 ? __umulhi(rh, m) : rh, jh = rh - ih * d;
    const uint32_t il = jh ? ih : ih - 1u, jl = jh ? jh - 1u : d - 1u;
    mat_ll2 w;
    w.y = oid[jh];
    w.x = lo_ok ? oid[jl] : 0;
    const uint64_t o = A0 + 2 * (uint64_t)q;
#pragma unroll
    for (int c = 0; c < P; c++) {
      mat_ll2 uu;
      uu.y = uid[c][ib + ih];
      uu.x = lo_ok ? uid[c][ib + il] : 0;
      if (lo_ok && hi_ok)
        __builtin_nontemporal_store(uu, reinterpret_cast<mat_ll2 *>(a.out[c] + o));
      else if (lo_ok)
        a.out[c][o] = uu.x;
      else if (hi_ok)
        a.out[c][o + 1] = uu.y;
    }
    if (lo_ok && hi_ok) {
      __builtin_nontemporal_store(xx, reinterpret_cast<mat_ll2 *>(a.out[P] + o));
      __builtin_nontemporal_store(w, reinterpret_cast<mat_ll2 *>(a.out[P + 1] + o));
    } else if (lo_ok) {
      a.out[P][o] = xid;
      a.out[P + 1][o] = w.x;
    } else if (hi_ok) {
      a.out[P][o + 1] = xid;
      a.out[P + 1][o + 1] = w.y;
    }
  }
}

template <int P>
__global__ __launch_bounds__(256) void k_mat_front(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                   const uint32_t *__restrict__ skey, const uint32_t *__restrict__ perm,
                                                   const int64_t *__restrict__ vid, const uint64_t *__restrict__ foff,
                                                   const uint32_t *__restrict__ tile_entry, uint64_t n, uint64_t M2,
                                                   MatFrontArgs a, uint32_t n_tiles) {
  __shared__ int64_t s_uid[P ? P : 1][256 + 1];
  __shared__ int64_t s_oid[MAT_CAP + 1];
  __shared__ uint64_t s_base[256];
  __shared__ uint32_t s_x[256], s_run[257], s_wcnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x;
  const uint64_t t_lo = mat_tile_start(tile, n_tiles, M2), t_hi = mat_tile_start(tile + 1ull, n_tiles, M2);
  const uint64_t e_lo = tile_entry[tile];
  uint64_t e_hi = (uint64_t)tile_entry[tile + 1] + 1;
  e_hi = e_hi < n ? e_hi : n;
  for (uint64_t eb = e_lo; eb < e_hi; eb += 256) {
    const uint64_t p = eb + threadIdx.x;
    const bool valid = p < e_hi;
    uint32_t x = INVALID_U32;
    __syncthreads();
    if (valid) {
      x = skey[p];
      const uint32_t row = perm[p];
#pragma unroll
      for (int c = 0; c < P; c++) s_uid[c][threadIdx.x] = vid[a.pcol[c][row]];
      s_base[threadIdx.x] = foff[p];
    }
    s_x[threadIdx.x] = x;
    __syncthreads();
    const bool head = valid && (threadIdx.x == 0 || s_x[threadIdx.x - 1] != x);
    const uint64_t hm = __ballot(head);
    if (lane == 0) s_wcnt[wave] = (uint32_t)__popcll(hm);
    __syncthreads();
    uint32_t before = 0, nruns = 0;
    for (int q = 0; q < 4; q++) {
      if (q < wave) before += s_wcnt[q];
      nruns += s_wcnt[q];
    }
    if (head) s_run[before + __popcll(hm & ((1ULL << lane) - 1ULL))] = threadIdx.x;
    if (threadIdx.x == 0) s_run[nruns] = e_hi - eb < 256 ? (uint32_t)(e_hi - eb) : 256u;
    __syncthreads();
    for (uint32_t r = 0; r < nruns; r++) {
      const uint32_t i0 = s_run[r], i1 = s_run[r + 1];
      const uint32_t xr = s_x[i0], st = off[xr], dout = off[xr + 1] - st;
      if (dout == 0) continue;
      const uint64_t Bs = s_base[i0], Be = Bs + (uint64_t)(i1 - i0) * dout;
      const uint64_t lo = Bs > t_lo ? Bs : t_lo, hi = Be < t_hi ? Be : t_hi;
      if (lo >= hi) continue;
      const long long xid = vid[xr];
      for (uint32_t jc = 0; jc < dout; jc += MAT_CAP) {
        const uint32_t clen = dout - jc < MAT_CAP ? dout - jc : MAT_CAP;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < clen; j += 256) s_oid[j] = vid[nbr[st + jc + j]];
        __syncthreads();
        if (dout <= MAT_CAP) {
          mat_write_rows_p<P>(Bs, (uint32_t)(lo - Bs), (uint32_t)(hi - Bs), dout, i0, xid, s_uid, s_oid, a);
        } else {
          for (uint32_t i = i0; i < i1; i++) {
            const uint64_t S = s_base[i] + jc, slo = S > t_lo ? S : t_lo, shi = S + clen < t_hi ? S + clen : t_hi;
            if (slo < shi) mat_write_rows_p<P>(S, (uint32_t)(slo - S), (uint32_t)(shi - S), clen, i, xid, s_uid, s_oid, a);
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_iota_u32(uint64_t n, uint32_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint32_t)i;
}

// children of the sorted level-(k-1) row p: the out-degree of its last vertex
__global__ __launch_bounds__(256) void k_mat_front_prepare(const uint32_t *__restrict__ off, const uint32_t *__restrict__ skey,
                                                           uint64_t n, uint64_t *__restrict__ foff) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) foff[i] = (uint64_t)(off[skey[i] + 1] - off[skey[i]]);
}

// the last level (h = P + 1 hops, h + 1 id columns in out_cols) of a materialisation as a product; M: its rows
int khop_materialise_front(gg_ctx *ctx, const gg_csr *csr, const std::vector<uint32_t *> &prev_cols, uint64_t n_prev,
                           uint64_t M, int64_t *const *out_cols) {
  const int P = (int)prev_cols.size() - 1;
  uint32_t *iota = nullptr, *skey = nullptr, *perm = nullptr, *tile_entry = nullptr;
  uint64_t *foff = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&iota, n_prev * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&skey, n_prev * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&perm, n_prev * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&foff, (n_prev + 1) * sizeof(uint64_t)));
  const unsigned grid = (unsigned)((n_prev + 255) / 256);
  hipLaunchKernelGGL(k_iota_u32, dim3(grid), dim3(256), 0, ctx->stream, n_prev, iota);
  int key_bits = 1;
  while ((1ull << key_bits) < csr->V) key_bits++;
  GG_TRY(sort_pairs_by_key(ctx, prev_cols[P], iota, n_prev, key_bits, skey, perm));
  hipLaunchKernelGGL(k_mat_front_prepare, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t *)csr->off,
                     (const uint32_t *)skey, n_prev, foff);
  uint64_t M2 = 0;
  GG_TRY(offsets_from_deg(ctx, foff, n_prev, &M2));
  if (M2 != M) {
    set_error("materialisation: %llu rows behind the sorted level, %llu expected", (unsigned long long)M2,
              (unsigned long long)M);
    return GG_ERR_HIP;
  }
  const uint64_t n_tiles = (M2 + MAT_ROWS - 1) / MAT_ROWS;
  GG_TRY(ctx->dev_alloc((void **)&tile_entry, (n_tiles + 1) * sizeof(uint32_t)));
  GG_LAUNCH(ctx, "mat_tile_entries", k_mat_tile_entries, dim3((unsigned)((n_tiles + 256) / 256)), dim3(256), 0,
            (const uint64_t *)foff, n_prev, n_tiles, M2, tile_entry);
  MatFrontArgs a;
  for (int c = 0; c < GG_MAX_HOPS; c++) a.pcol[c] = c < P ? prev_cols[c] : nullptr;
  for (int c = 0; c <= GG_MAX_HOPS; c++) a.out[c] = c <= P + 1 ? out_cols[c] : nullptr;
#define GG_MAT_FRONT(PP)                                                                                              \
  GG_LAUNCH(ctx, "mat_front", (k_mat_front<PP>), dim3((unsigned)n_tiles), dim3(256), 0, csr->off, csr->nbr,           \
            (const uint32_t *)skey, (const uint32_t *)perm, csr->vid, (const uint64_t *)foff,                        \
            (const uint32_t *)tile_entry, n_prev, M2, a, (uint32_t)n_tiles)
  switch (P) {
  case 0: GG_MAT_FRONT(0); break;
  case 1: GG_MAT_FRONT(1); break;
  case 2: GG_MAT_FRONT(2); break;
  default: GG_MAT_FRONT(3); break;
  }
#undef GG_MAT_FRONT
  for (void *b : {(void *)iota, (void *)skey, (void *)perm, (void *)foff, (void *)tile_entry}) ctx->dev_free(b);
  return GG_OK;
}

// 2-hop rows (and, with k_min == 1, the 1-hop rows) whose MIDDLE vertex lies in [mid_lo, mid_hi), through
// k_mat_mid2; the whole graph for [0, V).  The 1-hop table of a middle range holds the edges INTO the range.
int khop_materialise_mid2(gg_ctx *ctx, gg_csr *csr, uint64_t mid_lo, uint64_t mid_hi, int k_min, gg_result *res) {
  GG_TRY(ensure_reverse(ctx, csr));
  uint64_t e0 = 0, n = csr->E_rev;
  if (!(mid_lo == 0 && mid_hi == csr->V)) {
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, csr->roff + mid_lo, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch + 1, csr->roff + mid_hi, sizeof(uint32_t), hipMemcpyDeviceToHost,
                          ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    uint32_t ends[2];
    memcpy(&ends[0], ctx->pin_scratch, sizeof(uint32_t));
    memcpy(&ends[1], ctx->pin_scratch + 1, sizeof(uint32_t));
    e0 = ends[0];
    n = (uint64_t)ends[1] - ends[0];
  }
  uint64_t *foff = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&foff, (n + 1) * sizeof(uint64_t)));
  if (n)
    GG_LAUNCH(ctx, "mat_mid2_prepare", k_mat_mid2_prepare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, csr->off,
              csr->rrow, e0, n, foff);
  uint64_t M2 = 0;
  GG_TRY(offsets_from_deg(ctx, foff, n, &M2));
  res->rows[2] = M2;
  {
    // the three columns: one pooled block when small; when large, three blocks the pool chose to lie in different
    // memory ranks — lockstep stores into one rank run at 5.8 TB/s, into two or three at 7.0-7.2 (gg_runtime.hip
    // "Placement of large result columns")
    void *cols[3] = {nullptr, nullptr, nullptr};
    GG_TRY(ctx->dev_alloc_columns(cols, (M2 ? M2 : 1) * sizeof(int64_t)));
    for (int c = 0; c <= 2; c++) {
      res->cols[2][c] = reinterpret_cast<int64_t *>(cols[c]);
      ctx->keep(cols[c]);
    }
  }
  if (M2) {
    const uint64_t n_tiles = (M2 + MAT_ROWS - 1) / MAT_ROWS;
    uint32_t *tile_entry = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&tile_entry, (n_tiles + 1) * sizeof(uint32_t)));
    GG_LAUNCH(ctx, "mat_tile_entries", k_mat_tile_entries, dim3((unsigned)((n_tiles + 256) / 256)), dim3(256), 0,
              (const uint64_t *)foff, n, n_tiles, M2, tile_entry);
    GG_LAUNCH(ctx, "mat_mid2", k_mat_mid2, dim3((unsigned)n_tiles), dim3(256), 0, csr->off, csr->nbr, csr->rrow,
              csr->rnbr, csr->vid, (const uint64_t *)foff, (const uint32_t *)tile_entry, e0, n, M2, res->cols[2][0],
              res->cols[2][1], res->cols[2][2], (uint32_t)n_tiles);
    ctx->dev_free(tile_entry);
  }
  if (k_min <= 1) {
    res->rows[1] = n;
    for (int c = 0; c <= 1; c++) {
      GG_TRY(ctx->dev_alloc((void **)&res->cols[1][c], (n ? n : 1) * sizeof(int64_t)));
      ctx->keep(res->cols[1][c]);
      if (n)
        GG_LAUNCH(ctx, "gather_ids", k_gather_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                  (c ? csr->rrow : csr->rnbr) + e0, csr->vid, n, res->cols[1][c]);
    }
  }
  ctx->dev_free(foff);
  return GG_OK;
}

// materialise walks as int64 id columns (correctness config; level-by-level)
int khop_materialise(gg_ctx *ctx, const gg_csr *csr, const uint32_t *fv0, uint64_t n0, int k_min, int k_max,
                     gg_result *res, bool with_edges = false) {
  // level tables of dense columns (with_edges: and of the CSR positions of the edges taken, one column per hop)
  std::vector<uint32_t *> cols_prev, cols_cur, epos_prev, epos_cur;
  const uint32_t **e_in = nullptr;
  uint32_t **e_out = nullptr;
  if (with_edges) {
    GG_TRY(ctx->dev_alloc((void **)&e_in, (GG_MAX_HOPS + 1) * sizeof(void *)));
    GG_TRY(ctx->dev_alloc((void **)&e_out, (GG_MAX_HOPS + 1) * sizeof(void *)));
  }
  uint64_t n_prev = n0;
  uint32_t *c0 = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&c0, (n0 ? n0 : 1) * sizeof(uint32_t)));
  if (n0) GG_HIP(hipMemcpyAsync(c0, fv0, n0 * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
  cols_prev.push_back(c0);
  // degrees of level-0 entries
  uint64_t *foff = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&foff, (n0 + 1) * sizeof(uint64_t)));
  {
    uint32_t *tmpv = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&tmpv, (n0 ? n0 : 1) * sizeof(uint32_t)));
    uint64_t *cursor = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&cursor, sizeof(uint64_t)));
    GG_HIP(hipMemsetAsync(cursor, 0, sizeof(uint64_t), ctx->stream));
    uint64_t *fq = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&fq, (n0 ? n0 : 1) * sizeof(uint64_t)));
    // fv0 holds only valid dense indices, so compaction is the identity here; reuse it for degrees
    if (n0)
      GG_LAUNCH(ctx, "compact_sources", k_compact_sources, dim3((unsigned)((n0 + 1023) / 1024)), dim3(256), 0, fv0, n0,
                tmpv, fq, foff, csr->off, (unsigned long long *)cursor);
    // compaction may permute entries across waves: take the permuted list as level 0
    if (n0) GG_HIP(hipMemcpyAsync(c0, tmpv, n0 * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    ctx->dev_free(tmpv);
    ctx->dev_free(cursor);
    ctx->dev_free(fq);
  }
  uint64_t M = 0;
  GG_TRY(offsets_from_deg(ctx, foff, n0, &M));

  const uint32_t **d_in = nullptr;
  uint32_t **d_out = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&d_in, (GG_MAX_HOPS + 1) * sizeof(void *)));
  GG_TRY(ctx->dev_alloc((void **)&d_out, (GG_MAX_HOPS + 1) * sizeof(void *)));

  for (int h = 1; h <= k_max; h++) {
    if (h == k_max) {
      // last level: straight to int64 id columns, no dense intermediate
      res->rows[h] = M;
      IdCols oc;
      for (int c = 0; c <= GG_MAX_HOPS; c++) oc.c[c] = nullptr;
      if (h == 2) {  // three columns written in lockstep: a placed set when they are large (gg_runtime.hip "Placement")
        void *three[3] = {nullptr, nullptr, nullptr};
        GG_TRY(ctx->dev_alloc_columns(three, (M ? M : 1) * sizeof(int64_t)));
        for (int c = 0; c <= h; c++) res->cols[h][c] = reinterpret_cast<int64_t *>(three[c]);
      } else {
        for (int c = 0; c <= h; c++) GG_TRY(ctx->dev_alloc((void **)&res->cols[h][c], (M ? M : 1) * sizeof(int64_t)));
      }
      for (int c = 0; c <= h; c++) {
        ctx->keep(res->cols[h][c]);
        oc.c[c] = res->cols[h][c];
      }
      if (M && with_edges) {
        EdgeCols ec;
        for (int c = 0; c <= GG_MAX_HOPS; c++) ec.v[c] = ec.e[c] = nullptr;
        for (int c = 0; c <= h; c++) ec.v[c] = res->cols[h][c];
        for (int c = 0; c < h; c++) {
          GG_TRY(ctx->dev_alloc((void **)&res->ecols[h][c], M * sizeof(int64_t)));
          ctx->keep(res->ecols[h][c]);
          ec.e[c] = res->ecols[h][c];
        }
        uint32_t *tile_entry = nullptr;
        uint64_t n_tiles = 0;
        GG_TRY(make_tiles<uint64_t>(ctx, foff, n_prev, M, &tile_entry, &n_tiles));
        GG_HIP(hipMemcpyAsync(d_in, cols_prev.data(), cols_prev.size() * sizeof(void *), hipMemcpyHostToDevice,
                              ctx->stream));
        if (!epos_prev.empty())
          GG_HIP(hipMemcpyAsync(e_in, epos_prev.data(), epos_prev.size() * sizeof(void *), hipMemcpyHostToDevice,
                                ctx->stream));
        GG_HIP(hipStreamSynchronize(ctx->stream));
        GG_LAUNCH(ctx, "mat_rows_edges", (k_mat_rows_edges<uint64_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off,
                  csr->nbr, csr->vid, (const int64_t *)csr->eid, (const uint32_t *)csr->epos, (const uint64_t *)foff, n_prev, M,
                  (const uint32_t *)tile_entry, h - 1, d_in, e_in, ec);
        ctx->dev_free(tile_entry);
      } else if (M && (int)cols_prev.size() - 1 <= MAT_FRONT_MAX_P && n_prev < 0xFFFFFFFFull && csr->V > 1 &&
                 ((ctx->force_frontier == 0 && M >= MAT_FRONT_MIN) || ctx->force_frontier >= 2)) {
        // the product form: level h - 1 sorted by its last vertex, out-rows' ids gathered once per run
        GG_TRY(khop_materialise_front(ctx, csr, cols_prev, n_prev, M, res->cols[h]));
      } else if (M) {
        GG_HIP(hipMemcpyAsync(d_in, cols_prev.data(), cols_prev.size() * sizeof(void *), hipMemcpyHostToDevice,
                              ctx->stream));
        GG_HIP(hipStreamSynchronize(ctx->stream));
        GG_LAUNCH(ctx, "mat_last", k_mat_last, dim3((unsigned)((n_prev + XT - 1) / XT)), dim3(XT), 0, csr->off, csr->nbr,
                  csr->vid, foff, n_prev, h - 1, d_in, oc);
      }
      break;
    }
    cols_cur.assign((size_t)h + 1, nullptr);
    for (int c = 0; c <= h; c++) GG_TRY(ctx->dev_alloc((void **)&cols_cur[c], (M ? M : 1) * sizeof(uint32_t)));
    epos_cur.assign(with_edges ? (size_t)h : 0, nullptr);
    for (auto &pc : epos_cur) GG_TRY(ctx->dev_alloc((void **)&pc, (M ? M : 1) * sizeof(uint32_t)));
    uint64_t *noff = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&noff, (M + 1) * sizeof(uint64_t)));
    uint64_t Mn = 0;
    if (M) {
      uint32_t *tile_entry = nullptr;
      uint64_t n_tiles = 0;
      GG_TRY(make_tiles<uint64_t>(ctx, foff, n_prev, M, &tile_entry, &n_tiles));
      GG_HIP(hipMemcpyAsync(d_in, cols_prev.data(), cols_prev.size() * sizeof(void *), hipMemcpyHostToDevice,
                            ctx->stream));
      GG_HIP(hipMemcpyAsync(d_out, cols_cur.data(), cols_cur.size() * sizeof(void *), hipMemcpyHostToDevice,
                            ctx->stream));
      if (with_edges) {
        if (!epos_prev.empty())
          GG_HIP(hipMemcpyAsync(e_in, epos_prev.data(), epos_prev.size() * sizeof(void *), hipMemcpyHostToDevice,
                                ctx->stream));
        GG_HIP(hipMemcpyAsync(e_out, epos_cur.data(), epos_cur.size() * sizeof(void *), hipMemcpyHostToDevice,
                              ctx->stream));
      }
      GG_HIP(hipStreamSynchronize(ctx->stream));  // host vectors are reused below
      GG_LAUNCH(ctx, "mat_fill", (k_mat_fill<uint64_t>), dim3((unsigned)n_tiles), dim3(XT), 0, csr->off, csr->nbr,
                foff, n_prev, M, tile_entry, h - 1, d_in, d_out, noff, with_edges ? e_in : (const uint32_t **)nullptr,
                with_edges ? e_out : (uint32_t **)nullptr);
      ctx->dev_free(tile_entry);
      GG_TRY(offsets_from_deg(ctx, noff, M, &Mn));
    }
    if (h >= k_min) {  // convert this level to int64 ids
      res->rows[h] = M;
      for (int c = 0; c <= h; c++) {
        GG_TRY(ctx->dev_alloc((void **)&res->cols[h][c], (M ? M : 1) * sizeof(int64_t)));
        ctx->keep(res->cols[h][c]);
        if (M)
          GG_LAUNCH(ctx, "gather_ids", k_gather_ids, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, cols_cur[c],
                    csr->vid, M, res->cols[h][c]);
      }
    }
    for (auto p : cols_prev) ctx->dev_free(p);
    for (auto p : epos_prev) ctx->dev_free(p);
    ctx->dev_free(foff);
    cols_prev = cols_cur;
    epos_prev = epos_cur;
    foff = noff;
    n_prev = M;
    M = Mn;
  }
  for (auto p : cols_prev) ctx->dev_free(p);
  for (auto p : epos_prev) ctx->dev_free(p);
  ctx->dev_free(foff);
  ctx->dev_free(d_in);
  ctx->dev_free(d_out);
  ctx->dev_free(e_in);
  ctx->dev_free(e_out);
  GG_HIP(hipStreamSynchronize(ctx->stream));
  return GG_OK;
}

// ---- probe of a batch of keys (the device side of a generic single-key inner join) ---------------------------------
// matches of probe key i = the CSR row of its dense index; deg[i] entries (0 for a key that is not a vertex)
__global__ __launch_bounds__(256) void k_join_deg(const uint32_t *__restrict__ dense, uint64_t n,
                                                  const uint32_t *__restrict__ off, uint64_t *__restrict__ deg) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t d = dense[i];
    deg[i] = d == INVALID_U32 ? 0ull : (uint64_t)(off[d + 1] - off[d]);
  }
}

// output row r: probe position i = upper_bound(poff, r) - 1, the (r - poff[i])-th entry of that key's row, and the
// rowid the build side's Sink passed with it
__global__ __launch_bounds__(256) void k_join_emit(const uint64_t *__restrict__ poff, uint64_t n, uint64_t M,
                                                   const uint32_t *__restrict__ dense, const uint32_t *__restrict__ off,
                                                   const int64_t *__restrict__ eid, const uint32_t *__restrict__ epos,
                                                   int64_t *__restrict__ out_pos, int64_t *__restrict__ out_rowid) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  uint64_t lo = 0, hi = n;  // first i in [0, n] with poff[i] > r
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (poff[mid] <= r)
      lo = mid + 1;
    else
      hi = mid;
  }
  const uint64_t i = lo - 1;
  const uint64_t e = (uint64_t)off[dense[i]] + (r - poff[i]);
  out_pos[r] = (int64_t)i;
  out_rowid[r] = eid ? eid[e] : (int64_t)epos[e];
}

// ---- walk COUNTS from degrees ---------------------------------------------------------------------------------------
// count(*) over a join chain needs no row and no digest (the reference's aggregate above the joins only counts the
// chunks' cardinalities): with w_0(v) = how often v is a source and w_h(v) = sum over the in-neighbours u of v of
// w_{h-1}(u) = the h-hop walks that end in v, the h-hop walks number  rows_h = sum over v of w_{h-1}(v) * outdeg(v).
// From every vertex w_0 = 1 and w_1 = indeg, so the 1- and 2-hop counts are two passes over offset arrays and every
// further hop is one pull over the reverse rows.  Counts wrap mod 2^64 like the digest-bearing kernels' counters.
__global__ __launch_bounds__(256) void k_wc_seed(const uint32_t *__restrict__ dense, uint64_t n,
                                                 unsigned long long *__restrict__ w0) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && dense[i] != INVALID_U32) atomicAdd(&w0[dense[i]], 1ULL);
}

// out[0] += sum over v of weight(v) * outdeg(v); weight = w[v], or indeg(v) (w == nullptr, roff given), or 1
__global__ __launch_bounds__(256) void k_wc_dot(const uint32_t *__restrict__ off, const uint32_t *__restrict__ roff,
                                                const unsigned long long *__restrict__ w, uint64_t V,
                                                unsigned long long *__restrict__ out) {
  __shared__ uint64_t s_red[4];
  uint64_t acc = 0;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t wt = w ? w[v] : (roff ? (uint64_t)(roff[v + 1] - roff[v]) : 1ull);
    acc += wt * (uint64_t)(off[v + 1] - off[v]);
  }
  acc = wave_reduce_add_u64(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(s_red[0] + s_red[1] + s_red[2] + s_red[3]));
}

// w_out[v] = sum over the in-neighbours u of v of weight(u) (w_in[u], or indeg(u) when w_in == nullptr); 16 lanes per vertex
__global__ __launch_bounds__(256) void k_wc_pull(const uint32_t *__restrict__ roff, const uint32_t *__restrict__ rnbr,
                                                 const unsigned long long *__restrict__ w_in, uint64_t V,
                                                 unsigned long long *__restrict__ w_out) {
  const uint64_t v = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  uint64_t acc = 0;
  if (v < V) {
    for (uint32_t i = roff[v] + sub; i < roff[v + 1]; i += 16) {
      const uint32_t u = rnbr[i];
      acc += w_in ? w_in[u] : (uint64_t)(roff[u + 1] - roff[u]);
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (v < V && sub == 0) w_out[v] = acc;
}

// rows[h] for h in [k_min, k_max]; dense_src == nullptr: from every vertex (shards: the walks whose middle vertex —
// 1-hop rows: destination — is owned, k_max <= 2), else from the n_src dense indices (INVALID_U32 entries skipped)
int khop_count_rows(gg_ctx *ctx, gg_csr *csr, const uint32_t *dense_src, uint64_t n_src, int k_min, int k_max,
                    uint64_t *rows) {
  for (int h = 0; h <= GG_MAX_HOPS; h++) rows[h] = 0;
  const uint64_t V = csr->V;
  if (V == 0) return GG_OK;
  const bool all = dense_src == nullptr;
  if (k_max >= 2 || (all && csr->n_parts > 1)) GG_TRY(ensure_reverse(ctx, csr));
  unsigned long long *out = nullptr, *wa = nullptr, *wb = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&out, (GG_MAX_HOPS + 1) * sizeof(unsigned long long)));
  GG_HIP(hipMemsetAsync(out, 0, (GG_MAX_HOPS + 1) * sizeof(unsigned long long), ctx->stream));
  const unsigned dot_grid = (unsigned)std::min<uint64_t>((V + 255) / 256, 1024);
  const unsigned pull_grid = (unsigned)((V * 16 + 255) / 256);
  const unsigned long long *w = nullptr;  // weights of the level the next dot takes (nullptr: implicit)
  int level = 0;                         // w describes w_level
  if (!all) {
    GG_TRY(ctx->dev_alloc((void **)&wa, V * sizeof(unsigned long long)));
    GG_HIP(hipMemsetAsync(wa, 0, V * sizeof(unsigned long long), ctx->stream));
    if (n_src)
      GG_LAUNCH(ctx, "wc_seed", k_wc_seed, dim3((unsigned)((n_src + 255) / 256)), dim3(256), 0, dense_src, n_src, wa);
    w = wa;
  }
  for (int h = 1; h <= k_max; h++) {
    // rows_h = sum of w_{h-1} * outdeg
    if (all && h == 1) {
      // (w_0 = 1: every forward entry, or for a shard every entry into an owned vertex — a constant)
    } else if (all && h == 2) {
      GG_LAUNCH(ctx, "wc_dot", k_wc_dot, dim3(dot_grid), dim3(256), 0, csr->off, csr->roff,
                (const unsigned long long *)nullptr, V, out + h);
    } else {
      // make w_{h-1} explicit if it is not yet
      while (level < h - 1) {
        unsigned long long *&dst = (w == wa) ? wb : wa;
        if (!dst) GG_TRY(ctx->dev_alloc((void **)&dst, V * sizeof(unsigned long long)));
        // from every vertex the first explicit level is w_2 = pull(indeg)
        const bool from_indeg = all && level < 2;
        GG_LAUNCH(ctx, "wc_pull", k_wc_pull, dim3(pull_grid), dim3(256), 0, csr->roff, csr->rnbr,
                  from_indeg ? (const unsigned long long *)nullptr : w, V, dst);
        level = from_indeg ? 2 : level + 1;
        w = dst;
      }
      if (h >= k_min)
        GG_LAUNCH(ctx, "wc_dot", k_wc_dot, dim3(dot_grid), dim3(256), 0, csr->off, (const uint32_t *)nullptr, w, V, out + h);
    }
  }
  GG_HIP(hipMemcpyAsync(ctx->pin_scratch, out, (GG_MAX_HOPS + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                        ctx->stream));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  for (int h = k_min; h <= k_max; h++) rows[h] = ctx->pin_scratch[h];
  if (all && k_min <= 1) rows[1] = csr->n_parts > 1 ? csr->E_rev : csr->E;
  ctx->dev_free(out);
  ctx->dev_free(wa);
  ctx->dev_free(wb);
  return GG_OK;
}

int check_args(gg_ctx *ctx, const gg_csr *csr, int k_min, int k_max, gg_khop_stats *stats) {
  if (!ctx || !csr || !stats || csr->ctx != ctx) {
    set_error("gg_expand_khop: bad context/csr/stats argument");
    return GG_ERR_INVALID_ARG;
  }
  if (csr->n_parts > 1 && k_max != 2) {
    set_error("a CSR shard (gg_csr_build_shard) only supports all-source 2-hop count expansion");
    return GG_ERR_STATE;
  }
  if (k_min < 1 || k_max < k_min || k_max > GG_MAX_HOPS) {
    set_error("gg_expand_khop: need 1 <= k_min <= k_max <= %d (got %d..%d)", GG_MAX_HOPS, k_min, k_max);
    return GG_ERR_INVALID_ARG;
  }
  return GG_OK;
}

}  // namespace

extern "C" int gg_expand_khop_range(gg_ctx *ctx, const gg_csr *csr, uint64_t src_lo, uint64_t src_hi, int k_min,
                                    int k_max, int materialise, gg_khop_stats *stats, gg_result **out_result) {
  ApiScope scope(ctx);
  GG_TRY(check_args(ctx, csr, k_min, k_max, stats));
  if (out_result) *out_result = nullptr;
  if (materialise && !out_result) return GG_ERR_INVALID_ARG;
  if (src_hi > csr->V) src_hi = csr->V;
  if (src_lo > src_hi) src_lo = src_hi;
  GG_HIP(hipSetDevice(ctx->device));
  const uint64_t n0 = src_hi - src_lo;
  uint64_t M1 = csr->E;
  if (!(src_lo == 0 && src_hi == csr->V)) {
    uint32_t ends[2] = {0, 0};
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, csr->off + src_lo, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch + 1, csr->off + src_hi, sizeof(uint32_t), hipMemcpyDeviceToHost,
                          ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(&ends[0], ctx->pin_scratch, sizeof(uint32_t));
    memcpy(&ends[1], ctx->pin_scratch + 1, sizeof(uint32_t));
    M1 = (uint64_t)ends[1] - ends[0];
  }
  if (csr->n_parts > 1 && (materialise || !(src_lo == 0 && src_hi == csr->V))) {
    set_error("a CSR shard (gg_csr_build_shard) only supports all-source 2-hop count expansion");
    return GG_ERR_STATE;
  }
  if (k_max == 2 && src_lo == 0 && src_hi == csr->V && (ctx->force_frontier == 0 || csr->n_parts > 1)) {
    // every vertex is a source: the 2-hop walks are the per-vertex products in(x) x out(x)
    GG_TRY(khop_count_mid(ctx, const_cast<gg_csr *>(csr), 0, csr->V, k_min, stats));
  } else if (k_max == 3 && src_lo == 0 && src_hi == csr->V && ctx->force_frontier == 0 && csr->n_parts <= 1) {
    // ... and the 3-hop walks the products {2-hop rows ending in b} x out(b)
    GG_TRY(khop_count_mid3(ctx, const_cast<gg_csr *>(csr), k_min, stats));
  } else {
    GG_TRY(khop_count(ctx, csr, true, (uint32_t)src_lo, n0, M1, DevFrontier(), k_min, k_max, stats));
  }
  if (materialise) {
    gg_result *res = new gg_result();
    res->ctx = ctx;
    res->k_min = k_min;
    res->k_max = k_max;
    uint32_t *fv = nullptr;
    uint64_t *fdeg = nullptr;
    int rc = GG_OK;
    if (k_max == 2 && src_lo == 0 && src_hi == csr->V && ctx->force_frontier == 0) {
      // every vertex is a source: the rows are the per-vertex products in(x) x out(x), grouped by x (k_mat_mid2)
      rc = khop_materialise_mid2(ctx, const_cast<gg_csr *>(csr), 0, csr->V, k_min, res);
    } else {
      rc = ctx->dev_alloc((void **)&fv, (n0 ? n0 : 1) * sizeof(uint32_t));
      if (rc == GG_OK) rc = ctx->dev_alloc((void **)&fdeg, (n0 ? n0 : 1) * sizeof(uint64_t));
      if (rc == GG_OK && n0) {
        hipLaunchKernelGGL(k_iota_deg, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)src_lo,
                           n0, csr->off, fv, fdeg);
      }
      if (rc == GG_OK) rc = khop_materialise(ctx, csr, fv, n0, k_min, k_max, res);
    }
    ctx->dev_free(fv);
    ctx->dev_free(fdeg);
    if (rc != GG_OK) {
      gg_result_destroy(res);
      return rc;
    }
    *out_result = res;
  }
  return GG_OK;
}

extern "C" int gg_expand_khop(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, uint64_t n_src, int k_min,
                              int k_max, int materialise, gg_khop_stats *stats, gg_result **out_result) {
  ApiScope scope(ctx);
  if (!src_ids)
    return gg_expand_khop_range(ctx, csr, 0, csr ? csr->V : 0, k_min, k_max, materialise, stats, out_result);
  GG_TRY(check_args(ctx, csr, k_min, k_max, stats));
  if (csr->n_parts > 1) {
    set_error("a CSR shard (gg_csr_build_shard) only supports all-source 2-hop count expansion");
    return GG_ERR_STATE;
  }
  if (out_result) *out_result = nullptr;
  if (materialise && !out_result) return GG_ERR_INVALID_ARG;
  GG_HIP(hipSetDevice(ctx->device));

  // ids -> dense -> compacted frontier 0
  int64_t *ids_dev = nullptr;
  uint32_t *dense = nullptr;
  DevFrontier f0;
  unsigned long long *cursor = nullptr;
  const uint64_t n = n_src;
  GG_TRY(ctx->dev_alloc((void **)&ids_dev, (n ? n : 1) * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&dense, (n ? n : 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&f0.fv, (n ? n : 1) * sizeof(uint32_t)));
  GG_TRY(ctx->dev_alloc((void **)&f0.fq, (n ? n : 1) * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&f0.foff, (n + 1) * sizeof(uint64_t)));
  GG_TRY(ctx->dev_alloc((void **)&cursor, sizeof(unsigned long long)));
  GG_HIP(hipMemsetAsync(cursor, 0, sizeof(unsigned long long), ctx->stream));
  uint64_t n_valid = 0, M1 = 0;
  if (n) {
    GG_HIP(hipMemcpyAsync(ids_dev, src_ids, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));  // src_ids is caller memory: consumed before return
    GG_TRY(lookup_ids(ctx, csr, ids_dev, n, dense));
    GG_LAUNCH(ctx, "compact_sources", k_compact_sources, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, dense, n,
              f0.fv, f0.fq, f0.foff, csr->off, cursor);
    GG_TRY(read_u64(ctx, (const uint64_t *)cursor, &n_valid));
    GG_TRY(offsets_from_deg(ctx, f0.foff, n_valid, &M1));
  } else {
    GG_HIP(hipMemsetAsync(f0.foff, 0, sizeof(uint64_t), ctx->stream));
  }
  f0.n = n_valid;
  f0.M = M1;
  int rc = khop_count(ctx, csr, false, 0, n_valid, M1, f0, k_min, k_max, stats);
  if (rc == GG_OK && materialise) {
    gg_result *res = new gg_result();
    res->ctx = ctx;
    res->k_min = k_min;
    res->k_max = k_max;
    rc = khop_materialise(ctx, csr, f0.fv, n_valid, k_min, k_max, res);
    if (rc != GG_OK)
      gg_result_destroy(res);
    else
      *out_result = res;
  }
  ctx->dev_free(ids_dev);
  ctx->dev_free(dense);
  ctx->dev_free(cursor);
  free_frontier(ctx, f0);
  return rc;
}

extern "C" int gg_join_probe(gg_ctx *ctx, const gg_csr *csr, const int64_t *keys, uint64_t n, uint64_t *n_matches,
                            gg_result **out_result) {
  ApiScope scope(ctx);
  if (!ctx || !csr || csr->ctx != ctx || !n_matches || !out_result || (n && !keys)) return GG_ERR_INVALID_ARG;
  *out_result = nullptr;
  *n_matches = 0;
  if (csr->n_parts > 1 || !csr->has_rowid) {
    set_error("gg_join_probe needs a whole CSR built with edge rowids (gg_ctx_set_edge_rowid(ctx, 1))");
    return GG_ERR_STATE;
  }
  GG_HIP(hipSetDevice(ctx->device));
  gg_result *res = new gg_result();
  res->ctx = ctx;
  res->k_min = res->k_max = 1;
  uint64_t M = 0;
  int64_t *ids_dev = nullptr;
  uint32_t *dense = nullptr;
  uint64_t *poff = nullptr;
  int rc = GG_OK;
  if (n && csr->V) {
    rc = ctx->dev_alloc((void **)&ids_dev, n * sizeof(int64_t));
    if (rc == GG_OK) rc = ctx->dev_alloc((void **)&dense, n * sizeof(uint32_t));
    if (rc == GG_OK) rc = ctx->dev_alloc((void **)&poff, (n + 1) * sizeof(uint64_t));
    if (rc == GG_OK) {
      hipError_t e = hipMemcpyAsync(ids_dev, keys, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // keys is caller memory: consumed before return
      if (e != hipSuccess) {
        set_error("gg_join_probe: %s", hipGetErrorString(e));
        rc = GG_ERR_HIP;
      }
    }
    if (rc == GG_OK) rc = lookup_ids(ctx, csr, ids_dev, n, dense);
    if (rc == GG_OK) {
      hipLaunchKernelGGL(k_join_deg, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t *)dense, n,
                         (const uint32_t *)csr->off, poff);
      rc = offsets_from_deg(ctx, poff, n, &M);
    }
  }
  for (int c = 0; c < 2 && rc == GG_OK; c++) {
    rc = ctx->dev_alloc((void **)&res->cols[1][c], (M ? M : 1) * sizeof(int64_t));
    if (rc == GG_OK) ctx->keep(res->cols[1][c]);
  }
  if (rc == GG_OK && M)
    hipLaunchKernelGGL(k_join_emit, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream, (const uint64_t *)poff, n, M,
                       (const uint32_t *)dense, (const uint32_t *)csr->off, (const int64_t *)csr->eid,
                       (const uint32_t *)csr->epos, res->cols[1][0], res->cols[1][1]);
  if (rc != GG_OK) {
    gg_result_destroy(res);
    return rc;
  }
  res->rows[1] = M;
  *n_matches = M;
  *out_result = res;
  return GG_OK;
}

extern "C" int gg_khop_count(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, uint64_t n_src, int k_min, int k_max,
                            uint64_t *rows) {
  ApiScope scope(ctx);
  gg_khop_stats unused;
  GG_TRY(check_args(ctx, csr, k_min, k_max, &unused));
  if (!rows) return GG_ERR_INVALID_ARG;
  if (csr->n_parts > 1 && src_ids) {
    set_error("a CSR shard (gg_csr_build_shard) only supports all-source 2-hop count expansion");
    return GG_ERR_STATE;
  }
  GG_HIP(hipSetDevice(ctx->device));
  if (!src_ids) return khop_count_rows(ctx, const_cast<gg_csr *>(csr), nullptr, 0, k_min, k_max, rows);
  int64_t *ids_dev = nullptr;
  uint32_t *dense = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&ids_dev, (n_src ? n_src : 1) * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&dense, (n_src ? n_src : 1) * sizeof(uint32_t)));
  int rc = GG_OK;
  if (n_src) {
    GG_HIP(hipMemcpyAsync(ids_dev, src_ids, n_src * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    GG_HIP(hipStreamSynchronize(ctx->stream));  // src_ids is caller memory: consumed before return
    rc = lookup_ids(ctx, csr, ids_dev, n_src, dense);
  }
  if (rc == GG_OK) rc = khop_count_rows(ctx, const_cast<gg_csr *>(csr), dense, n_src, k_min, k_max, rows);
  ctx->dev_free(ids_dev);
  ctx->dev_free(dense);
  return rc;
}

extern "C" int gg_expand_khop_result(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, uint64_t n_src, int k_min,
                                     int k_max, gg_khop_stats *stats, gg_result **out_result) {
  return gg_expand_khop(ctx, csr, src_ids, n_src, k_min, k_max, 1, stats, out_result);
}

extern "C" int gg_expand_khop_edges(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, uint64_t n_src, int k,
                                    gg_khop_stats *stats, gg_result **out_result) {
  ApiScope scope(ctx);
  GG_TRY(check_args(ctx, csr, k, k, stats));
  if (!out_result) return GG_ERR_INVALID_ARG;
  *out_result = nullptr;
  if (csr->n_parts > 1 || !csr->has_rowid) {
    set_error("gg_expand_khop_edges needs a whole CSR built with edge rowids (gg_ctx_set_edge_rowid(ctx, 1))");
    return GG_ERR_STATE;
  }
  GG_HIP(hipSetDevice(ctx->device));
  // level 0: the given sources (dense, compacted) or every vertex
  int64_t *ids_dev = nullptr;
  uint32_t *dense = nullptr, *fv = nullptr;
  uint64_t n0 = 0;
  DevFrontier f0;
  unsigned long long *cursor = nullptr;
  int rc = GG_OK;
  if (src_ids) {
    const uint64_t n = n_src;
    GG_TRY(ctx->dev_alloc((void **)&ids_dev, (n ? n : 1) * sizeof(int64_t)));
    GG_TRY(ctx->dev_alloc((void **)&dense, (n ? n : 1) * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&f0.fv, (n ? n : 1) * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&f0.fq, (n ? n : 1) * sizeof(uint64_t)));
    GG_TRY(ctx->dev_alloc((void **)&f0.foff, (n + 1) * sizeof(uint64_t)));
    GG_TRY(ctx->dev_alloc((void **)&cursor, sizeof(unsigned long long)));
    GG_HIP(hipMemsetAsync(cursor, 0, sizeof(unsigned long long), ctx->stream));
    if (n) {
      GG_HIP(hipMemcpyAsync(ids_dev, src_ids, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
      GG_HIP(hipStreamSynchronize(ctx->stream));  // src_ids is caller memory: consumed before return
      GG_TRY(lookup_ids(ctx, csr, ids_dev, n, dense));
      GG_LAUNCH(ctx, "compact_sources", k_compact_sources, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, dense, n,
                f0.fv, f0.fq, f0.foff, csr->off, cursor);
      GG_TRY(read_u64(ctx, (const uint64_t *)cursor, &n0));
    }
    fv = f0.fv;
  } else {
    n0 = csr->V;
    uint64_t *fdeg = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&fv, (n0 ? n0 : 1) * sizeof(uint32_t)));
    GG_TRY(ctx->dev_alloc((void **)&fdeg, (n0 ? n0 : 1) * sizeof(uint64_t)));
    if (n0)
      hipLaunchKernelGGL(k_iota_deg, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, ctx->stream, 0u, n0, csr->off, fv,
                         fdeg);
    ctx->dev_free(fdeg);
  }
  gg_result *res = new gg_result();
  res->ctx = ctx;
  res->k_min = res->k_max = k;
  rc = khop_materialise(ctx, csr, fv, n0, k, k, res, true);
  if (rc == GG_OK) {
    memset(stats, 0, sizeof(*stats));
    stats->rows[k] = res->rows[k];
  }
  if (!src_ids) ctx->dev_free(fv);
  ctx->dev_free(ids_dev);
  ctx->dev_free(dense);
  ctx->dev_free(cursor);
  if (src_ids) free_frontier(ctx, f0);
  if (rc != GG_OK) {
    gg_result_destroy(res);
    return rc;
  }
  *out_result = res;
  return GG_OK;
}

extern "C" int gg_result_fetch_edges(const gg_result *res, int hops, uint64_t offset, uint32_t max_rows,
                                     int64_t *const *ecols, uint32_t *n_out) {
  if (!res || !ecols || !n_out || hops < res->k_min || hops > res->k_max) return GG_ERR_INVALID_ARG;
  gg_ctx *ctx = res->ctx;
  GG_HIP(hipSetDevice(ctx->device));
  const uint64_t total = res->rows[hops];
  if (offset >= total) {
    *n_out = 0;
    return GG_OK;
  }
  uint64_t take = total - offset;
  if (take > max_rows) take = max_rows;
  void *dst[GG_MAX_HOPS + 1];
  const void *src[GG_MAX_HOPS + 1];
  for (int c = 0; c < hops; c++) {
    if (!ecols[c] || !res->ecols[hops][c]) {
      set_error("gg_result_fetch_edges: the result carries no edge columns (gg_expand_khop_edges makes them)");
      return GG_ERR_INVALID_ARG;
    }
    dst[c] = ecols[c];
    src[c] = res->ecols[hops][c] + offset;
  }
  GG_TRY(ctx->fetch_columns(dst, src, hops, take * sizeof(int64_t)));
  *n_out = (uint32_t)take;
  return GG_OK;
}

extern "C" int gg_khop_partition(gg_ctx *ctx, const gg_csr *csr, int n_parts, uint64_t *bounds) {
  ApiScope scope(ctx);
  if (!ctx || !csr || n_parts < 1 || !bounds || csr->n_parts > 1) return GG_ERR_INVALID_ARG;
  GG_HIP(hipSetDevice(ctx->device));
  const uint64_t V = csr->V;
  bounds[0] = 0;
  bounds[n_parts] = V;
  if (V == 0 || n_parts == 1) {
    for (int i = 1; i < n_parts; i++) bounds[i] = V;
    return GG_OK;
  }
  uint64_t *work = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&work, (V + 1) * sizeof(uint64_t)));
  GG_LAUNCH(ctx, "twohop_work", k_twohop_work, dim3((unsigned)((V * 64 + 255) / 256)), dim3(256), 0, csr->off,
            csr->nbr, V, work);
  uint64_t total = 0;
  GG_TRY(offsets_from_deg(ctx, work, V, &total));
  std::vector<uint64_t> h(V + 1);
  GG_HIP(hipMemcpy(h.data(), work, (V + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
  ctx->dev_free(work);
  for (int i = 1; i < n_parts; i++) {
    uint64_t target = (uint64_t)((__uint128_t)total * (unsigned)i / (unsigned)n_parts);
    uint64_t lo = 0, hi = V;  // first vertex whose exclusive prefix >= target
    while (lo < hi) {
      uint64_t mid = (lo + hi) >> 1;
      if (h[mid] < target)
        lo = mid + 1;
      else
        hi = mid;
    }
    bounds[i] = lo;
  }
  return GG_OK;
}


extern "C" int gg_expand_khop_mid(gg_ctx *ctx, gg_csr *csr, uint64_t mid_lo, uint64_t mid_hi, int k_min, int k_max,
                                  gg_khop_stats *stats) {
  ApiScope scope(ctx);
  GG_TRY(check_args(ctx, csr, k_min, k_max, stats));
  if (k_max != 2) {
    set_error("gg_expand_khop_mid: only k_max == 2 is supported (got %d)", k_max);
    return GG_ERR_INVALID_ARG;
  }
  if (mid_hi > csr->V) mid_hi = csr->V;
  if (mid_lo > mid_hi) mid_lo = mid_hi;
  GG_HIP(hipSetDevice(ctx->device));
  return khop_count_mid(ctx, csr, mid_lo, mid_hi, k_min, stats);
}

extern "C" int gg_expand_khop_mid_result(gg_ctx *ctx, gg_csr *csr, uint64_t mid_lo, uint64_t mid_hi, int k_min,
                                         gg_khop_stats *stats, gg_result **out_result) {
  ApiScope scope(ctx);
  gg_khop_stats unused;
  GG_TRY(check_args(ctx, csr, k_min, 2, stats ? stats : &unused));
  if (!out_result) return GG_ERR_INVALID_ARG;
  *out_result = nullptr;
  if (mid_hi > csr->V) mid_hi = csr->V;
  if (mid_lo > mid_hi) mid_lo = mid_hi;
  GG_HIP(hipSetDevice(ctx->device));
  // (stats == NULL: the rows only — no counting expansion in front of the materialisation; gg_result_rows has the counts)
  if (stats) GG_TRY(khop_count_mid(ctx, csr, mid_lo, mid_hi, k_min, stats));
  gg_result *res = new gg_result();
  res->ctx = ctx;
  res->k_min = k_min;
  res->k_max = 2;
  const int rc = khop_materialise_mid2(ctx, csr, mid_lo, mid_hi, k_min, res);
  if (rc != GG_OK) {
    gg_result_destroy(res);
    return rc;
  }
  *out_result = res;
  return GG_OK;
}

extern "C" int gg_expand_khop_dev(gg_ctx *ctx, gg_csr *csr, int k_min, void **stats_dev) {
  ApiScope scope(ctx);
  gg_khop_stats unused;
  GG_TRY(check_args(ctx, csr, k_min, 2, &unused));
  if (!stats_dev) return GG_ERR_INVALID_ARG;
  GG_HIP(hipSetDevice(ctx->device));
  if (!ctx->stats_dev) GG_HIP(hipMalloc((void **)&ctx->stats_dev, 8 * sizeof(unsigned long long)));
  GG_TRY(khop_count_mid(ctx, csr, 0, csr->V, k_min, nullptr, ctx->stats_dev));
  *stats_dev = ctx->stats_dev;
  return GG_OK;
}

extern "C" int gg_stream_wait(gg_ctx *ctx, void *other_stream, int direction) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  GG_HIP(hipSetDevice(ctx->device));
  if (!ctx->xstream_event) GG_HIP(hipEventCreateWithFlags(&ctx->xstream_event, hipEventDisableTiming));
  hipStream_t other = (hipStream_t)other_stream;
  if (direction == 0) {  // the other stream waits for everything queued on the context's stream
    GG_HIP(hipEventRecord(ctx->xstream_event, ctx->stream));
    GG_HIP(hipStreamWaitEvent(other, ctx->xstream_event, 0));
  } else {  // the context's stream waits for everything queued on the other stream
    GG_HIP(hipEventRecord(ctx->xstream_event, other));
    GG_HIP(hipStreamWaitEvent(ctx->stream, ctx->xstream_event, 0));
  }
  return GG_OK;
}

extern "C" int gg_khop_partition_mid(gg_ctx *ctx, gg_csr *csr, int n_parts, uint64_t *bounds) {
  ApiScope scope(ctx);
  if (!ctx || !csr || n_parts < 1 || !bounds) return GG_ERR_INVALID_ARG;
  GG_HIP(hipSetDevice(ctx->device));
  const uint64_t V = csr->V;
  bounds[0] = 0;
  bounds[n_parts] = V;
  if (V == 0 || n_parts == 1) {
    for (int i = 1; i < n_parts; i++) bounds[i] = V;
    return GG_OK;
  }
  GG_TRY(ensure_reverse(ctx, csr));
  uint64_t *work = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&work, (V + 1) * sizeof(uint64_t)));
  GG_LAUNCH(ctx, "mid_work", k_mid_work, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, csr->off, csr->roff, V,
            work);
  uint64_t total = 0;
  GG_TRY(offsets_from_deg(ctx, work, V, &total));
  std::vector<uint64_t> h(V + 1);
  GG_HIP(hipMemcpy(h.data(), work, (V + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
  ctx->dev_free(work);
  for (int i = 1; i < n_parts; i++) {
    uint64_t target = (uint64_t)((__uint128_t)total * (unsigned)i / (unsigned)n_parts);
    uint64_t lo = 0, hi = V;
    while (lo < hi) {
      uint64_t mid = (lo + hi) >> 1;
      if (h[mid] < target)
        lo = mid + 1;
      else
        hi = mid;
    }
    bounds[i] = lo;
  }
  return GG_OK;
}

extern "C" int gg_result_rows(const gg_result *res, int hops, uint64_t *n_rows) {
  if (!res || !n_rows || hops < res->k_min || hops > res->k_max) return GG_ERR_INVALID_ARG;
  *n_rows = res->rows[hops];
  return GG_OK;
}

extern "C" int gg_result_fetch(const gg_result *res, int hops, uint64_t offset, uint32_t max_rows,
                               int64_t *const *cols, uint32_t *n_out) {
  if (!res || !cols || !n_out || hops < res->k_min || hops > res->k_max) return GG_ERR_INVALID_ARG;
  gg_ctx *ctx = res->ctx;
  GG_HIP(hipSetDevice(ctx->device));
  uint64_t total = res->rows[hops];
  if (offset >= total) {
    *n_out = 0;
    return GG_OK;
  }
  uint64_t take = total - offset;
  if (take > max_rows) take = max_rows;
  void *dst[GG_MAX_HOPS + 1];
  const void *src[GG_MAX_HOPS + 1];
  for (int c = 0; c <= hops; c++) {
    if (!cols[c]) return GG_ERR_INVALID_ARG;
    dst[c] = cols[c];
    src[c] = res->cols[hops][c] + offset;
  }
  GG_TRY(ctx->fetch_columns(dst, src, hops + 1, take * sizeof(int64_t)));
  *n_out = (uint32_t)take;
  return GG_OK;
}

// ---- digest of materialised rows ------------------------------------------------------------------------------------
// Reads the id columns a materialising expansion left in HBM, maps every id back to its dense index through the
// CSR's id table and sums the low halves of the row hashes (DESIGN.md "Row digest"): the checksum a count-mode
// expansion of the same walks reports, taken from what was actually WRITTEN.
__global__ __launch_bounds__(256) void k_result_digest(const int64_t *__restrict__ c0, const int64_t *__restrict__ c1,
                                                       const int64_t *__restrict__ c2, const int64_t *__restrict__ c3,
                                                       const int64_t *__restrict__ c4, int hops, uint64_t n,
                                                       const HtSlot *__restrict__ ht, uint64_t cap, int64_t min_idx,
                                                       unsigned long long *__restrict__ out /* [2]: digest, bad ids */) {
  const int64_t *cols[5] = {c0, c1, c2, c3, c4};
  uint32_t sum = 0, bad = 0;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t P = 0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      if (j <= hops) {
        const uint32_t d = ht_lookup(ht, cap, min_idx, cols[j][r]);
        ok = ok && d != INVALID_U32;
        P = j == 0 ? (uint64_t)d : dig_leaf(dig_q(P, j - 1), d);
      }
    }
    if (ok)
      sum += (uint32_t)P;
    else
      bad++;
  }
  sum = wave_total_u32(sum);
  bad = wave_total_u32(bad);
  if ((threadIdx.x & 63) == 0) {
    if (sum) atomicAdd(reinterpret_cast<unsigned int *>(&out[0]), sum);  // (mod 2^32: the high half stays zero)
    if (bad) atomicAdd(&out[1], (unsigned long long)bad);
  }
}

extern "C" int gg_result_digest(gg_ctx *ctx, const gg_csr *csr, const gg_result *res, int hops, uint64_t *n_rows,
                                uint64_t *digest) {
  if (!ctx || !csr || !res || !digest || res->ctx != ctx || csr->ctx != ctx) return GG_ERR_INVALID_ARG;
  if (hops < res->k_min || hops > res->k_max || hops < 1 || hops > 4) {
    set_error("gg_result_digest: hops %d outside the result's range or above 4", hops);
    return GG_ERR_INVALID_ARG;
  }
  ApiScope scope(ctx);
  GG_HIP(hipSetDevice(ctx->device));
  GG_TRY(ensure_ht(ctx, const_cast<gg_csr *>(csr)));
  const uint64_t n = res->rows[hops];
  unsigned long long *out = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&out, 2 * sizeof(unsigned long long)));
  GG_HIP(hipMemsetAsync(out, 0, 2 * sizeof(unsigned long long), ctx->stream));
  if (n) {
    const uint64_t want = (n + 255) / 256, cap = (uint64_t)ctx->num_cus * 32;
    GG_LAUNCH(ctx, "result_digest", k_result_digest, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0,
              (const int64_t *)res->cols[hops][0], (const int64_t *)res->cols[hops][1],
              (const int64_t *)res->cols[hops][2], (const int64_t *)res->cols[hops][3],
              (const int64_t *)res->cols[hops][4], hops, n, (const HtSlot *)csr->ht, csr->ht_cap, csr->ht_min_idx, out);
  }
  GG_HIP(hipMemcpyAsync(ctx->pin_scratch, out, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  ctx->dev_free(out);
  if (ctx->pin_scratch[1]) {
    set_error("gg_result_digest: %llu rows hold an id that is not a vertex", (unsigned long long)ctx->pin_scratch[1]);
    return GG_ERR_STATE;
  }
  if (n_rows) *n_rows = n;
  *digest = ctx->pin_scratch[0];
  return GG_OK;
}

extern "C" void gg_result_destroy(gg_result *res) {
  if (!res) return;
  gg_ctx *ctx = res->ctx;
  if (ctx) {
    // no synchronisation: the blocks return to the context's pool and their next user is queued on the same stream
    // behind whatever still writes them (gg_csr_destroy: same argument); a part-by-part producer thus queues the next
    // part while the last kernel of this one runs.  Fetches (gg_result_fetch) are complete when they return.
    for (int h = 0; h <= GG_MAX_HOPS; h++)
      for (int c = 0; c <= GG_MAX_HOPS; c++) {
        ctx->dev_free(res->cols[h][c]);
        ctx->dev_free(res->ecols[h][c]);
      }
  }
  delete res;
}
