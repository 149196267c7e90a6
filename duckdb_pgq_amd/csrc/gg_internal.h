// gg_internal.h — shared internals of libgg.so (HIP, gfx950 only; not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "gg.h"

namespace gg {

void set_error(const char *fmt, ...);

#define GG_HIP(call)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      gg::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
      return (e_ == hipErrorOutOfMemory) ? GG_ERR_OOM : GG_ERR_HIP;                                \
    }                                                                                              \
  } while (0)

#define GG_TRY(call)            \
  do {                          \
    int rc_ = (call);           \
    if (rc_ != GG_OK) return rc_; \
  } while (0)

constexpr uint32_t INVALID_U32 = 0xFFFFFFFFu;
constexpr int64_t HT_EMPTY = INT64_MIN;  // empty-slot sentinel of the id hash table

// digest constants — must match oracle/gg_oracle.c (DESIGN.md "Row digest")
constexpr uint32_t DIG_K32 = 0x9E3779B1u;
constexpr uint64_t DIG_GOLD = 0x9E3779B97F4A7C15ULL;

struct DevBlock {
  void *ptr;
  size_t size;
  bool in_use;
  uint64_t serial;  // allocation number (ApiScope frees what an API call allocated and did not keep)
  bool keep;        // owned by a long-lived object (gg_csr, gg_result)
  bool reserved = false;  // member of a placed column set (gg_ctx::dev_alloc_columns): only handed out with its set
};

struct ProfRec {
  int name_idx;
  hipEvent_t start, stop;
};

struct alignas(16) HtSlot {
  int64_t key;
  uint32_t val;
  uint32_t pad;
};

struct Column {
  int64_t *dev = nullptr;
  size_t cap = 0;  // rows
};

struct BuildStatus {  // device-side status block of a CSR build, copied back once per build
  unsigned long long dup_vertex;   // !=0: duplicate vertex id seen
  long long min_idx;               // dense index of the vertex with id == HT_EMPTY, or -1
  unsigned long long kept;         // edges kept in the forward CSR (both endpoints are vertices, source owned)
  unsigned long long kept_rev;     // edges kept in the reverse CSR (shard builds; destination owned)
  unsigned long long owned;        // vertices owned by this shard
  unsigned long long scan_error;   // !=0: a chained scan gave up waiting for a predecessor tile (gg_runtime.hip)
  unsigned long long dict_mode;    // bucketed build: DictMode the edge densification used (DICT_WIDE16: csr->ht is built)
};

// id -> dense index dictionary used while densifying edge rows, chosen on the device from the vertex ids'
// min/max (no host synchronisation): a direct-address array, a hash table of packed 8-byte slots, or the
// general 16-byte-slot table (gg_csr.hip, gg_csr_fast.hip)
constexpr uint64_t DIRECT_MAX_RANGE = 1u << 20;
enum DictMode : unsigned long long { DICT_WIDE16 = 0, DICT_PACKED8 = 1, DICT_DIRECT = 2 };
struct DirectMap {
  long long min_id;              // INT64_MAX until k_id_minmax ran
  long long max_id;
  unsigned long long enabled;    // ids span < DIRECT_MAX_RANGE values (mode == DICT_DIRECT)
  unsigned long long mode;       // DictMode
  unsigned long long idx_bits;   // packed slots: bits of the dense index field
  unsigned long long span_bits;  // bits of max_id - min_id
  unsigned long long pairs;      // packed table: number of 16-byte slot pairs (any number, gg_dict.h)
  unsigned long long decided;    // the mode chosen from min/max (mode may fall back to DICT_WIDE16 afterwards)
};

}  // namespace gg

namespace gg {
// Guards the few words an edge-row reservation touches.  Sink threads take it once per DataChunk; with a
// sleeping mutex 64+ of them form a convoy (measured: 40 M rows staged in 33 ms by 8 threads, 600 ms by
// 256), so the hot path spins for its ~50 ns critical section and only block switches use mu/cv.
// Test-and-test-and-set with a growing pause: 255 waiters that all WRITE the lock word (a bare test_and_set loop)
// keep its cache line away from the one thread that wants to release it.
struct SpinLock {
  std::atomic<bool> held{false};
  void lock() {
    for (unsigned spins = 1;; spins = spins < 64 ? spins * 2 : 64) {
      if (!held.exchange(true, std::memory_order_acquire)) return;
      do {
        for (unsigned i = 0; i < spins; i++) __builtin_ia32_pause();
      } while (held.load(std::memory_order_relaxed));
    }
  }
  void unlock() { held.store(false, std::memory_order_release); }
};
}  // namespace gg

struct gg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cus = 256;

  // ---- staging ----
  // mu: vertex staging, device columns, block flushes and everything slow.  edge_spin: the reservation
  // state of the edge blocks (cur_e, fill, base, has_rowid, OPEN/CLOSED, n_edges, implicit ranges).
  // Lock order: mu before edge_spin.
  std::mutex mu;
  gg::SpinLock edge_spin;
  std::condition_variable cv;                    // an edge block became FREE / the open block changed
  gg::Column c_vid, c_src, c_dst, c_rowid;
  uint64_t n_vertices = 0, n_edges = 0;          // rows resident, in flight to the device or reserved in a block
  bool rowid_explicit = false;                   // some append passed explicit rowids
  // rows appended without rowids, as merged (first row, count) ranges: their rowid is their position and
  // is only written (on the device, at build time) if some other append did pass rowids
  std::vector<std::pair<uint64_t, uint64_t>> implicit_rowid_ranges;
  static constexpr size_t STAGE_ROWS = 1u << 20; // rows per pinned staging block
  int64_t *pin_v[2] = {nullptr, nullptr};        // vertex ids
  hipEvent_t pin_v_free[2] = {nullptr, nullptr};
  int cur_v = 0;
  size_t fill_v = 0;
  // Edge blocks: [src | dst | rowid], each STAGE_ROWS.  Appenders reserve a row range under mu and copy
  // into the block OUTSIDE the lock (concurrent Sink calls overlap their memcpys); whoever reserves the
  // last row closes the block, opens the other one for everybody else, waits for the block's other
  // writers and sends it to the device.
  struct EdgeBlock {
    int64_t *pin = nullptr;
    hipEvent_t free_ev = nullptr;  // recorded after the block's H2D copies
    size_t fill = 0;               // rows reserved
    uint64_t base = 0;             // position of the block's first row in the staged columns
    std::atomic<int> writers{0};   // reservations whose memcpy has not finished (not guarded by mu)
    bool has_rowid = false;        // some reservation in the block carries explicit rowids
    enum State { OPEN, CLOSED, FREE } state = FREE;
  };
  // A ring of blocks (two: a ring of four staged SF100's edge table no faster, 33 GB/s on the box where both were
  // measured — 8 MB copies back to back reach 54 GB/s, scripts/ubench_h2d.hip: it is the host side that fills the
  // blocks at that pace, not the copy engine that waits for it)
  static constexpr int EDGE_BLOCKS = 2;
  EdgeBlock eblk[EDGE_BLOCKS];
  int cur_e = 0;                                 // the OPEN block

  // ---- pinned host buffers handed out by gg_host_alloc (guarded by host_mu) ----
  struct HostBlock {
    void *ptr;
    size_t bytes;
    bool in_use;
  };
  std::mutex host_mu;
  std::vector<HostBlock> host_blocks;

  // ---- caching device allocator ----
  // pool_mu: compute calls on one context are externally serialised, but results of earlier calls are
  // destroyed (their blocks freed) from whatever thread finishes with them
  std::mutex pool_mu;
  std::vector<gg::DevBlock> blocks;
  size_t bytes_allocated = 0;

  // ---- profiling ----
  int force_frontier = 0;  // gg_debug_force_frontier: 1 frontier kernels only, 2 product form of the last hop at any size
  bool rank_mode_forced = false;  // rank_mode was set by gg_debug_rank_mode, not decided by the probe
  int rank_mode = 0;            // gg_debug_rank_mode: 0 probe the LDS atomic order once, 1 ds_add_rtn ranks, 2 match masks
  bool legacy_build = false;    // gg_debug_force_legacy_build: the multi-pass LSD build (also taken for > 2^22 vertices)
  bool keep_edge_rowid = true;  // gg_ctx_set_edge_rowid
  uint64_t max_grid_tiles = 0;  // gg_debug_max_grid_tiles: workgroups per expansion launch (0: the hardware bound)
  bool profiling = false;
  std::vector<std::string> prof_names;
  std::vector<uint64_t> prof_launches;
  std::vector<double> prof_ms;
  std::vector<gg::ProfRec> prof_pending;
  std::vector<hipEvent_t> prof_event_pool;   // events are reused: creating two per launch costs more than timing
  std::vector<std::string> prof_selected;    // gg_profile_select: time only these kernels (empty: all)

  // small pinned scratch for D2H of counters
  uint64_t *pin_scratch = nullptr;  // 64 x u64 (word 63: copy of dev_err, see scan_error_fetch)
  // the bucketed build's status words are final long before its last kernel: copied to pin_scratch behind the
  // column scan, status_ev recorded behind the copy — gg_csr_build waits for that, not for the stream
  hipEvent_t status_ev = nullptr;
  bool status_early = false;
  unsigned long long *dev_err = nullptr;  // device word: != 0 after a chained scan gave up waiting
  uint32_t scan_spin_limit = 1u << 24;    // polls per predecessor before a scan tile gives up
  uint64_t scan_mute_tile = ~0ull;        // gg_debug_scan_fault: this scan tile never publishes (tests)

  unsigned long long *stats_dev = nullptr;  // gg_expand_khop_dev: the six result words of the last such call (8 x u64)
  hipEvent_t xstream_event = nullptr;       // gg_stream_wait
  // Result fetches (gg_result_fetch[_edges]) go over FETCH_LANES streams of their own, round robin: the pipeline's
  // threads drain a result in copies of ~1 MB per column, and one stream moves copies of that size at 33-37 GB/s
  // (a gap between copies; profiles/r04_ubench_h2d_numa.txt: 1 MiB pieces 36.6 GB/s on one stream, 51-55 on two or
  // four), so the lanes overlap one copy's tail with the next one's head.  A lane is ordered behind the library's
  // stream by an event per call (the result may still be in flight there) and serialises its own callers.
  static constexpr int FETCH_LANES = 4;
  struct FetchLane {
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;
    std::mutex mu;
  };
  FetchLane fetch_lane[FETCH_LANES];
  std::atomic<uint32_t> fetch_next{0};
  uint32_t fetch_lanes_used = FETCH_LANES;  // GG_FETCH_LANES (1..FETCH_LANES)
  // copy n_cols column slices to the host over one lane and wait for them (dst[c] <- src[c], bytes each)
  int fetch_columns(void *const *dst, const void *const *src, int n_cols, size_t bytes);

  uint64_t next_serial = 1;
  int dev_alloc(void **out, size_t bytes);
  // Three result columns of col_bytes each that will be written in lockstep (k_mat_mid2).  Small ones share one pooled
  // block; large ones are three blocks chosen so that they lie in different memory ranks (gg_runtime.hip "Placement"),
  // kept together as a set and recycled as a set.  cols[0..2] are freed individually with dev_free.
  int dev_alloc_columns(void **cols, size_t col_bytes);
  struct PlacedSet {
    void *col[3];
    size_t bytes;  // per column
  };
  std::vector<PlacedSet> placed_sets;
  int place_probes = 12;                                       // GG_PLACE_PROBES: at most this many candidate blocks per set (<= 3: no probing)
  uint64_t placed_built = 0, placed_fast_pairs = 0;            // diagnostics: sets built, fast pairs in the last one
  void dev_free(void *p);
  void keep(void *p);  // the block outlives the API call that allocated it
  int prof_begin(const char *name);
  void prof_end(int rec);
  int prof_flush();
};

struct gg_csr {
  gg_ctx *ctx = nullptr;
  uint64_t V = 0, E = 0, dropped = 0;
  uint64_t E_cap = 0;          // allocation size of the per-edge arrays (= staged edge rows)
  uint64_t E_rev = 0;          // entries of the reverse CSR (== E unless this is a shard)
  uint64_t owned_vertices = 0; // vertices owned by this shard (== V unless this is a shard)
  int part = 0, n_parts = 1;   // shard identity (gg_csr_build_shard)
  bool has_rowid = true;       // epos/eid valid (gg_ctx_set_edge_rowid, never for shards)
  uint32_t *off = nullptr;     // V+1 row offsets
  uint32_t *nbr = nullptr;     // E   dense neighbour (destination) per entry
  uint32_t *row = nullptr;     // E   dense source per entry (COO view, sorted by source)
  uint32_t *epos = nullptr;    // E   append position of the edge row (implicit rowid)
  int64_t *eid = nullptr;      // E   explicit edge rowids (only if the Sink passed rowids), else null
  int64_t *vid = nullptr;      // V   vertex ids by dense index
  gg::HtSlot *ht = nullptr;    // id hash table (open addressing, 16-byte slots: one line per probe)
  bool ht_built = false;       // filled at build time only if the densification needed it; else on first use (ensure_ht)
  uint64_t ht_cap = 0;         // slots; slot = mulhi(key * GOLD, ht_cap)
  int64_t ht_min_idx = -1;     // dense index of the vertex whose id == HT_EMPTY, if any
  // reverse CSR (in-neighbours), built lazily by ensure_reverse(): row x lists the sources u of
  // every edge u->x in ascending (u, rowid) order
  uint32_t *roff = nullptr;    // V+1
  uint32_t *rnbr = nullptr;    // E   source u of the reverse entry
  uint32_t *rrow = nullptr;    // E   destination x of the reverse entry (COO view, sorted by x)
  // the reverse entries grouped by SOURCE (gg_bfs.hip: ensure_push_in, shards only, on first use): row u lists
  // the owned destinations of u's edges, so a shard can push a light frontier into the words it owns
  uint32_t *pin_off = nullptr; // V+1
  uint32_t *pin_nbr = nullptr; // E_rev
};

struct gg_result {
  gg_ctx *ctx = nullptr;
  int k_min = 0, k_max = 0;
  uint64_t rows[GG_MAX_HOPS + 1] = {0};
  int64_t *cols[GG_MAX_HOPS + 1][GG_MAX_HOPS + 1] = {{nullptr}};  // cols[h][c], device
  int64_t *ecols[GG_MAX_HOPS + 1][GG_MAX_HOPS + 1] = {{nullptr}};  // ecols[h][j]: rowid of the walk's (j + 1)-th edge (gg_expand_khop_edges)
};

namespace gg {

// Put one at the top of every C-ABI entry point that allocates: whatever the call allocated from the pool
// and neither freed nor handed to a long-lived object (gg_ctx::keep) goes back to the pool when the call
// returns — including every early error return.
struct ApiScope {
  gg_ctx *ctx;
  uint64_t mark;
  explicit ApiScope(gg_ctx *c) : ctx(c), mark(c ? c->next_serial : 0) {}
  ~ApiScope() {
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    for (auto &b : ctx->blocks)
      if (b.in_use && !b.keep && b.serial >= mark) b.in_use = false;
  }
};

// RAII-free helper: launch wrapper that records events when profiling is on.
// (A launch of 2^32 threads or more is not refused by the runtime: the global size WRAPS and the kernel silently covers
// a fraction of its input — seen with one thread per endpoint of 2.4 G edge rows.  Kernels over that many items stride;
// the macro refuses whatever would still get there.)
#define GG_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                         \
  do {                                                                                \
    {                                                                                 \
      const dim3 g_ = (grid), b_ = (block);                                           \
      if ((unsigned long long)g_.x * g_.y * g_.z * b_.x * b_.y * b_.z >= (1ULL << 32)) { \
        set_error(name ": a launch of 2^32 threads or more");                         \
        return GG_ERR_TOO_LARGE;                                                      \
      }                                                                               \
    }                                                                                 \
    int prof_rec_ = (ctx)->profiling ? (ctx)->prof_begin(name) : -1;                  \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);       \
    if (prof_rec_ >= 0) (ctx)->prof_end(prof_rec_);                                   \
    GG_HIP(hipGetLastError());                                                        \
  } while (0)

// exclusive scan of n uint32 values (in place allowed: out may equal in); writes the grand
// total (as uint64) to *total_dev if non-null.  One hand-written kernel, tiles chained by decoupled
// look-back (gg_runtime.hip).
int scan_exclusive_u32(gg_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, uint64_t *total_dev);
// the error word of the chained scans: enqueue the fetch before a synchronisation, test after it
int scan_error_fetch(gg_ctx *ctx);
int scan_error_test(gg_ctx *ctx);
// exclusive scan of n uint64 values
int scan_exclusive_u64(gg_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t n, uint64_t *total_dev);

// id -> dense lookup of n host ids; writes dense (uint32, INVALID_U32 if absent) to out_dev
int lookup_ids(gg_ctx *ctx, const gg_csr *csr, const int64_t *ids_dev, uint64_t n, uint32_t *out_dev);
// n (key, value) pairs of u32 sorted by key < 2^key_bits, stably, into (key_out, val_out) — device arrays (gg_csr.hip)
int sort_pairs_by_key(gg_ctx *ctx, const uint32_t *key, const uint32_t *val, uint64_t n, int key_bits, uint32_t *key_out,
                      uint32_t *val_out);
int sort_triples_by_key(gg_ctx *ctx, const uint32_t *key, const uint32_t *a, const uint32_t *b, uint64_t n, int key_bits,
                        uint32_t *key_out, uint32_t *a_out, uint32_t *b_out);
// make a staged column hold need_rows rows, keeping the first live_rows (gg_runtime.hip; caller holds mu)
int grow_column(gg_ctx *ctx, Column &c, size_t live_rows, size_t need_rows);
// GG_STAGING_TRACE=1: print and reset the edge staging's waiting-time counters (gg_runtime.hip)
void staging_trace_print();
// build csr->roff / csr->rnbr if absent (gg_csr.hip)
int ensure_reverse(gg_ctx *ctx, gg_csr *csr);
// fill csr->ht from csr->vid if the build did not need it (gg_csr.hip); every ht_lookup user calls this first
int ensure_ht(gg_ctx *ctx, gg_csr *csr);
// bucketed two-level build of forward + reverse CSR (gg_csr_fast.hip); *taken = 0 if the graph is outside
// its range (> 2^22 vertices) and nothing was launched
int csr_build_fast(gg_ctx *ctx, gg_csr *csr, BuildStatus *st, int *taken);

__device__ __forceinline__ uint64_t fmix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
__device__ __forceinline__ uint64_t dig_q(uint64_t p, int j) { return fmix64(p + DIG_GOLD * (uint64_t)(j + 1)); }
__device__ __forceinline__ uint64_t dig_leaf(uint64_t q, uint32_t d) {
  return q ^ ((uint64_t)d * (uint64_t)DIG_K32);  // one v_mad_u64_u32
}

// slot of a key in a table of `cap` slots (any capacity, not only powers of two): multiplicative hash,
// then multiply-high range reduction; linear probing wraps at cap
__device__ __forceinline__ uint64_t ht_slot(int64_t key, uint64_t cap) {
  // Fibonacci hashing.  Database keys are mostly (pieces of) arithmetic progressions, which one odd multiplication
  // spreads almost evenly — shorter probe runs than a random function gives at the same load factor (measured at
  // SF100: a murmur-style finaliser here made the inserts 2.4x and the edge densification 2.7x slower).
  return __umul64hi((uint64_t)key * DIG_GOLD, cap);
}
__device__ __forceinline__ uint64_t ht_next(uint64_t slot, uint64_t cap) { return slot + 1 == cap ? 0 : slot + 1; }

__device__ __forceinline__ uint32_t ht_lookup(const HtSlot *__restrict__ ht, uint64_t cap,
                                              int64_t min_idx, int64_t key) {
  if (key == HT_EMPTY) return min_idx >= 0 ? (uint32_t)min_idx : INVALID_U32;
  uint64_t slot = ht_slot(key, cap);
  while (true) {
    // one 16-byte load per probe: key and value share a cache line
    const uint4 raw = *reinterpret_cast<const uint4 *>(&ht[slot]);
    const int64_t k = (int64_t)(((uint64_t)raw.y << 32) | raw.x);
    if (k == key) return raw.z;
    if (k == HT_EMPTY) return INVALID_U32;
    slot = ht_next(slot, cap);
  }
}

// finish a lookup whose first probe (slot, raw) is already loaded; continues linear probing on a miss
__device__ __forceinline__ uint32_t ht_resolve(const HtSlot *__restrict__ ht, uint64_t cap, int64_t min_idx,
                                               int64_t key, uint64_t slot, uint4 raw) {
  if (key == HT_EMPTY) return min_idx >= 0 ? (uint32_t)min_idx : INVALID_U32;
  while (true) {
    const int64_t k = (int64_t)(((uint64_t)raw.y << 32) | raw.x);
    if (k == key) return raw.z;
    if (k == HT_EMPTY) return INVALID_U32;
    slot = ht_next(slot, cap);
    raw = *reinterpret_cast<const uint4 *>(&ht[slot]);
  }
}

// shard ownership of a vertex id: independent of table order, so every rank decides it alone
__device__ __forceinline__ bool owns(int64_t id, uint32_t part, uint32_t n_parts) {
  // owner = floor(h32 * n_parts / 2^32) with h32 the high half of the multiplicative hash (no division)
  return n_parts <= 1 || (uint32_t)((((((uint64_t)id * DIG_GOLD) >> 32)) * (uint64_t)n_parts) >> 32) == part;
}

// Digest sums are 32-bit: the LOW half of each row hash, summed mod 2^32 (carried in u64 fields whose
// high half stays zero).  One v_xad_u32 (xor + add) per walk in the product kernel.
__host__ __device__ __forceinline__ uint64_t dsum_add(uint64_t a, uint64_t b) {
  return (uint64_t)(uint32_t)((uint32_t)a + (uint32_t)b);
}
__host__ __device__ __forceinline__ uint64_t dsum_sub(uint64_t a, uint64_t b) {
  return (uint64_t)(uint32_t)((uint32_t)a - (uint32_t)b);
}

// Inclusive prefix sums (mod 2^32) across the wave's lanes with data-parallel-primitive moves instead of
// ds_bpermute: four shifts inside each row of 16 lanes, then the totals of rows 0 / 2 into rows 1 / 3 and the total
// of the lower half into the upper.  Call with the whole wave active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {  // the moved value; 0 where no lane feeds this one
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t row_scan_incl(uint32_t v) {  // within rows of 16 lanes
  v += dpp_or_zero<0x111, 0xF>(v);  // row_shr:1
  v += dpp_or_zero<0x112, 0xF>(v);  // row_shr:2
  v += dpp_or_zero<0x114, 0xF>(v);  // row_shr:4
  v += dpp_or_zero<0x118, 0xF>(v);  // row_shr:8
  return v;
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
  v = row_scan_incl(v);
  v += dpp_or_zero<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_or_zero<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// sums over the wave with the DPP scan: the total is the last lane's prefix.  Call with the whole wave active.
__device__ __forceinline__ uint32_t wave_total_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(v), 63);
}
__device__ __forceinline__ uint64_t wave_reduce_dsum(uint64_t v) { return (uint64_t)wave_total_u32((uint32_t)v); }
// a count below 2^50 per lane: two 32-bit totals (24 low bits, the rest)
__device__ __forceinline__ uint64_t wave_total_u50(uint64_t v) {
  const uint32_t lo = wave_total_u32((uint32_t)v & 0xFFFFFFu), hi = wave_total_u32((uint32_t)(v >> 24));
  return ((uint64_t)hi << 24) + lo;
}

__device__ __forceinline__ uint64_t wave_reduce_add_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace gg
