// gg_runtime.hip — context, staging (Sink side), caching allocator, event timing, scans.
// HIP for gfx950; host side of the C-ABI declared in include/gg.h.
#include <string>
#include <chrono>
#include <thread>

#include "gg_internal.h"

#include <atomic>
#include <thread>

namespace gg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace gg

using namespace gg;

extern "C" const char *gg_version(void) { return "gg 0.1 (gfx950)"; }
extern "C" const char *gg_last_error(void) { return g_err; }

extern "C" int gg_device_count(int *out_count) {
  if (!out_count) return GG_ERR_INVALID_ARG;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *out_count = n;
  return GG_OK;
}

// ------------------------------------------------------------------------------------------
// caching device allocator: the build/expand path is re-run per query (and per bench step), so
// blocks are recycled instead of hipMalloc/hipFree inside the hot path (guide: Guideline 9).
// ------------------------------------------------------------------------------------------
int gg_ctx::dev_alloc(void **out, size_t bytes) {
  std::lock_guard<std::mutex> lk(pool_mu);
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) & ~size_t(255);
  if (bytes >= (size_t(64) << 20)) {
    // large blocks in steps of an eighth of their power of two: results produced part by part (near-equal parts of
    // several GB each) then recycle one set of blocks instead of growing the pool by a slightly larger set per part
    size_t step = size_t(1) << 26;
    while ((step << 4) <= bytes) step <<= 1;
    bytes = (bytes + step - 1) / step * step;
  }
  int best = -1;
  for (size_t i = 0; i < blocks.size(); i++) {
    if (!blocks[i].in_use && !blocks[i].reserved && blocks[i].size >= bytes && blocks[i].size <= bytes * 2 + (1u << 20)) {
      if (best < 0 || blocks[i].size < blocks[best].size) best = (int)i;
    }
  }
  if (best >= 0) {
    blocks[best].in_use = true;
    blocks[best].serial = next_serial++;
    blocks[best].keep = false;
    *out = blocks[best].ptr;
    return GG_OK;
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    // drop every cached free block (placed column sets too: a set with a member gone is forgotten) and retry once
    for (auto &b : blocks)
      if (!b.in_use && b.ptr) {
        if (b.reserved)
          for (size_t si = 0; si < placed_sets.size();) {
            auto &set = placed_sets[si];
            if (set.col[0] == b.ptr || set.col[1] == b.ptr || set.col[2] == b.ptr) {
              for (auto &o : blocks)
                if (o.ptr && (o.ptr == set.col[0] || o.ptr == set.col[1] || o.ptr == set.col[2])) o.reserved = false;
              placed_sets.erase(placed_sets.begin() + si);
            } else {
              si++;
            }
          }
        (void)hipFree(b.ptr);
        bytes_allocated -= b.size;
        b.ptr = nullptr;
        b.size = 0;
      }
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
      return GG_ERR_OOM;
    }
  }
  bytes_allocated += bytes;
  blocks.push_back({p, bytes, true, next_serial++, false});
  *out = p;
  return GG_OK;
}

// ------------------------------------------------------------------------------------------
// Placement of large result columns.  Columns that are written in lockstep (k_mat_mid2: three int64 columns) reach
// 7.0-7.2 TB/s or 5.5-5.9 TB/s depending on WHERE in device memory they lie.  scripts/ubench_fill_map.hip
// (profiles/r04_ubench_fill_map.txt) maps it: device memory falls into THREE classes of ~96 GB each (the three ranks of
// the 12-high HBM3E stacks, by every appearance; a 208 GiB allocation is a patchwork of them in pieces of 8-64 GiB);
// two streams in the same class run at 5.8 TB/s, in different classes at 7.15; three streams 5.75 when all share a
// class, 7.0 with two classes, 7.2 with three — independent of the order of the writes and of offsets inside a class.
// Nothing in a virtual address says which class it is in, so the pool asks the hardware: for columns of >= 1 GiB it
// allocates up to `place_probes` candidate blocks one after the other (with spacers between them, so that they reach
// into different classes), sorts each into a class by sparse lockstep fills against one block per class found so far,
// stops when it has three classes, keeps one block of each as a SET and returns the rest to the driver.  A set is recycled as a set (its blocks are reserved: the
// general pool does not hand them out), so the probing is paid once per context and column size.
// ------------------------------------------------------------------------------------------
typedef long long place_ll2 __attribute__((ext_vector_type(2)));
// chunk c of every `sample` chunks (PROBE_CHUNK pairs = 256 KiB per stream) of two streams, in lockstep
constexpr uint64_t PROBE_CHUNK = 16384;
__global__ __launch_bounds__(256) void k_place_probe(place_ll2 *__restrict__ a, place_ll2 *__restrict__ b, uint64_t pairs,
                                                     uint32_t sample) {
  const uint64_t lo = (uint64_t)blockIdx.x * sample * PROBE_CHUNK;
  const uint64_t hi = lo + PROBE_CHUNK < pairs ? lo + PROBE_CHUNK : pairs;
  place_ll2 v;
  v.x = (long long)blockIdx.x;
  v.y = (long long)threadIdx.x;
  for (uint64_t q = lo + threadIdx.x; q < hi; q += 256) {
    __builtin_nontemporal_store(v, a + q);
    __builtin_nontemporal_store(v, b + q);
  }
}

int gg_ctx::dev_alloc_columns(void **cols, size_t col_bytes) {
  constexpr size_t PROBE_MIN = size_t(1) << 30;
  const size_t stride = (col_bytes + 255) & ~size_t(255);
  if (stride < PROBE_MIN || place_probes <= 3) {  // one pooled block, the columns side by side
    char *base = nullptr;
    GG_TRY(dev_alloc((void **)&base, 3 * stride));
    for (int c = 0; c < 3; c++) cols[c] = base + c * stride;  // (cols[1], cols[2] are not pool blocks: dev_free ignores them)
    return GG_OK;
  }
  size_t bytes = stride;
  {  // the size classes of dev_alloc
    size_t step = size_t(1) << 26;
    while ((step << 4) <= bytes) step <<= 1;
    bytes = (bytes + step - 1) / step * step;
  }
  {
    std::lock_guard<std::mutex> lk(pool_mu);
    for (auto &set : placed_sets) {
      if (set.bytes < bytes || set.bytes > bytes * 2 + (1u << 20)) continue;
      DevBlock *blk[3] = {nullptr, nullptr, nullptr};
      for (auto &b : blocks)
        for (int c = 0; c < 3; c++)
          if (b.ptr == set.col[c]) blk[c] = &b;
      if (!blk[0] || !blk[1] || !blk[2] || blk[0]->in_use || blk[1]->in_use || blk[2]->in_use) continue;
      for (int c = 0; c < 3; c++) {
        blk[c]->in_use = true;
        blk[c]->serial = next_serial++;
        blk[c]->keep = false;
        cols[c] = blk[c]->ptr;
      }
      return GG_OK;
    }
  }
  // build a set
  const bool place_trace = getenv("GG_PLACE_TRACE") != nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    e0 = e1 = nullptr;
  }
  // Candidates one at a time, each CLASSIFIED as it arrives: "slow pair" is an equivalence (two blocks of the same
  // class: 5.2-5.7 TB/s; of different classes: 6.7-7.1 — 6.3 divides them), so a new block is probed against one
  // representative per class found so far and either joins the class of the first slow partner or founds a class.
  // Three classes found = three pairwise fast blocks: stop.  (Six candidates with every pair probed, the first form,
  // found only two classes on some boxes — 0.833 instead of 0.855 of HBM for k_mat_mid2 — because 6 x 16 GiB of
  // address space need not reach the third.)  Spacers stay allocated until the end so that the candidates advance
  // through the device; an eighth of the device is left alone.
  std::vector<void *> cand, spacers;
  std::vector<int> cls;      // class of each candidate
  std::vector<size_t> reps;  // first candidate of each class
  const size_t spacer = bytes < (size_t(16) << 30) ? (size_t(16) << 30) - bytes : 0;
  const uint64_t pairs = stride / 16;
  const uint64_t chunks = (pairs + PROBE_CHUNK - 1) / PROBE_CHUNK;
  const uint32_t sample = chunks >= 4096 ? 8 : 1;  // (columns of >= 1 GiB: an eighth of the chunks, over the whole extent)
  const unsigned grid = (unsigned)((chunks + sample - 1) / sample);
  bool probed = e0 != nullptr;
  std::string trace;
  auto probe = [&](void *x, void *y) -> double {
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {  // (the first launch after an allocation pays for its page tables)
      float ms = 0.f;
      (void)hipEventRecord(e0, stream);
      hipLaunchKernelGGL(k_place_probe, dim3(grid), dim3(256), 0, stream, (place_ll2 *)x, (place_ll2 *)y, pairs, sample);
      (void)hipEventRecord(e1, stream);
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) {
        (void)hipGetLastError();
        probed = false;
        return 0.0;
      }
      best = ms < best ? ms : best;
    }
    const double written = 2.0 * 16.0 * (double)std::min<uint64_t>(pairs, (uint64_t)grid * PROBE_CHUNK);
    return written / (best * 1e-3);
  };
  for (int t = 0; t < place_probes && e0 && reps.size() < 3; t++) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) break;
    if (t >= 3 && free_b < bytes + spacer + total_b / 8) break;  // (leave an eighth of the device alone)
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    cand.push_back(p);
    int c = -1;
    for (size_t r = 0; r < reps.size() && probed && c < 0; r++) {
      const double rate = probe(p, cand[reps[r]]);
      if (place_trace) trace += " " + std::to_string(t) + "-" + std::to_string(reps[r]) + " " + std::to_string(rate / 1e12).substr(0, 4);
      if (probed && rate < 6.3e12) c = cls[reps[r]];
    }
    if (c < 0 && probed) {
      c = (int)reps.size();
      reps.push_back((size_t)t);
    }
    cls.push_back(c < 0 ? 0 : c);
    if (reps.size() < 3 && spacer && t + 1 < place_probes && free_b > bytes + 2 * spacer + total_b / 8) {
      void *sp = nullptr;
      if (hipMalloc(&sp, spacer) == hipSuccess)
        spacers.push_back(sp);
      else
        (void)hipGetLastError();
    }
  }
  for (void *sp : spacers) (void)hipFree(sp);
  if (e0) {
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  if (cand.size() < 3) {  // not enough memory for separate blocks (or no events): whatever the pool has
    for (void *p : cand) (void)hipFree(p);
    char *base = nullptr;
    GG_TRY(dev_alloc((void **)&base, 3 * stride));
    for (int c = 0; c < 3; c++) cols[c] = base + c * stride;
    return GG_OK;
  }
  const size_t n = cand.size();
  // one block per class; with fewer than three classes the rest from the candidates in order (two classes: two of the
  // three pairs are fast, 7.0 instead of 7.2 TB/s)
  size_t pick[3];
  size_t n_pick = 0;
  for (size_t r : reps)
    if (n_pick < 3) pick[n_pick++] = r;
  for (size_t i = 0; i < n && n_pick < 3; i++) {
    bool taken = false;
    for (size_t k = 0; k < n_pick; k++) taken = taken || pick[k] == i;
    if (!taken) pick[n_pick++] = i;
  }
  int pick_fast = 0;
  for (int x = 0; x < 3; x++)
    for (int y = x + 1; y < 3; y++) pick_fast += probed && cls[pick[x]] != cls[pick[y]];
  if (place_trace) {
    fprintf(stderr, "[gg] column set of 3 x %.1f GiB: %zu candidates, classes", bytes / 1073741824.0, n);
    for (size_t i = 0; i < n; i++) fprintf(stderr, " %d", cls[i]);
    fprintf(stderr, "; probes (TB/s):%s -> blocks %zu %zu %zu (%d fast pairs)\n", trace.c_str(), pick[0], pick[1], pick[2],
            pick_fast);
  }
  for (size_t i = 0; i < n; i++)
    if (i != pick[0] && i != pick[1] && i != pick[2]) (void)hipFree(cand[i]);
  placed_built++;
  placed_fast_pairs = (uint64_t)pick_fast;
  std::lock_guard<std::mutex> lk(pool_mu);
  PlacedSet set;
  set.bytes = bytes;
  for (int c = 0; c < 3; c++) {
    set.col[c] = cols[c] = cand[pick[c]];
    bytes_allocated += bytes;
    DevBlock blk{cand[pick[c]], bytes, true, next_serial++, false};
    blk.reserved = true;
    blocks.push_back(blk);
  }
  placed_sets.push_back(set);
  return GG_OK;
}

void gg_ctx::keep(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(pool_mu);
  for (auto &b : blocks)
    if (b.ptr == p) {
      b.keep = true;
      return;
    }
}

void gg_ctx::dev_free(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(pool_mu);
  for (auto &b : blocks)
    if (b.ptr == p) {
      b.in_use = false;
      return;
    }
}

// ------------------------------------------------------------------------------------------
// event timing on the library's stream
// ------------------------------------------------------------------------------------------
static bool take_event(gg_ctx *ctx, hipEvent_t *ev) {
  if (!ctx->prof_event_pool.empty()) {
    *ev = ctx->prof_event_pool.back();
    ctx->prof_event_pool.pop_back();
    return true;
  }
  return hipEventCreate(ev) == hipSuccess;
}

int gg_ctx::prof_begin(const char *name) {
  if (!prof_selected.empty()) {
    bool wanted = false;
    for (auto &s : prof_selected) wanted = wanted || s == name;
    if (!wanted) return -1;
  }
  int idx = -1;
  for (size_t i = 0; i < prof_names.size(); i++)
    if (prof_names[i] == name) {
      idx = (int)i;
      break;
    }
  if (idx < 0) {
    idx = (int)prof_names.size();
    prof_names.push_back(name);
    prof_launches.push_back(0);
    prof_ms.push_back(0.0);
  }
  ProfRec r;
  r.name_idx = idx;
  if (!take_event(this, &r.start)) return -1;
  if (!take_event(this, &r.stop)) {
    prof_event_pool.push_back(r.start);
    return -1;
  }
  (void)hipEventRecord(r.start, stream);
  prof_pending.push_back(r);
  return (int)prof_pending.size() - 1;
}
void gg_ctx::prof_end(int rec) { (void)hipEventRecord(prof_pending[rec].stop, stream); }
int gg_ctx::prof_flush() {
  if (prof_pending.empty()) return GG_OK;
  GG_HIP(hipStreamSynchronize(stream));
  for (auto &r : prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      prof_ms[r.name_idx] += ms;
      prof_launches[r.name_idx] += 1;
    }
    prof_event_pool.push_back(r.start);
    prof_event_pool.push_back(r.stop);
  }
  prof_pending.clear();
  return GG_OK;
}

extern "C" int gg_profile_select(gg_ctx *ctx, const char *names) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  GG_TRY(ctx->prof_flush());
  ctx->prof_selected.clear();
  if (!names) return GG_OK;
  std::string all(names), cur;
  for (char c : all + ",") {
    if (c == ',') {
      if (!cur.empty()) ctx->prof_selected.push_back(cur);
      cur.clear();
    } else if (c != ' ') {
      cur.push_back(c);
    }
  }
  return GG_OK;
}

extern "C" int gg_ctx_set_edge_rowid(gg_ctx *ctx, int keep) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  ctx->keep_edge_rowid = keep != 0;
  return GG_OK;
}

extern "C" int gg_debug_force_frontier(gg_ctx *ctx, int on) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  ctx->force_frontier = on < 0 || on > 3 ? 1 : on;
  return GG_OK;
}

extern "C" int gg_debug_rank_mode(gg_ctx *ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return GG_ERR_INVALID_ARG;
  ctx->rank_mode = mode;
  ctx->rank_mode_forced = mode != 0;
  return GG_OK;
}

extern "C" int gg_debug_force_legacy_build(gg_ctx *ctx, int on) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  ctx->legacy_build = on != 0;
  return GG_OK;
}

extern "C" int gg_debug_scan_fault(gg_ctx *ctx, uint32_t spin_limit, uint64_t mute_tile) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  ctx->scan_spin_limit = spin_limit ? spin_limit : (1u << 24);
  ctx->scan_mute_tile = mute_tile;
  return GG_OK;
}

extern "C" int gg_debug_max_grid_tiles(gg_ctx *ctx, uint64_t max_tiles) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  ctx->max_grid_tiles = max_tiles;
  return GG_OK;
}

extern "C" int gg_debug_placement(gg_ctx *ctx, uint64_t *sets_built, uint64_t *fast_pairs_of_last_set) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  if (sets_built) *sets_built = ctx->placed_built;
  if (fast_pairs_of_last_set) *fast_pairs_of_last_set = ctx->placed_fast_pairs;
  return GG_OK;
}

extern "C" int gg_debug_reset(gg_ctx *ctx) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  ctx->force_frontier = 0;
  ctx->legacy_build = false;
  if (ctx->rank_mode_forced) ctx->rank_mode = 0;  // (a mode the probe decided is kept: it does not change)
  ctx->rank_mode_forced = false;
  ctx->scan_spin_limit = 1u << 24;
  ctx->scan_mute_tile = ~0ull;
  ctx->max_grid_tiles = 0;
  ctx->keep_edge_rowid = true;
  if (ctx->dev_err) {  // a fault-injection test may have left the chained scans' error word set
    GG_HIP(hipSetDevice(ctx->device));
    GG_HIP(hipMemsetAsync(ctx->dev_err, 0, sizeof(unsigned long long), ctx->stream));
  }
  return GG_OK;
}

extern "C" int gg_profile_enable(gg_ctx *ctx, int on) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  GG_TRY(ctx->prof_flush());
  ctx->profiling = on != 0;
  return GG_OK;
}
extern "C" int gg_profile_reset(gg_ctx *ctx) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  GG_TRY(ctx->prof_flush());
  ctx->prof_names.clear();
  ctx->prof_launches.clear();
  ctx->prof_ms.clear();
  return GG_OK;
}
extern "C" int gg_profile_count(gg_ctx *ctx, int *n) {
  if (!ctx || !n) return GG_ERR_INVALID_ARG;
  GG_TRY(ctx->prof_flush());
  *n = (int)ctx->prof_names.size();
  return GG_OK;
}
extern "C" int gg_profile_get(gg_ctx *ctx, int index, const char **name, uint64_t *launches, double *total_ms) {
  if (!ctx || index < 0 || index >= (int)ctx->prof_names.size()) return GG_ERR_INVALID_ARG;
  if (name) *name = ctx->prof_names[index].c_str();
  if (launches) *launches = ctx->prof_launches[index];
  if (total_ms) *total_ms = ctx->prof_ms[index];
  return GG_OK;
}

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
extern "C" int gg_ctx_create(int device, gg_ctx **out) {
  if (!out) return GG_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available: the gg hot path has no CPU fallback");
    return GG_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_error("device %d out of range (0..%d)", device, n - 1);
    return GG_ERR_INVALID_ARG;
  }
  GG_HIP(hipSetDevice(device));
  gg_ctx *ctx = new gg_ctx();
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
  GG_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  if (const char *e = getenv("GG_FETCH_LANES"))
    ctx->fetch_lanes_used = atoi(e) >= 1 && atoi(e) <= gg_ctx::FETCH_LANES ? (uint32_t)atoi(e) : gg_ctx::FETCH_LANES;
  if (const char *e = getenv("GG_PLACE_PROBES")) ctx->place_probes = atoi(e) > 0 ? (atoi(e) < 16 ? atoi(e) : 16) : 1;
  for (int i = 0; i < 2; i++) {
    GG_HIP(hipHostMalloc((void **)&ctx->pin_v[i], gg_ctx::STAGE_ROWS * sizeof(int64_t), hipHostMallocDefault));
    GG_HIP(hipEventCreateWithFlags(&ctx->pin_v_free[i], hipEventDisableTiming));
  }
  for (auto &b : ctx->eblk) {
    GG_HIP(hipHostMalloc((void **)&b.pin, 3 * gg_ctx::STAGE_ROWS * sizeof(int64_t), hipHostMallocDefault));
    GG_HIP(hipEventCreateWithFlags(&b.free_ev, hipEventDisableTiming));
  }
  ctx->eblk[0].state = gg_ctx::EdgeBlock::OPEN;
  for (auto &lane : ctx->fetch_lane) {
    GG_HIP(hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking));
    GG_HIP(hipEventCreateWithFlags(&lane.ready, hipEventDisableTiming));
  }
  GG_HIP(hipHostMalloc((void **)&ctx->pin_scratch, 64 * sizeof(uint64_t), hipHostMallocDefault));
  memset(ctx->pin_scratch, 0, 64 * sizeof(uint64_t));
  GG_HIP(hipMalloc((void **)&ctx->dev_err, sizeof(unsigned long long)));
  GG_HIP(hipEventCreateWithFlags(&ctx->status_ev, hipEventDisableTiming));
  GG_HIP(hipMemset(ctx->dev_err, 0, sizeof(unsigned long long)));
  *out = ctx;
  return GG_OK;
}

extern "C" void gg_ctx_destroy(gg_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)ctx->prof_flush();
  for (auto &b : ctx->blocks)
    if (b.ptr) (void)hipFree(b.ptr);
  for (int i = 0; i < 2; i++) {
    if (ctx->pin_v[i]) (void)hipHostFree(ctx->pin_v[i]);
    if (ctx->pin_v_free[i]) (void)hipEventDestroy(ctx->pin_v_free[i]);
  }
  for (auto &b : ctx->eblk) {
    if (b.pin) (void)hipHostFree(b.pin);
    if (b.free_ev) (void)hipEventDestroy(b.free_ev);
  }
  if (ctx->pin_scratch) (void)hipHostFree(ctx->pin_scratch);
  if (ctx->stats_dev) (void)hipFree(ctx->stats_dev);
  if (ctx->xstream_event) (void)hipEventDestroy(ctx->xstream_event);
  for (auto &lane : ctx->fetch_lane) {
    if (lane.stream) {
      (void)hipStreamSynchronize(lane.stream);
      (void)hipStreamDestroy(lane.stream);
    }
    if (lane.ready) (void)hipEventDestroy(lane.ready);
  }
  if (ctx->dev_err) (void)hipFree(ctx->dev_err);
  if (ctx->status_ev) (void)hipEventDestroy(ctx->status_ev);
  for (auto ev : ctx->prof_event_pool) (void)hipEventDestroy(ev);
  for (auto &h : ctx->host_blocks)
    if (h.ptr) (void)hipHostFree(h.ptr);
  // staged columns are plain hipMalloc (they grow by doubling, outside the block cache)
  if (ctx->c_vid.dev) (void)hipFree(ctx->c_vid.dev);
  if (ctx->c_src.dev) (void)hipFree(ctx->c_src.dev);
  if (ctx->c_dst.dev) (void)hipFree(ctx->c_dst.dev);
  if (ctx->c_rowid.dev) (void)hipFree(ctx->c_rowid.dev);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int gg_ctx::fetch_columns(void *const *dst, const void *const *src, int n_cols, size_t bytes) {
  // The lanes are for destinations in page-locked memory (gg_host_alloc of this or another context: the operators' slabs), where
  // the rate matters.  A copy into pageable memory is staged by the runtime and gains nothing from them (GG_KEY_JOIN's
  // <= 1024 pairs per chunk, the ctypes harness): those keep the library's stream.
  static const bool lanes_for_all = getenv("GG_FETCH_LANES_PAGEABLE") != nullptr;  // (diagnostic: the behaviour described above)
  bool pinned = true;
  if (!lanes_for_all) {
    std::lock_guard<std::mutex> lk(host_mu);
    for (int c = 0; c < n_cols && pinned; c++) {
      bool found = false;
      for (auto &h : host_blocks)
        found = found || ((const char *)dst[c] >= (const char *)h.ptr && (const char *)dst[c] + bytes <= (const char *)h.ptr + h.bytes);
      pinned = found;
    }
  }
  if (!pinned && !lanes_for_all) {
    // not this context's: the slab may be another context's (a sharded graph's streams fetch every part's rows into
    // slabs of part 0) — ask the runtime whether the memory is page-locked
    pinned = true;
    for (int c = 0; c < n_cols && pinned; c++) {
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, dst[c]) != hipSuccess) {
        (void)hipGetLastError();  // (an ordinary host pointer is an error to this call)
        pinned = false;
      } else {
        pinned = attr.type == hipMemoryTypeHost;
      }
    }
  }
  if (!pinned) {
    for (int c = 0; c < n_cols; c++) GG_HIP(hipMemcpyAsync(dst[c], src[c], bytes, hipMemcpyDeviceToHost, stream));
    GG_HIP(hipStreamSynchronize(stream));
    return GG_OK;
  }
  FetchLane &lane = fetch_lane[fetch_next.fetch_add(1, std::memory_order_relaxed) % fetch_lanes_used];
  std::lock_guard<std::mutex> lk(lane.mu);
  GG_HIP(hipEventRecord(lane.ready, stream));  // whatever produced the rows on the library's stream
  GG_HIP(hipStreamWaitEvent(lane.stream, lane.ready, 0));
  for (int c = 0; c < n_cols; c++) GG_HIP(hipMemcpyAsync(dst[c], src[c], bytes, hipMemcpyDeviceToHost, lane.stream));
  GG_HIP(hipStreamSynchronize(lane.stream));
  return GG_OK;
}

extern "C" int gg_host_alloc(gg_ctx *ctx, uint64_t bytes, void **out) {
  if (!ctx || !out) return GG_ERR_INVALID_ARG;
  *out = nullptr;
  if (bytes == 0) bytes = 1;
  std::lock_guard<std::mutex> lk(ctx->host_mu);
  size_t best = (size_t)-1;
  for (size_t i = 0; i < ctx->host_blocks.size(); i++) {  // smallest idle block that is large enough
    auto &h = ctx->host_blocks[i];
    if (!h.in_use && h.bytes >= bytes && (best == (size_t)-1 || h.bytes < ctx->host_blocks[best].bytes)) best = i;
  }
  if (best != (size_t)-1) {
    ctx->host_blocks[best].in_use = true;
    *out = ctx->host_blocks[best].ptr;
    return GG_OK;
  }
  GG_HIP(hipSetDevice(ctx->device));
  void *p = nullptr;
  GG_HIP(hipHostMalloc(&p, bytes, hipHostMallocPortable));
  ctx->host_blocks.push_back({p, (size_t)bytes, true});
  *out = p;
  return GG_OK;
}

extern "C" void gg_host_free(gg_ctx *ctx, void *ptr) {
  if (!ctx || !ptr) return;
  std::lock_guard<std::mutex> lk(ctx->host_mu);
  for (auto &h : ctx->host_blocks)
    if (h.ptr == ptr) h.in_use = false;
}

// ------------------------------------------------------------------------------------------
// staging: DataChunk columns -> pinned block -> HBM   (Sink side; thread-safe)
// ------------------------------------------------------------------------------------------
namespace gg {
int grow_column(gg_ctx *ctx, Column &c, size_t live_rows, size_t need_rows) {
  if (need_rows <= c.cap) return GG_OK;
  size_t ncap = c.cap ? c.cap : (size_t)1 << 16;
  while (ncap < need_rows) ncap *= 2;
  int64_t *p = nullptr;
  GG_HIP(hipMalloc((void **)&p, ncap * sizeof(int64_t)));
  if (c.dev && live_rows)
    GG_HIP(hipMemcpyAsync(p, c.dev, live_rows * sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
  if (c.dev) {
    GG_HIP(hipStreamSynchronize(ctx->stream));
    GG_HIP(hipFree(c.dev));
  }
  c.dev = p;
  c.cap = ncap;
  return GG_OK;
}
}  // namespace gg

// push the current (partially) filled pinned vertex block to the device; caller holds mu
static int flush_vertices(gg_ctx *ctx) {
  if (ctx->fill_v == 0) return GG_OK;
  size_t live = ctx->n_vertices - ctx->fill_v;
  GG_TRY(grow_column(ctx, ctx->c_vid, live, ctx->n_vertices));
  int b = ctx->cur_v;
  GG_HIP(hipMemcpyAsync(ctx->c_vid.dev + live, ctx->pin_v[b], ctx->fill_v * sizeof(int64_t), hipMemcpyHostToDevice,
                        ctx->stream));
  GG_HIP(hipEventRecord(ctx->pin_v_free[b], ctx->stream));
  ctx->cur_v ^= 1;
  ctx->fill_v = 0;
  GG_HIP(hipEventSynchronize(ctx->pin_v_free[ctx->cur_v]));  // the other block must have drained
  return GG_OK;
}

// send the reserved rows of an edge block to the device; caller holds mu and the block has no writer left.
// Blocks are flushed in the order they were opened, so b.base rows are already resident.
static int flush_edge_block(gg_ctx *ctx, gg_ctx::EdgeBlock &b) {
  if (b.fill == 0) return GG_OK;
  const size_t live = (size_t)b.base, need = live + b.fill;
  GG_TRY(grow_column(ctx, ctx->c_src, live, need));
  GG_TRY(grow_column(ctx, ctx->c_dst, live, need));
  const size_t S = gg_ctx::STAGE_ROWS;
  const size_t bytes = b.fill * sizeof(int64_t);
  GG_HIP(hipMemcpyAsync(ctx->c_src.dev + live, b.pin, bytes, hipMemcpyHostToDevice, ctx->stream));
  GG_HIP(hipMemcpyAsync(ctx->c_dst.dev + live, b.pin + S, bytes, hipMemcpyHostToDevice, ctx->stream));
  if (b.has_rowid) {  // rows without explicit rowids are filled in on the device if ever needed
    GG_TRY(grow_column(ctx, ctx->c_rowid, live, need));
    GG_HIP(hipMemcpyAsync(ctx->c_rowid.dev + live, b.pin + 2 * S, bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  GG_HIP(hipEventRecord(b.free_ev, ctx->stream));
  return GG_OK;
}

static void reset_edge_blocks(gg_ctx *ctx) {  // caller holds mu
  ctx->edge_spin.lock();
  for (auto &b : ctx->eblk) {
    b.fill = 0;
    b.base = 0;
    b.has_rowid = false;
    b.state = gg_ctx::EdgeBlock::FREE;
  }
  ctx->cur_e = 0;
  ctx->eblk[0].state = gg_ctx::EdgeBlock::OPEN;
  ctx->implicit_rowid_ranges.clear();
  ctx->rowid_explicit = false;
  ctx->edge_spin.unlock();
}

extern "C" int gg_vertices_append(gg_ctx *ctx, const int64_t *id, uint64_t n) {
  if (!ctx || (!id && n)) return GG_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  GG_HIP(hipSetDevice(ctx->device));
  if (ctx->n_vertices + n >= (uint64_t)INVALID_U32) {
    set_error("vertex table larger than 2^32-2 rows is not supported");
    return GG_ERR_TOO_LARGE;
  }
  while (n) {
    size_t room = gg_ctx::STAGE_ROWS - ctx->fill_v;
    size_t take = n < room ? (size_t)n : room;
    memcpy(ctx->pin_v[ctx->cur_v] + ctx->fill_v, id, take * sizeof(int64_t));
    ctx->fill_v += take;
    ctx->n_vertices += take;
    id += take;
    n -= take;
    if (ctx->fill_v == gg_ctx::STAGE_ROWS) GG_TRY(flush_vertices(ctx));
  }
  return GG_OK;
}

// GG_STAGING_TRACE=1: where the appenders of the edge staging spend their waiting time (printed by gg_staging_sync)
static std::atomic<uint64_t> g_tr_full_ns{0}, g_tr_closer_ns{0}, g_tr_drain_ns{0}, g_tr_copy_ns{0}, g_tr_calls{0};
static const bool g_tr_on = getenv("GG_STAGING_TRACE") != nullptr;
static inline uint64_t tr_now() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

namespace gg {
void staging_trace_print() {
  if (g_tr_on)
    fprintf(stderr, "staging trace: %llu appends; summed over threads: copies %.2f ms, waiting for a block with room %.2f ms, "
            "closers waiting for the other block %.2f ms, closers waiting for writers %.2f ms\n",
            (unsigned long long)g_tr_calls.exchange(0), g_tr_copy_ns.exchange(0) / 1e6, g_tr_full_ns.exchange(0) / 1e6,
            g_tr_closer_ns.exchange(0) / 1e6, g_tr_drain_ns.exchange(0) / 1e6);
}
}  // namespace gg

extern "C" int gg_edges_append(gg_ctx *ctx, const int64_t *src, const int64_t *dst, const int64_t *rowid,
                               uint64_t n) {
  if (!ctx || ((!src || !dst) && n)) return GG_ERR_INVALID_ARG;
  const size_t S = gg_ctx::STAGE_ROWS;
  while (n) {
    // ---- reserve rows in the open block (spin lock: a handful of words)
    ctx->edge_spin.lock();
    gg_ctx::EdgeBlock &b = ctx->eblk[ctx->cur_e];
    if (b.fill == S) {
      // full: its closer has not switched blocks yet (it may be waiting for the other block's copies)
      ctx->edge_spin.unlock();
      const uint64_t t0 = g_tr_on ? tr_now() : 0;
      std::unique_lock<std::mutex> lk(ctx->mu);
      ctx->cv.wait(lk, [&] {
        ctx->edge_spin.lock();
        const bool room = ctx->eblk[ctx->cur_e].fill < S;
        ctx->edge_spin.unlock();
        return room;
      });
      if (g_tr_on) g_tr_full_ns += tr_now() - t0;
      continue;
    }
    if (ctx->n_edges + n >= (uint64_t)INVALID_U32) {
      ctx->edge_spin.unlock();
      set_error("edge table larger than 2^32-2 rows is not supported");
      return GG_ERR_TOO_LARGE;
    }
    const size_t room = S - b.fill;
    const size_t take = n < room ? (size_t)n : room;
    const size_t off = b.fill;
    if (off == 0) b.base = ctx->n_edges;
    const uint64_t first_row = b.base + off;
    b.fill += take;
    ctx->n_edges += take;
    b.writers.fetch_add(1, std::memory_order_relaxed);
    if (rowid) {
      b.has_rowid = true;
      ctx->rowid_explicit = true;
    } else if (!ctx->implicit_rowid_ranges.empty() &&
               ctx->implicit_rowid_ranges.back().first + ctx->implicit_rowid_ranges.back().second == first_row) {
      ctx->implicit_rowid_ranges.back().second += take;
    } else {
      ctx->implicit_rowid_ranges.emplace_back(first_row, (uint64_t)take);
    }
    const bool closer = b.fill == S;
    ctx->edge_spin.unlock();

    // ---- copy outside any lock: concurrent Sink calls overlap here
    const uint64_t tc0 = g_tr_on ? tr_now() : 0;
    memcpy(b.pin + off, src, take * sizeof(int64_t));
    memcpy(b.pin + S + off, dst, take * sizeof(int64_t));
    if (rowid) {
      memcpy(b.pin + 2 * S + off, rowid, take * sizeof(int64_t));
      rowid += take;
    }
    b.writers.fetch_sub(1, std::memory_order_release);
    if (g_tr_on) {
      g_tr_copy_ns += tr_now() - tc0;
      g_tr_calls++;
    }
    src += take;
    dst += take;
    n -= take;
    if (closer) {
      // Whoever reserved the block's last row sends it off and opens the other one.  In THIS order: the flush is
      // queued as soon as the block's writers are done — behind the other block's copies, which are normally still
      // on their way, so the copy engine goes from one block to the next without a pause — and only then does the
      // closer wait for the copies of the ring's next block to have left its pinned memory, and reopen it.  (The other way round
      // — reopen first, flush after — left the engine idle from the end of one block's copies until the host had
      // woken up and issued the next: 30 instead of 40 GB/s over PCIe; everybody else waits for the switch either way.)
      const int next_e = (int)((&b - ctx->eblk) + 1) % gg_ctx::EDGE_BLOCKS;
      gg_ctx::EdgeBlock &o = ctx->eblk[next_e];  // the next block of the ring: the one flushed longest ago
      const uint64_t td0 = g_tr_on ? tr_now() : 0;
      while (b.writers.load(std::memory_order_acquire) != 0) std::this_thread::yield();
      if (g_tr_on) g_tr_drain_ns += tr_now() - td0;
      std::unique_lock<std::mutex> lk(ctx->mu);
      int rc = hipSetDevice(ctx->device) == hipSuccess ? flush_edge_block(ctx, b) : GG_ERR_HIP;
      b.state = gg_ctx::EdgeBlock::FREE;  // flushed (also on error: nobody may wait for this block forever)
      const uint64_t t0 = g_tr_on ? tr_now() : 0;
      ctx->cv.wait(lk, [&] { return o.state == gg_ctx::EdgeBlock::FREE; });  // (its closer flushed it before it opened b)
      if (rc == GG_OK && hipEventSynchronize(o.free_ev) != hipSuccess) {
        set_error("HIP error while waiting for a staging block");
        rc = GG_ERR_HIP;
      }
      if (g_tr_on) g_tr_closer_ns += tr_now() - t0;
      ctx->edge_spin.lock();
      o.fill = 0;
      o.has_rowid = false;
      o.state = gg_ctx::EdgeBlock::OPEN;
      ctx->cur_e = next_e;
      ctx->edge_spin.unlock();
      lk.unlock();
      ctx->cv.notify_all();  // appenders waiting for a block with room
      if (rc != GG_OK) return rc;
    }
  }
  return GG_OK;
}

extern "C" int gg_staging_sync(gg_ctx *ctx) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  gg::staging_trace_print();
  std::lock_guard<std::mutex> lk(ctx->mu);
  GG_HIP(hipSetDevice(ctx->device));
  GG_TRY(flush_vertices(ctx));
  // the open block, partially filled (appends running concurrently with a sync are outside the contract,
  // but never leave a half-copied reservation behind)
  ctx->edge_spin.lock();
  gg_ctx::EdgeBlock &b = ctx->eblk[ctx->cur_e];
  ctx->edge_spin.unlock();
  while (b.writers.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  GG_TRY(flush_edge_block(ctx, b));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  ctx->edge_spin.lock();
  b.fill = 0;  // the pinned memory is free again; the block stays open
  b.has_rowid = false;
  ctx->edge_spin.unlock();
  return GG_OK;
}

extern "C" int gg_staging_counts(gg_ctx *ctx, uint64_t *n_vertices, uint64_t *n_edges) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n_vertices) *n_vertices = ctx->n_vertices;
  if (n_edges) *n_edges = ctx->n_edges;
  return GG_OK;
}

extern "C" int gg_staging_clear(gg_ctx *ctx) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  GG_HIP(hipSetDevice(ctx->device));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  ctx->n_vertices = ctx->n_edges = 0;
  ctx->fill_v = 0;
  reset_edge_blocks(ctx);
  return GG_OK;
}

extern "C" int gg_staging_clear_edges(gg_ctx *ctx) {
  if (!ctx) return GG_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  GG_HIP(hipSetDevice(ctx->device));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  ctx->n_edges = 0;
  reset_edge_blocks(ctx);
  return GG_OK;
}

// ------------------------------------------------------------------------------------------
// exclusive scans (hand-written: one pass, tiles chained by decoupled look-back)
// ------------------------------------------------------------------------------------------
namespace gg {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;  // per thread; block tile = 4096
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix and
// writes the block total to *total (all threads get it).
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *total, T *lds /* >= 4 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T incl = wave_incl_scan(v, lane);
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  T wbase = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++) {
    T s = lds[w];
    if (w < wave) wbase += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return wbase + incl - v;
}

// ---- single-pass scan with decoupled look-back ---------------------------------------------------------
// One kernel instead of three (reduce, scan of block sums, apply) and one read of the input instead of two:
// the arrays scanned here are the (digit x tile) counters of a radix pass — a megabyte or less per scan,
// seven scans per build — so launch latency and the second read are most of their cost.
// Tiles take a ticket (so a tile only ever waits for tiles that already run), publish their sum
// (FLAG_SUM), then wave 0 looks back 64 predecessors at a time, adding sums until it meets a tile that
// already knows its inclusive prefix (FLAG_PREFIX), and publishes its own.  A status word carries flag and
// value together (one 8-byte agent-scope atomic), so no fence ordering between separate words is needed.
constexpr unsigned long long SCAN_FLAG_SUM = 1ULL << 62, SCAN_FLAG_PREFIX = 2ULL << 62;
constexpr unsigned long long SCAN_VALUE_MASK = (1ULL << 62) - 1;

template <typename TIn, typename TOut>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chained(const TIn *__restrict__ in, TOut *__restrict__ out,
                                                               uint64_t n, uint64_t nb,
                                                               unsigned long long *__restrict__ status /* nb + 1 */,
                                                               uint64_t *__restrict__ total_dev,
                                                               unsigned long long *__restrict__ err, uint32_t spin_limit,
                                                               uint64_t mute_tile /* fault injection: never publishes */) {
  __shared__ uint64_t lds[4];
  __shared__ uint64_t s_tile, s_prefix;
  if (threadIdx.x == 0) s_tile = atomicAdd(&status[nb], 1ULL);  // the ticket counter lives behind the status words
  __syncthreads();
  const uint64_t tile = s_tile;
  // thread t owns SCAN_ITEMS consecutive elements -> serial scan in registers + block scan of sums
  const uint64_t base = tile * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t v[SCAN_ITEMS];
  uint64_t sum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const uint64_t idx = base + i;
    v[i] = idx < n ? (uint64_t)in[idx] : 0;
    sum += v[i];
  }
  uint64_t tot;
  uint64_t ex = block_excl_scan<uint64_t>(sum, &tot, lds);
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {  // wave 0: publish, look back, publish
    const bool mute = tile == mute_tile;
    if (tile > 0 && lane == 0 && !mute)
      __hip_atomic_store(&status[tile], SCAN_FLAG_SUM | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t run = 0;
    int64_t hi = (int64_t)tile - 1;  // nearest predecessor not yet accounted for
    while (hi >= 0) {
      const int64_t j = hi - lane;
      unsigned long long w = SCAN_FLAG_PREFIX;  // before tile 0: prefix 0
      if (j >= 0) {
        uint32_t spins = 0;
        do {
          w = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((w >> 62) == 0 && ++spins < spin_limit);  // predecessors hold lower tickets: they are running
        // gave up (a predecessor never published): the prefix is wrong from here on — say so, the host turns
        // the flag into GG_ERR_HIP at its next synchronisation (scan_error_fetch / scan_error_test)
        if ((w >> 62) == 0) atomicOr(err, 1ULL);
      }
      const unsigned long long is_prefix = __ballot((w >> 62) == 2);
      // lanes up to (and including) the nearest tile that knows its prefix contribute; a lane that gave
      // up waiting contributes nothing (the grid still drains; the error word above is set)
      const int stop = is_prefix ? __ffsll((long long)is_prefix) - 1 : 63;
      uint64_t part = (lane <= stop && (w >> 62) != 0) ? (uint64_t)(w & SCAN_VALUE_MASK) : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      run += part;
      if (is_prefix) break;
      hi -= 64;
    }
    if (lane == 0) {
      if (!mute)
        __hip_atomic_store(&status[tile], SCAN_FLAG_PREFIX | ((run + tot) & SCAN_VALUE_MASK), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = run;
      if (tile == nb - 1 && total_dev) *total_dev = run + tot;
    }
  }
  __syncthreads();
  ex += s_prefix;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const uint64_t idx = base + i;
    if (idx < n) out[idx] = (TOut)ex;
    ex += v[i];
  }
}

template <typename TIn, typename TOut, typename TAcc>
static int scan_impl(gg_ctx *ctx, const TIn *in, TOut *out, uint64_t n, uint64_t *total_dev) {
  if (n == 0) {
    if (total_dev) GG_HIP(hipMemsetAsync(total_dev, 0, sizeof(uint64_t), ctx->stream));
    return GG_OK;
  }
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  unsigned long long *status = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&status, (nb + 1) * sizeof(unsigned long long)));
  GG_HIP(hipMemsetAsync(status, 0, (nb + 1) * sizeof(unsigned long long), ctx->stream));
  GG_LAUNCH(ctx, "scan_chained", (k_scan_chained<TIn, TOut>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, in, out, n, nb,
            status, total_dev, ctx->dev_err, ctx->scan_spin_limit, ctx->scan_mute_tile);
  ctx->dev_free(status);  // stream-ordered reuse: later work on the same stream runs after this kernel
  return GG_OK;
}

// A chained scan that gave up waiting sets ctx->dev_err.  Callers enqueue scan_error_fetch before a
// synchronisation they do anyway and call scan_error_test after it.
int scan_error_fetch(gg_ctx *ctx) {
  GG_HIP(hipMemcpyAsync(ctx->pin_scratch + 63, ctx->dev_err, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                        ctx->stream));
  return GG_OK;
}
int scan_error_test(gg_ctx *ctx) {
  if (ctx->pin_scratch[63] == 0) return GG_OK;
  ctx->pin_scratch[63] = 0;
  GG_HIP(hipMemsetAsync(ctx->dev_err, 0, sizeof(unsigned long long), ctx->stream));
  set_error("a chained prefix scan gave up waiting for a predecessor tile: the result of this call is not valid");
  return GG_ERR_HIP;
}

int scan_exclusive_u32(gg_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, uint64_t *total_dev) {
  return scan_impl<uint32_t, uint32_t, uint64_t>(ctx, in, out, n, total_dev);
}
int scan_exclusive_u64(gg_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t n, uint64_t *total_dev) {
  return scan_impl<uint64_t, uint64_t, uint64_t>(ctx, in, out, n, total_dev);
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lookup_ids(const int64_t *__restrict__ ids, uint64_t n,
                                                    const HtSlot *__restrict__ ht, uint64_t cap,
                                                    int64_t min_idx, uint32_t *__restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ht_lookup(ht, cap, min_idx, ids[i]);
}

int lookup_ids(gg_ctx *ctx, const gg_csr *csr, const int64_t *ids_dev, uint64_t n, uint32_t *out_dev) {
  if (n == 0) return GG_OK;
  GG_TRY(ensure_ht(ctx, const_cast<gg_csr *>(csr)));
  GG_LAUNCH(ctx, "lookup_ids", k_lookup_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ids_dev, n,
            csr->ht, csr->ht_cap, csr->ht_min_idx, out_dev);
  return GG_OK;
}

}  // namespace gg

extern "C" int gg_csr_lookup(gg_ctx *ctx, const gg_csr *csr, const int64_t *ids, uint64_t n, uint32_t *dense_out) {
  if (!ctx || !csr || csr->ctx != ctx || ((!ids || !dense_out) && n)) return GG_ERR_INVALID_ARG;
  if (n == 0) return GG_OK;
  ApiScope scope(ctx);
  GG_HIP(hipSetDevice(ctx->device));
  int64_t *ids_dev = nullptr;
  uint32_t *out_dev = nullptr;
  GG_TRY(ctx->dev_alloc((void **)&ids_dev, n * sizeof(int64_t)));
  GG_TRY(ctx->dev_alloc((void **)&out_dev, n * sizeof(uint32_t)));
  GG_HIP(hipMemcpyAsync(ids_dev, ids, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  GG_HIP(hipStreamSynchronize(ctx->stream));  // `ids` may be pageable: do not return before it was read
  GG_TRY(gg::lookup_ids(ctx, csr, ids_dev, n, out_dev));
  GG_HIP(hipMemcpyAsync(dense_out, out_dev, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  GG_HIP(hipStreamSynchronize(ctx->stream));
  return GG_OK;
}
