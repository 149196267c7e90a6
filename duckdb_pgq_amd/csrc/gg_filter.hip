// gg_filter.hip — same-neighbour filter over materialised path rows.
//
// Train Benchmark ConnectedSegments (benchmark/trainbenchmark/queries/connectedsegments.sql:1-25) is a
// 5-edge path over connectsTo whose six segments must all be monitored by the SAME sensor: in the
// reference, six more hash joins with monitoredBy plus five equality predicates on the sensor column
// (PhysicalHashJoin probes, src/execution/operator/join/physical_hash_join.cpp:217-254).  Here the
// path rows are already in HBM (gg_expand_khop, materialised); one pass keeps, for every row, each
// neighbour w of the row's FIRST vertex in the filter graph that is also a neighbour of all the other
// vertices of the row (join multiplicities multiply, as the chained joins would produce).
//   k_filter_count  per row: walk adj_f(v0) (a few entries), membership tests by linear scan
//   scan            exclusive prefix of the per-row output counts
//   k_filter_fill   same walk, writing (w, v0..vh) as int64 ids
#include "gg_internal.h"

using namespace gg;

namespace gg {

struct RowCols {
  const int64_t *c[GG_MAX_HOPS + 1];
};
struct OutCols {
  int64_t *c[GG_MAX_HOPS + 2];
};

// number of filter edges v -> w
__device__ __forceinline__ uint32_t mult_of(const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                            uint32_t v, uint32_t w) {
  uint32_t m = 0;
  for (uint32_t i = off[v]; i < off[v + 1]; i++) m += (nbr[i] == w);
  return m;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_filter(RowCols in, int ncols, uint64_t n_rows, const HtSlot *__restrict__ ht,
                                                uint64_t cap, int64_t min_idx,
                                                const uint32_t *__restrict__ off, const uint32_t *__restrict__ nbr,
                                                const int64_t *__restrict__ vid, uint64_t *__restrict__ counts,
                                                const uint64_t *__restrict__ offsets, OutCols out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  uint32_t d[GG_MAX_HOPS + 1];
  bool ok = true;
  for (int c = 0; c < ncols; c++) {
    d[c] = ht_lookup(ht, cap, min_idx, in.c[c][r]);  // row vertex -> dense index in the FILTER graph
    ok = ok && d[c] != INVALID_U32;
  }
  uint64_t n = 0, o = FILL ? offsets[r] : 0;
  if (ok) {
    for (uint32_t i = off[d[0]]; i < off[d[0] + 1]; i++) {
      const uint32_t w = nbr[i];
      uint64_t m = 1;
      for (int c = 1; c < ncols && m; c++) m *= mult_of(off, nbr, d[c], w);
      if (FILL) {
        for (uint64_t k = 0; k < m; k++, o++) {
          out.c[0][o] = vid[w];
          for (int c = 0; c < ncols; c++) out.c[c + 1][o] = in.c[c][r];
        }
      }
      n += m;
    }
  }
  if (!FILL) counts[r] = n;
}

}  // namespace gg

extern "C" int gg_result_filter_common_neighbour(gg_ctx *ctx, const gg_result *res, int hops, const gg_csr *filter,
                                                 gg_result **out) {
  if (!ctx || !res || !filter || !out || res->ctx != ctx || filter->ctx != ctx || hops < res->k_min ||
      hops > res->k_max || hops + 1 > GG_MAX_HOPS || filter->n_parts > 1) {
    set_error("gg_result_filter_common_neighbour: bad argument");
    return GG_ERR_INVALID_ARG;
  }
  *out = nullptr;
  ApiScope scope(ctx);
  GG_HIP(hipSetDevice(ctx->device));
  const uint64_t n_rows = res->rows[hops];
  const int ncols = hops + 1;
  gg_result *o = new gg_result();
  o->ctx = ctx;
  o->k_min = o->k_max = hops + 1;  // the table with hops+2 columns: (w, v0..vh)
  struct Guard {
    gg_result *r;
    bool armed = true;
    ~Guard() {
      if (armed) gg_result_destroy(r);
    }
  } guard{o};
  RowCols in;
  for (int c = 0; c < ncols; c++) in.c[c] = res->cols[hops][c];
  OutCols oc;
  for (int c = 0; c < GG_MAX_HOPS + 2; c++) oc.c[c] = nullptr;
  uint64_t total = 0;
  if (n_rows) {
    GG_TRY(ensure_ht(ctx, const_cast<gg_csr *>(filter)));
    uint64_t *counts = nullptr, *tot = nullptr;
    GG_TRY(ctx->dev_alloc((void **)&counts, (n_rows + 1) * sizeof(uint64_t)));
    GG_TRY(ctx->dev_alloc((void **)&tot, sizeof(uint64_t)));
    const unsigned grid = (unsigned)((n_rows + 255) / 256);
    GG_LAUNCH(ctx, "filter_count", (k_filter<false>), dim3(grid), dim3(256), 0, in, ncols, n_rows, filter->ht,
              filter->ht_cap, filter->ht_min_idx, filter->off, filter->nbr, filter->vid, counts,
              (const uint64_t *)nullptr, oc);
    GG_TRY(scan_exclusive_u64(ctx, counts, counts, n_rows, tot));
    GG_HIP(hipMemcpyAsync(ctx->pin_scratch, tot, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    GG_TRY(scan_error_fetch(ctx));
    GG_HIP(hipStreamSynchronize(ctx->stream));
    GG_TRY(scan_error_test(ctx));
    total = ctx->pin_scratch[0];
    for (int c = 0; c <= ncols; c++) {
      GG_TRY(ctx->dev_alloc((void **)&o->cols[hops + 1][c], (total ? total : 1) * sizeof(int64_t)));
      ctx->keep(o->cols[hops + 1][c]);
      oc.c[c] = o->cols[hops + 1][c];
    }
    if (total)
      GG_LAUNCH(ctx, "filter_fill", (k_filter<true>), dim3(grid), dim3(256), 0, in, ncols, n_rows, filter->ht,
                filter->ht_cap, filter->ht_min_idx, filter->off, filter->nbr, filter->vid,
                (uint64_t *)nullptr, (const uint64_t *)counts, oc);
    GG_HIP(hipStreamSynchronize(ctx->stream));
    ctx->dev_free(counts);
    ctx->dev_free(tot);
  } else {
    for (int c = 0; c <= ncols; c++) {
      GG_TRY(ctx->dev_alloc((void **)&o->cols[hops + 1][c], sizeof(int64_t)));
      ctx->keep(o->cols[hops + 1][c]);
    }
  }
  o->rows[hops + 1] = total;
  guard.armed = false;
  *out = o;
  return GG_OK;
}
