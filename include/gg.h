/* gg.h — C-ABI of the MI355X-native graph pattern-matching hot path ("gg" = GPU graph).
 *
 * This is the drop-in boundary between DuckDB-style host operators (C++, see
 * duckdb_pgq_amd/host/) and the hand-written HIP kernels for gfx950 (duckdb_pgq_amd/csrc/).
 * Plain pointers and sizes only: no C++ types, no torch types, no exceptions cross it.
 *
 * What each entry point replaces in the reference (cwida/duckdb-pgq.old, paths relative to
 * /root/reference; the reference has no C ABI at operator level — SURVEY.md §8b — so these are
 * the calls our PhysicalOperator subclasses make where the reference's operators call their
 * CPU data structures):
 *
 *   gg_vertices_append / gg_edges_append
 *        <- PhysicalHashJoin::Sink -> JoinHashTable::Build      src/execution/operator/join/physical_hash_join.cpp:128-154,
 *                                                               src/execution/join_hashtable.cpp:150-238
 *           (called concurrently from Sink threads, one call per <=1024-row DataChunk column set)
 *   gg_csr_build
 *        <- PhysicalHashJoin::Finalize -> JoinHashTable::Finalize/InsertHashes
 *                                                               physical_hash_join.cpp:165-185, join_hashtable.cpp:240-302
 *           (single-threaded finalize of the adjacency index keyed on the source vertex)
 *   gg_expand_khop / gg_result_fetch
 *        <- PhysicalHashJoin::Execute -> JoinHashTable::Probe + ScanStructure::NextInnerJoin (chain of k joins)
 *                                                               physical_hash_join.cpp:217-254, join_hashtable.cpp:304-476
 *           (fixed-length path expansion; fetch hands back <=1024-row slices = one DataChunk)
 *   gg_bfs64 / gg_bfs64_pairs / gg_bfs_sharded_* (one BFS over a vertex-partitioned graph, several GPUs)
 *        <- PhysicalRecursiveCTE::{Sink,GetData,ExecuteRecursivePipelines} + GroupedAggregateHashTable::FindOrCreateGroups
 *           + PhysicalHashAggregate (min(hopCount) GROUP BY start, friend)
 *                                                               src/execution/operator/set/physical_recursive_cte.cpp:48-139,
 *                                                               src/execution/aggregate_hashtable.cpp:367-504,
 *                                                               src/execution/operator/aggregate/physical_hash_aggregate.cpp:152-266
 *           (the friends/friends_shortest CTE pair of benchmark/ldbc/queries/bi-10-shortestpath.sql:8-31)
 *   gg_walk_endpoints
 *        <- PhysicalUnion + the hash-aggregate dedupe above it (friends UNION friends of friends)
 *                                                               src/execution/physical_plan/plan_distinct.cpp:12-78,
 *                                                               physical_hash_aggregate.cpp:152-266
 *   gg_vertices_from_edges
 *        <- the implicit vertex set of a join chain over an edge table alone (interactive-complex-3.sql:9-11)
 *   gg_result_filter_common_neighbour
 *        <- the six monitoredBy hash joins of benchmark/trainbenchmark/queries/connectedsegments.sql:1-25
 *   gg_csr_lookup
 *        <- JoinHashTable::Probe of plain keys                 join_hashtable.cpp:304-330
 *
 * Conventions
 *   - every int-returning function returns GG_OK (0) or a negative GG_ERR_*; the message is
 *     available from gg_last_error() (thread-local).  The C++ operators turn a non-zero status
 *     into a duckdb::IOException, mirroring how the reference reports operator errors.
 *   - all pointers are HOST memory unless the name ends in _dev; inputs are copied, outputs are
 *     written into caller-allocated arrays.
 *   - thread safety: gg_vertices_append / gg_edges_append (concurrent Sink calls), gg_result_rows /
 *     gg_result_fetch / gg_result_destroy (several pipeline threads drain one result into their own buffers)
 *     and gg_host_alloc / gg_host_free may be called concurrently on one gg_ctx; they use nothing but
 *     their own locks and thread-safe HIP calls.  Everything else on one gg_ctx is externally serialised
 *     (DuckDB calls Finalize single-threaded per operator; the source operators hold the graph's lock
 *     around expansion calls).
 *   - vertex ids are arbitrary int64 (LDBC person ids are sparse); the *dense index* of a vertex
 *     is its 0-based position in the vertex table as appended (= DuckDB rowid of the vertex row).
 *   - an edge whose src or dst id is not in the vertex table is dropped (inner-join semantics of
 *     `person p1, knows k, person p2 WHERE p1.id = k.src AND k.dst = p2.id`).
 *   - there is no CPU fallback: without a usable HIP device every compute call fails.
 */
#ifndef GG_H
#define GG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GG_OK 0
#define GG_ERR_INVALID_ARG (-1)
#define GG_ERR_HIP (-2)
#define GG_ERR_OOM (-3)
#define GG_ERR_DUPLICATE_VERTEX (-4)
#define GG_ERR_TOO_LARGE (-5) /* > 2^32-2 edge rows or distinct ids; a frontier LEVEL of >= 2^32 walks in an expansion that must
                               materialise it (k >= 4, or k = 3 from a source list with the frontier forms forced); one result
                               part of that many rows from the row-at-a-time kernels — refused, never wrapped */
#define GG_ERR_STATE (-6)
#define GG_ERR_NO_DEVICE (-7)

#define GG_MAX_HOPS 8    /* longest fixed-length pattern gg_expand_khop accepts */
#define GG_BFS_LANES 64  /* sources per bitset-BFS batch (one bit lane each) */
#define GG_CHUNK_ROWS 1024 /* STANDARD_VECTOR_SIZE, src/include/duckdb/common/vector_size.hpp:17 */

typedef struct gg_ctx gg_ctx;       /* device, stream, staging buffers; caller-owned */
typedef struct gg_csr gg_csr;       /* device-resident CSR adjacency index; caller-owned */
typedef struct gg_result gg_result; /* device-resident materialised path rows; caller-owned */

/* ---- library / context ------------------------------------------------------------------ */
const char *gg_version(void);
const char *gg_last_error(void);
int gg_device_count(int *out_count);
int gg_ctx_create(int device, gg_ctx **out);
void gg_ctx_destroy(gg_ctx *ctx);
/* Page-locked host memory for result slabs: gg_result_fetch into such a buffer runs at PCIe rate (into
 * pageable memory the runtime bounces through its own staging buffers, ~5 GB/s measured).  Freed buffers
 * are kept by the context and reused; gg_ctx_destroy releases them.  Thread-safe. */
int gg_host_alloc(gg_ctx *ctx, uint64_t bytes, void **out);
void gg_host_free(gg_ctx *ctx, void *ptr);

/* ---- staging: base-table columns -> HBM (Sink side) -------------------------------------- */
/* Append n vertex ids (vertex-table key column, in table order). */
int gg_vertices_append(gg_ctx *ctx, const int64_t *id, uint64_t n);
/* Append n edge rows.  rowid may be NULL (then the edge's 0-based append position is its rowid). */
int gg_edges_append(gg_ctx *ctx, const int64_t *src, const int64_t *dst, const int64_t *rowid, uint64_t n);
/* Block until every appended row is resident in HBM (bench: the timed region starts after this). */
int gg_staging_sync(gg_ctx *ctx);
int gg_staging_counts(gg_ctx *ctx, uint64_t *n_vertices, uint64_t *n_edges);
int gg_staging_clear(gg_ctx *ctx);

/* ---- CSR build (Finalize side) ------------------------------------------------------------ */
/* Whether later builds carry the edge rowid per CSR entry (default 1).  A MATCH that binds no edge
 * variable — e.g. count(*) or Person-KNOWS*1..2-Person returning persons — does not need it, exactly as
 * the reference's hash-join build side only carries the columns the query references; the build then
 * sorts 8 instead of 12 bytes per edge ("+8E if edge rowid kept").  With 0, gg_csr_export
 * reports -1 as rowid. */
int gg_ctx_set_edge_rowid(gg_ctx *ctx, int keep);
/* Densify ids (device hash table), histogram + prefix-scan + stable LSD radix scatter by source.
 * Within a CSR row, neighbours are in ascending edge-rowid (append) order: the build is
 * deterministic.  Staged columns stay resident, so the build can be repeated.
 * Fails with GG_ERR_DUPLICATE_VERTEX if the vertex key column is not unique.
 * Returns as soon as the build's outcome is known (duplicate ids, kept edges); the last kernels may still be
 * queued on the context's stream, behind which every later call on the context is ordered. */
int gg_csr_build(gg_ctx *ctx, gg_csr **out);
/* Multi-GPU sharding of the whole hot path with NO data-path collective: every rank stages the vertex
 * table and (at least) the edge rows with an endpoint it owns, and builds only the CSR rows of the vertices
 * it owns (owner = hash(vertex id) mod n_parts): forward rows of owned sources, reverse rows of owned
 * destinations.  Other edge rows are skipped before the id lookups, so staging the whole table works too.
 * gg_expand_khop(all sources, k_min..2, count) on a shard returns the walks whose MIDDLE vertex (1-hop
 * rows: destination) is owned; over all parts the counts add and the digests add mod 2^32.  gg_bfs_sharded_*
 * runs a BFS over the shards; other operations reject a shard (GG_ERR_STATE). */
int gg_csr_build_shard(gg_ctx *ctx, int part, int n_parts, gg_csr **out);
void gg_csr_destroy(gg_csr *csr);
/* Probe the CSR's id dictionary: dense_out[i] = dense index (vertex-table position) of ids[i], or
 * 0xFFFFFFFF if ids[i] is not a vertex — what probing the build side's hash table with n keys does in the
 * reference (JoinHashTable::Probe, src/execution/join_hashtable.cpp:304-330). */
int gg_csr_lookup(gg_ctx *ctx, const gg_csr *csr, const int64_t *ids, uint64_t n, uint32_t *dense_out);
int gg_csr_info(const gg_csr *csr, uint64_t *n_vertices, uint64_t *n_edges_kept, uint64_t *n_edges_dropped);
/* Probe of a batch of keys against the index keyed on the edge rows' source column, with the matches' rowids: for the
 * n keys (host memory; a key that is no vertex matches nothing) table 1 of *out_result gets one row (i, rowid) per
 * edge row whose source equals keys[i] — i the position in `keys` — in ascending i, a key's rows in rowid order; fetch
 * with gg_result_rows(res, 1, &m) / gg_result_fetch(res, 1, offset, max, cols[2], &got).  This is the device side of a
 * generic single-key inner hash join: JoinHashTable::Probe + ScanStructure::NextInnerJoin for one probe chunk
 * (src/execution/join_hashtable.cpp:304-476); the host operator (PhysicalGGKeyJoin) slices the probe chunk by i and
 * fetches the build side's columns by rowid.  The CSR must have been built with edge rowids kept. */
int gg_join_probe(gg_ctx *ctx, const gg_csr *csr, const int64_t *keys, uint64_t n, uint64_t *n_matches,
                  gg_result **out_result);
/* Parity export.  off: V+1 entries; nbr: E_kept dense neighbour indices; eid: E_kept edge rowids
 * (may be NULL); vid: V vertex ids by dense index (may be NULL). */
int gg_csr_export(const gg_csr *csr, int64_t *off, int64_t *nbr, int64_t *eid, int64_t *vid);

/* ---- fixed-length path expansion (k-hop MATCH) -------------------------------------------- */
typedef struct gg_khop_stats {
  uint64_t rows[GG_MAX_HOPS + 1];   /* rows[h] = number of h-hop walks emitted (h in k_min..k_max) */
  uint64_t digest[GG_MAX_HOPS + 1]; /* digest[h] = sum over those rows of the LOW 32 BITS of the gg row hash (DESIGN.md "Row digest"), mod 2^32, in a u64 field */
  uint64_t traversed_edges;         /* TE = adjacency entries read over all hops */
  uint64_t frontier_entries;        /* path prefixes whose adjacency list was expanded */
} gg_khop_stats;

/* All walks  s -> v1 -> ... -> vh  with h in [k_min, k_max] (1 <= k_min <= k_max <= GG_MAX_HOPS),
 * s drawn from the source list (with multiplicity).  src_ids == NULL: every vertex is a source.
 * Source ids absent from the vertex table contribute nothing.
 * materialise == 0: count + digest only (out_result may be NULL).
 * materialise != 0: rows are written to HBM as int64 vertex ids, one table per length h with h+1
 *                   columns, and handed back through *out_result. */
int gg_expand_khop(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, uint64_t n_src, int k_min, int k_max,
                   int materialise, gg_khop_stats *stats, gg_result **out_result);
/* Only the NUMBER of h-hop walks for h in [k_min, k_max] (rows[h]; other entries 0), computed from degrees: with
 * w_0(v) = how often v is a source and w_h(v) = sum over in-neighbours u of w_{h-1}(u), rows_h = sum_v w_{h-1}(v) *
 * outdeg(v) — from every vertex two passes over offset arrays for h <= 2, one pull over the reverse rows per further
 * hop.  This is what `SELECT count(*)` over the join chain asks for: the reference's aggregate above the hash joins
 * consumes chunk cardinalities, never the rows (src/execution/operator/aggregate/physical_simple_aggregate.cpp, fed by
 * ScanStructure::NextInnerJoin, src/execution/join_hashtable.cpp:442-476).  Equal to gg_khop_stats.rows of
 * gg_expand_khop with the same arguments (also on a shard: the walks whose middle vertex is owned); no digest. */
int gg_khop_count(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, uint64_t n_src, int k_min, int k_max,
                  uint64_t *rows /* GG_MAX_HOPS + 1 entries */);
/* Same, sources = dense vertex indices [src_lo, src_hi) — the multi-GPU sharding entry point
 * (each rank takes one contiguous range; no data-path collective). */
int gg_expand_khop_range(gg_ctx *ctx, const gg_csr *csr, uint64_t src_lo, uint64_t src_hi, int k_min, int k_max,
                         int materialise, gg_khop_stats *stats, gg_result **out_result);
/* Split [0,V) into n_parts contiguous source ranges of near-equal 2-hop work (sum over u of
 * sum over v in adj(u) of (1+deg(v))).  bounds: n_parts+1 entries. */
int gg_khop_partition(gg_ctx *ctx, const gg_csr *csr, int n_parts, uint64_t *bounds);

/* 2-hop walks (and, if k_min == 1, the 1-hop rows) of ALL sources whose MIDDLE vertex (1-hop rows:
 * destination) has dense index in [mid_lo, mid_hi).  k_max must be 2.  Disjoint ranges partition the
 * result of gg_expand_khop(all sources, k_min..2): counts add, digests add mod 2^32.  This is the
 * multi-GPU sharding entry point of the 2-hop product kernel (reads each CSR row once per shard). */
int gg_expand_khop_mid(gg_ctx *ctx, gg_csr *csr, uint64_t mid_lo, uint64_t mid_hi, int k_min, int k_max,
                       gg_khop_stats *stats);
/* The same rows MATERIALISED (int64 id columns in HBM, fetched with gg_result_fetch): the 2-hop rows u -> x -> w of
 * all sources with x in [mid_lo, mid_hi) and, if k_min == 1, the 1-hop rows u -> x into the range.  Disjoint ranges
 * partition the materialised result of gg_expand_khop(all sources, k_min..2) — how a result larger than device
 * memory (SF100: 12.8 G rows, 306 GB) is produced part by part, each part through the product kernel.  On a shard
 * (gg_csr_build_shard) the rows are those whose middle vertex is owned AND in the range: N ranks that each materialise
 * [0, V) of their shard — in gg_khop_partition_mid parts if need be — produce the whole result once, no exchange. */
int gg_expand_khop_mid_result(gg_ctx *ctx, gg_csr *csr, uint64_t mid_lo, uint64_t mid_hi, int k_min,
                              gg_khop_stats *stats /* nullable: no counts + digests in front of the rows */,
                              gg_result **out_result);
/* gg_expand_khop(all sources, k_min..2, count) with the result LEFT ON THE DEVICE and no host synchronisation: six
 * uint64 words (rows of 1-hop walks, rows of 2-hop walks, digest 1, digest 2, traversed edges, frontier entries; the
 * digests 32-bit sums in the low half) in a buffer owned by the context (*stats_dev; overwritten by the next such call).
 * For the ranks of a sharded query (gg_csr_build_shard): each rank's words are added by ONE collective on the device
 * (RCCL SUM through torch.distributed on a view of that memory — the library owns no communicator) and cross to the host
 * once, instead of host -> device -> all-reduce -> host per rank.  Order the collective's stream behind the library's
 * with gg_stream_wait. */
int gg_expand_khop_dev(gg_ctx *ctx, gg_csr *csr, int k_min, void **stats_dev);
/* Stream ordering without the host: direction 0 — everything queued on `other_stream` (a hipStream_t, e.g. torch's
 * current stream) from now on waits for everything queued on the context's stream so far; direction 1 — the reverse. */
int gg_stream_wait(gg_ctx *ctx, void *other_stream, int direction);
/* Split [0,V) into n_parts contiguous middle-vertex ranges of near-equal product work. */
int gg_khop_partition_mid(gg_ctx *ctx, gg_csr *csr, int n_parts, uint64_t *bounds);

int gg_result_rows(const gg_result *res, int hops, uint64_t *n_rows);
/* Copy rows [offset, offset+max_rows) of the h-hop table into cols[0..h] (host arrays of >= max_rows).  Destinations
 * in page-locked memory (gg_host_alloc) are filled over the context's fetch lanes (concurrent callers
 * overlap their copies: ~44 GB/s in 2 MB pieces); any other destination over the library's stream. */
int gg_result_fetch(const gg_result *res, int hops, uint64_t offset, uint32_t max_rows, int64_t *const *cols,
                    uint32_t *n_out);
/* k-hop walks with the ROWID of every edge taken (src_ids NULL: from every vertex): table k of the result has the id
 * columns v0..vk (gg_result_fetch) and the edge columns e1..ek (gg_result_fetch_edges: the rowid the Sink passed with
 * the edge row, or its append position if it passed none).  This is what a late join with the edge table's payload
 * columns needs (the reference gathers build-side columns per match, src/execution/join_hashtable.cpp:466-473).  The
 * CSR must have been built with edge rowids kept (gg_ctx_set_edge_rowid(ctx, 1), the default).  stats: rows[k] only. */
int gg_expand_khop_edges(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, uint64_t n_src, int k,
                         gg_khop_stats *stats, gg_result **out_result);
/* Copy rows [offset, offset+max_rows) of the edge columns e1..e_hops into ecols[0..hops-1]. */
int gg_result_fetch_edges(const gg_result *res, int hops, uint64_t offset, uint32_t max_rows, int64_t *const *ecols,
                          uint32_t *n_out);
/* Checksum of the `hops`-hop rows (1 <= hops <= 4) as they stand in HBM: every id of every row is mapped back to its
 * dense index and the rows' hashes are summed exactly as gg_khop_stats.digest[hops] sums them, so a materialising
 * expansion can be compared with the count-only expansion (and with the oracle) over ALL its rows without fetching
 * them.  Fails with GG_ERR_STATE if a row holds an id that is not a vertex of `csr`. */
int gg_result_digest(gg_ctx *ctx, const gg_csr *csr, const gg_result *res, int hops, uint64_t *n_rows,
                     uint64_t *digest);
void gg_result_destroy(gg_result *res);
/* gg_expand_khop with the result left on the device for further operators (same arguments; only the
 * handle is returned, nothing is copied to the host). */
int gg_expand_khop_result(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, uint64_t n_src, int k_min, int k_max,
                          gg_khop_stats *stats, gg_result **out_result);
/* Same-neighbour filter (Train Benchmark ConnectedSegments: six segments monitored by one sensor).
 * Input: the `hops`-hop table of `res` (columns v0..vh).  `filter` is a CSR over a vertex table that
 * contains the path vertices' ids and the filter targets (e.g. TrackElements and Sensors; edges =
 * monitoredBy).  Output table (fetch it with hops+1): one row (w, v0..vh) for every filter neighbour w
 * common to ALL of v0..vh, with the multiplicity the chained joins would give. */
int gg_result_filter_common_neighbour(gg_ctx *ctx, const gg_result *res, int hops, const gg_csr *filter,
                                      gg_result **out);
/* Forget the staged edge rows but keep the staged vertex table (several edge tables, one vertex set). */
int gg_staging_clear_edges(gg_ctx *ctx);
/* Replace the staged vertex table by the distinct endpoint ids of the staged edge rows, in ascending
 * (signed) order — the vertex set a join chain over the edge table ALONE ranges over
 * (`knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id`, benchmark/ldbc/queries/
 * interactive-complex-3.sql:9-11: no vertex table in the pattern, so every id that occurs is a vertex
 * and no edge row is dropped).  keep_staged_vertices != 0: the new table is the UNION of the ids already
 * staged as vertices and the endpoints (several edge tables over one id space: derive from the first,
 * gg_staging_clear_edges, stage the next, derive again with keep).  Computed on the device (hash set +
 * radix sort); *n_vertices (nullable) receives the number of distinct ids. */
int gg_vertices_from_edges(gg_ctx *ctx, int keep_staged_vertices, uint64_t *n_vertices);

/* ---- 64-lane bitset BFS (shortest path length) --------------------------------------------- */
typedef struct gg_bfs_stats {
  uint32_t levels;               /* levels expanded */
  uint64_t traversed_edges;      /* sum over levels of deg(v) over vertices active in any lane */
  uint64_t active_vertices;      /* sum over levels of |{v : frontier[v] != 0}| */
  uint64_t reached_pairs;        /* (lane, vertex) pairs with dist >= 0 */
} gg_bfs_stats;

/* Lane i starts at src_ids[i] (n_src <= 64).  out_dist[i*n + j] = length of the shortest walk from
 * src_ids[i] to target j, or -1 if none of length <= max_hops (max_hops < 0: run to fixpoint).
 * Targets: dst_ids == NULL -> every vertex in vertex-table order (n = V); else the n_dst given ids
 * (ids absent from the vertex table get -1).  A source absent from the vertex table reaches nothing.
 * out_dist == NULL: run the BFS and fill *stats only (no device-to-host copy of the distances).
 * Equals, for the reached pairs, the reference relation
 *   SELECT startPerson, friend, min(hopCount) FROM friends GROUP BY startPerson, friend
 * with the recursion bound `f.hopCount < max_hops` (bi-10-shortestpath.sql:8-31). */
int gg_bfs64(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, int n_src, int max_hops, const int64_t *dst_ids,
             uint64_t n_dst, int32_t *out_dist, gg_bfs_stats *stats);
/* gg_bfs64 with the answer compacted on the device: one row (source id, vertex id, distance) per reached
 * pair — the friends_shortest relation of bi-10-shortestpath.sql:26-31 for this batch of sources — instead
 * of the dense n_src x V matrix.  The rows are table 2 of *out_result: gg_result_rows(res, 2, &n),
 * gg_result_fetch(res, 2, offset, n, cols[3], &got); row order is unspecified.  Sources that are not
 * vertices contribute no row. */
int gg_bfs64_pairs(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, int n_src, int max_hops,
                   gg_bfs_stats *stats, gg_result **out_result);
/* The same rows as one 8-byte word each — lane << 58 | distance << 32 | dense vertex index, lane = index
 * into src_ids — in table 0 of *out_result (one column): a third of the bytes over PCIe for hosts that hold
 * the source list and the vertex ids (gg_csr_export) themselves. */
int gg_bfs64_pairs_packed(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, int n_src, int max_hops,
                          gg_bfs_stats *stats, gg_result **out_result);

/* Distinct endpoints of the walks of 1..k_max edges that start at any of the given sources — the device form of
 *   SELECT dst FROM e WHERE src = C  UNION  SELECT e2.dst FROM e e1, e e2 WHERE e1.src = C AND e1.dst = e2.src
 * (the friends / friends-of-friends table of benchmark/ldbc/queries/interactive-complex-3.sql:3-12), which the
 * reference evaluates as PhysicalUnion under a hash-aggregate dedupe
 * (src/execution/operator/set/physical_union.cpp, src/execution/operator/aggregate/physical_hash_aggregate.cpp:152-266).
 * Table 1 of *out_result has one row (vertex id, mask) per vertex that ends at least one such walk: bit h of mask
 * is set iff the vertex ends a walk of exactly h edges (walks, not shortest paths: a vertex can carry several bits,
 * and a source on a cycle is its own endpoint).  Rows come in vertex-table order; sources that are not vertices
 * contribute nothing.  Fetch with gg_result_rows(res, 1, &n) / gg_result_fetch(res, 1, offset, n, cols[2], &got). */
int gg_walk_endpoints(gg_ctx *ctx, const gg_csr *csr, const int64_t *src_ids, uint64_t n_src, int k_max,
                      gg_result **out_result);

/* ---- graph-sharded 64-lane BFS (one shard of the graph per GPU) ------------------------------- */
/* The layout north_star names for graphs that do not fit one GPU: `shard` comes from
 * gg_csr_build_shard; every rank holds the whole frontier (one uint64 of 64 lanes per vertex) and owns the
 * seen words, distances and result rows of its vertices.  Per level, on every rank:
 *     gg_bfs_sharded_expand   the next frontier words of the owned vertices (zero elsewhere): pulled over
 *                             their reverse rows when the frontier is heavy, pushed along the same edges
 *                             grouped by source when it is light (each rank decides from its own edges;
 *                             either way only owned words are written);
 *                             *next_words_dev points at the n_words uint64 words in HBM
 *     (exchange)              combine the ranks' words — disjoint supports, so a SUM all-reduce (RCCL through
 *                             torch.distributed on a view of that memory) is their OR; the library does
 *                             not own a communicator.  gg_bfs_sharded_words copies the words to / from the
 *                             host for hosts without a device-side collective (and for tests)
 *     gg_bfs_sharded_commit   the combined words become the next level's frontier
 * until no rank reports new pairs (or the hop bound is reached).  gg_bfs_sharded_pairs then returns the
 * rank's share of the (source, vertex, distance) rows, in the layout of gg_bfs64_pairs; their union over
 * the ranks is gg_bfs64_pairs on the whole graph.  A whole (unsharded) CSR is accepted as the 1-rank case. */
typedef struct gg_bfs_run gg_bfs_run;
int gg_bfs_sharded_begin(gg_ctx *ctx, const gg_csr *shard, const int64_t *src_ids, int n_src, gg_bfs_run **out);
int gg_bfs_sharded_expand(gg_bfs_run *run, void **next_words_dev, uint64_t *n_words, uint64_t *new_pairs_local);
int gg_bfs_sharded_words(gg_bfs_run *run, uint64_t *host_words, int write_back);
int gg_bfs_sharded_commit(gg_bfs_run *run);
int gg_bfs_sharded_pairs(gg_bfs_run *run, gg_result **out_result);
/* how many levels this rank pushed / pulled so far (diagnostic) */
int gg_bfs_sharded_levels(const gg_bfs_run *run, uint64_t *push_levels, uint64_t *pull_levels);
void gg_bfs_sharded_end(gg_bfs_run *run);

/* ---- in-library kernel timing (HIP events on the library's own stream) ---------------------- */
/* Testing knob: 1 forces gg_expand_khop onto the frontier kernels even where a product kernel applies; 2 and 3 keep
 * the all-sources product kernels out of it (as 1) but take a product form of an explicit frontier's last hops at any
 * size — 2: the last hop (pairs, sort by last vertex, fold; by itself from 65 536 frontier entries on), 3: the last two
 * hops (the frontier sorted by last vertex, k_expand_mid3's tiles over the reverse entries; by itself when the
 * frontier's children are an eighth of the edge table or more); 0 restores normal operation.  All must give identical
 * results. */
int gg_debug_force_frontier(gg_ctx *ctx, int on);
/* Testing knob: force gg_csr_build onto the multi-pass LSD build that graphs of more than 2^22 vertices
 * (and shard builds) take; both builds must export identical arrays. */
int gg_debug_force_legacy_build(gg_ctx *ctx, int on);
/* Testing knob: how the bucketed build ranks entries inside a wavefront.  0 (default): probe once whether one
 * ds_add_rtn serves colliding lanes in lane order and use it if so; 1: use it; 2: match masks by ballots.  Both
 * must export identical arrays. */
int gg_debug_rank_mode(gg_ctx *ctx, int mode);
/* Testing knob (fault injection): tile `mute_tile` of every chained prefix scan never publishes its sum and the
 * tiles behind it give up after `spin_limit` polls instead of 2^24; the call that ran the scan must then
 * fail with GG_ERR_HIP instead of returning a wrong result.  spin_limit 0 and mute_tile UINT64_MAX restore
 * normal operation. */
int gg_debug_scan_fault(gg_ctx *ctx, uint32_t spin_limit, uint64_t mute_tile);
/* Testing knob: the expansion kernels split their grids into launches of at most `max_tiles` workgroups (the
 * runtime takes gridDim.x * blockDim.x in 32 bits, so a 3-hop expansion over more than ~1.3e10 2-hop rows runs
 * as several launches); 0 restores the hardware bound.  Results must not depend on it. */
int gg_debug_max_grid_tiles(gg_ctx *ctx, uint64_t max_tiles);
/* Every testing knob above and gg_ctx_set_edge_rowid back to its default (a test suite that shares one context
 * calls this between tests). */
int gg_debug_reset(gg_ctx *ctx);
/* Diagnostics of the pool's placement of large result columns (DESIGN.md 4.2 "Placement"): how many column sets this
 * context has placed by probing, and how many of the three pairs of the LAST set lie in different memory ranks (3 = every
 * column in its own rank class, the 7.2 TB/s case; 2 = two classes, 7.0; 0 = one class or never probed).  Read-only. */
int gg_debug_placement(gg_ctx *ctx, uint64_t *sets_built, uint64_t *fast_pairs_of_last_set);

int gg_profile_enable(gg_ctx *ctx, int on);
/* Time only the kernels named in the comma-separated list (NULL: every kernel).  Two event records per
 * launch are not free — ~35 launches per build + expansion cost 0.3 ms of a 3.9 ms step — so a timed
 * region restricts them to the kernels it reports on. */
int gg_profile_select(gg_ctx *ctx, const char *names);
int gg_profile_reset(gg_ctx *ctx);
/* Number of distinct kernels seen; then per index: name, launches, total milliseconds. */
int gg_profile_count(gg_ctx *ctx, int *n);
int gg_profile_get(gg_ctx *ctx, int index, const char **name, uint64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* GG_H */
