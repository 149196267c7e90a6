/* gg_oracle.h — CPU oracle for the graph pattern-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (duckdb_pgq_amd/, include/) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do.  It restates, in plain C, the algorithms the reference (cwida/duckdb-pgq.old, a DuckDB
 * v0.3.1 fork) runs when these graph workloads are written as SQL:
 *   - equi hash join build/probe         src/execution/join_hashtable.cpp:150-476
 *   - recursive CTE with UNION dedupe     src/execution/operator/set/physical_recursive_cte.cpp:48-139
 *   - min(hop) GROUP BY (start, friend)   src/execution/operator/aggregate/physical_hash_aggregate.cpp:152-266
 * plus a direct CSR formulation of the same relations that is fast enough for full-size
 * inputs (validated against the join formulation and against the compiled reference,
 * oracle/_ref/libduckdb.so — see tests/test_oracle_*.py and tests/golden/).
 *
 * Parity status: PINNED — by the reference's own golden rows
 * (benchmark/trainbenchmark/connectedsegments.benchmark:34-38) and by fixtures produced by the
 * compiled reference on seeded synthetic LDBC-shaped tables (tests/golden/make_golden.py).
 */
#ifndef GG_ORACLE_H
#define GG_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_HOPS 8

/* ---- row digest shared with the HIP path (definition: DESIGN.md §"Row digest") ------------ */
uint64_t orc_fmix64(uint64_t x);
/* digest accumulation: (a + low32(b)) mod 2^32 */
uint64_t orc_dsum_add(uint64_t a, uint64_t b);
/* digest value of one walk given as dense vertex indices d[0..h] (h >= 1) */
uint64_t orc_row_hash(const uint32_t *d, int h);

/* ---- growable int64 row table -------------------------------------------------------------- */
typedef struct orc_rows {
  int ncols;
  uint64_t n, cap;
  int64_t *data; /* row-major: data[r*ncols + c] */
} orc_rows;
void orc_rows_free(orc_rows *t);
/* sort rows lexicographically (for order-insensitive comparison) */
void orc_rows_sort(orc_rows *t);

/* ---- restatement of JoinHashTable (single BIGINT key, INNER join) -------------------------- */
typedef struct orc_jht {
  uint64_t count, capacity, bitmask;
  int64_t *heads;      /* capacity entries: build row index of chain head, -1 = empty pointer   */
  int64_t *next;       /* count entries: next build row in chain, -1 = end                      */
  const int64_t *keys; /* build keys (borrowed)                                                 */
} orc_jht;
int orc_jht_build(orc_jht *ht, const int64_t *keys, uint64_t n);
void orc_jht_free(orc_jht *ht);
/* Probe with m keys; appends (probe_idx, build_idx) pairs to out (ncols = 2) in the order the
 * reference's ScanStructure::NextInnerJoin emits them within each 1024-key chunk. */
int orc_jht_probe(const orc_jht *ht, const int64_t *probe_keys, uint64_t m, orc_rows *out);

/* ---- k-hop MATCH through the join formulation (reference operators) ------------------------ */
/* Walks s -> v1 .. -> vh, every vertex in the vertex table, s from sources (NULL = all vertices).
 * out[h] (h in k_min..k_max) receives rows of h+1 DENSE vertex indices (vertex-table rowids). */
int orc_khop_join(const int64_t *vid, uint64_t V, const int64_t *esrc, const int64_t *edst, uint64_t E,
                  const int64_t *sources, uint64_t n_src, int k_min, int k_max, orc_rows *out /*[ORC_MAX_HOPS+1]*/);

/* ---- recursive-CTE shortest path (reference operators) ------------------------------------- */
/* Restates friends / friends_shortest of bi-10-shortestpath.sql:8-31 with the seed widened to
 * n_src sources and the bound `hopCount < max_hops`.  out rows: (start_id, friend_id, min_hop). */
int orc_cte_shortest(const int64_t *vid, uint64_t V, const int64_t *esrc, const int64_t *edst, uint64_t E,
                     const int64_t *sources, uint64_t n_src, int max_hops, orc_rows *out);

/* ---- direct CSR formulation (fast path for full-size inputs) -------------------------------- */
typedef struct orc_csr {
  uint64_t V, E, dropped;
  int64_t *off;  /* V+1 */
  uint32_t *nbr; /* E dense */
  int64_t *eid;  /* E rowids */
  int64_t *vid;  /* V ids (copy) */
} orc_csr;
/* rowid may be NULL (then rowid = position).  Returns 0, or -4 on duplicate vertex id. */
int orc_csr_build(orc_csr *g, const int64_t *vid, uint64_t V, const int64_t *esrc, const int64_t *edst,
                  const int64_t *rowid, uint64_t E);
void orc_csr_free(orc_csr *g);
/* id -> dense index or -1 */
int64_t orc_csr_lookup(const orc_csr *g, int64_t id);

typedef struct orc_khop_stats {
  uint64_t rows[ORC_MAX_HOPS + 1];
  uint64_t digest[ORC_MAX_HOPS + 1];
  uint64_t traversed_edges;
  uint64_t frontier_entries;
} orc_khop_stats;
/* sources: dense indices (with multiplicity), or NULL for the range [lo,hi).  threads <= 0: all. */
int orc_khop_csr(const orc_csr *g, const uint32_t *src_dense, uint64_t n_src, uint64_t lo, uint64_t hi, int k_min,
                 int k_max, int threads, orc_khop_stats *st);
/* Materialise the same walks as rows of ids (small inputs only). out[h] has h+1 columns. */
int orc_khop_csr_rows(const orc_csr *g, const uint32_t *src_dense, uint64_t n_src, uint64_t lo, uint64_t hi,
                      int k_min, int k_max, orc_rows *out /*[ORC_MAX_HOPS+1]*/);

typedef struct orc_bfs_stats {
  uint32_t levels;
  uint64_t traversed_edges, active_vertices, reached_pairs;
} orc_bfs_stats;
/* Level-synchronous 64-lane bitset BFS.  src_dense[i] < 0: lane reaches nothing.
 * dist: n_src * V int32 (vertex-table order), -1 = not reached within max_hops (<0: unbounded). */
int orc_bfs64_csr(const orc_csr *g, const int64_t *src_dense, int n_src, int max_hops, int32_t *dist,
                  orc_bfs_stats *st);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
