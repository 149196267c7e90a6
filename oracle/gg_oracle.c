/* gg_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY; see gg_oracle.h for scope and status).
 *
 * Every function names the reference code (paths relative to /root/reference) it restates.
 * Nothing here is copied from the reference: the reference is C++ over DataChunks/Vectors and
 * row-store pages; this is a plain-C restatement over int64 arrays that produces the same
 * relations.
 */
#define _GNU_SOURCE
#include "gg_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* Row digest (ours; DESIGN.md "Row digest").  Cheap per leaf entry, non-separable per row.   */
/* ------------------------------------------------------------------------------------------ */
#define ORC_K32 0x9E3779B1u
#define ORC_GOLD 0x9E3779B97F4A7C15ULL

uint64_t orc_fmix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
static inline uint64_t orc_q(uint64_t p, int j) { return orc_fmix64(p + ORC_GOLD * (uint64_t)(j + 1)); }
static inline uint64_t orc_leaf(uint64_t q, uint32_t d) { return q ^ ((uint64_t)d * (uint64_t)ORC_K32); }

/* digest sums are 32-bit: the low halves of the row hashes, summed mod 2^32 */
uint64_t orc_dsum_add(uint64_t a, uint64_t b) { return (uint64_t)(uint32_t)((uint32_t)a + (uint32_t)b); }

uint64_t orc_row_hash(const uint32_t *d, int h) {
  uint64_t p = d[0]; /* P_0 := d0 */
  for (int j = 0; j < h; j++) p = orc_leaf(orc_q(p, j), d[j + 1]);
  return p;
}

/* ------------------------------------------------------------------------------------------ */
/* growable row table                                                                         */
/* ------------------------------------------------------------------------------------------ */
static int rows_init(orc_rows *t, int ncols) {
  t->ncols = ncols;
  t->n = 0;
  t->cap = 0;
  t->data = NULL;
  return 0;
}
static inline int rows_reserve(orc_rows *t, uint64_t extra) {
  if (t->n + extra <= t->cap) return 0;
  uint64_t nc = t->cap ? t->cap * 2 : 1024;
  while (nc < t->n + extra) nc *= 2;
  int64_t *p = (int64_t *)realloc(t->data, (size_t)nc * (size_t)t->ncols * sizeof(int64_t));
  if (!p) return -3;
  t->data = p;
  t->cap = nc;
  return 0;
}
void orc_rows_free(orc_rows *t) {
  if (!t) return;
  free(t->data);
  t->data = NULL;
  t->n = t->cap = 0;
}
static int g_sort_ncols;
static int rows_cmp(const void *a, const void *b) {
  const int64_t *x = (const int64_t *)a, *y = (const int64_t *)b;
  for (int c = 0; c < g_sort_ncols; c++) {
    if (x[c] < y[c]) return -1;
    if (x[c] > y[c]) return 1;
  }
  return 0;
}
void orc_rows_sort(orc_rows *t) {
  if (!t || t->n < 2) return;
  g_sort_ncols = t->ncols;
  qsort(t->data, (size_t)t->n, (size_t)t->ncols * sizeof(int64_t), rows_cmp);
}

/* ------------------------------------------------------------------------------------------ */
/* JoinHashTable restatement                                                                  */
/* ------------------------------------------------------------------------------------------ */
/* hash: murmurhash64(x) = x * 0xbf58476d1ce4e5b9        src/include/duckdb/common/types/hash.hpp:22-24 */
static inline uint64_t ref_hash64(int64_t v) { return (uint64_t)v * 0xbf58476d1ce4e5b9ULL; }

static uint64_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

/* JoinHashTable::Finalize + InsertHashes               src/execution/join_hashtable.cpp:240-302
 * capacity = NextPowerOfTwo(max(2*count, BLOCK_SIZE/8 + 1)), BLOCK_SIZE = 262144 - 8
 * (src/include/duckdb/common/constants.hpp:74-76); rows are inserted in build order and each
 * insert PREPENDS to its bucket chain (:251-259), so a chain lists build rows newest-first. */
int orc_jht_build(orc_jht *ht, const int64_t *keys, uint64_t n) {
  memset(ht, 0, sizeof(*ht));
  const uint64_t block_size = 262144 - 8;
  uint64_t want = n * 2;
  if (want < block_size / 8 + 1) want = block_size / 8 + 1;
  ht->capacity = next_pow2(want);
  ht->bitmask = ht->capacity - 1;
  ht->count = n;
  ht->keys = keys;
  ht->heads = (int64_t *)malloc((size_t)ht->capacity * sizeof(int64_t));
  ht->next = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
  if (!ht->heads || !ht->next) return -3;
  for (uint64_t i = 0; i < ht->capacity; i++) ht->heads[i] = -1;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t slot = ref_hash64(keys[i]) & ht->bitmask; /* ApplyBitmask :77-79 */
    ht->next[i] = ht->heads[slot];
    ht->heads[slot] = (int64_t)i;
  }
  return 0;
}
void orc_jht_free(orc_jht *ht) {
  free(ht->heads);
  free(ht->next);
  memset(ht, 0, sizeof(*ht));
}

/* JoinHashTable::Probe + ScanStructure::{NextInnerJoin,ScanInnerJoin,AdvancePointers}
 *                                                       src/execution/join_hashtable.cpp:304-476
 * Probe keys are consumed in chunks of STANDARD_VECTOR_SIZE = 1024.  Within a chunk the scan
 * keeps one chain pointer per still-active probe row; each round emits the rows whose current
 * chain entry matches the key (RowOperations::Match), then advances every active pointer, and
 * drops the ones that reached the chain end — so all first matches come out before all second
 * matches, etc.  (The physical operator additionally caps each emitted DataChunk at 1024 rows;
 * that only slices this sequence, it does not reorder it.) */
int orc_jht_probe(const orc_jht *ht, const int64_t *probe_keys, uint64_t m, orc_rows *out) {
  if (out->ncols != 2) return -1;
  int64_t ptr[1024];
  uint32_t sel[1024];
  for (uint64_t base = 0; base < m; base += 1024) {
    uint32_t cnt = (uint32_t)((m - base < 1024) ? (m - base) : 1024);
    uint32_t active = 0;
    for (uint32_t i = 0; i < cnt; i++) {
      uint64_t slot = ref_hash64(probe_keys[base + i]) & ht->bitmask;
      ptr[i] = ht->heads[slot];
      if (ptr[i] >= 0) sel[active++] = i;
    }
    while (active) {
      if (rows_reserve(out, active)) return -3;
      for (uint32_t a = 0; a < active; a++) {
        uint32_t i = sel[a];
        if (ht->keys[ptr[i]] == probe_keys[base + i]) {
          out->data[out->n * 2] = (int64_t)(base + i);
          out->data[out->n * 2 + 1] = ptr[i];
          out->n++;
        }
      }
      uint32_t na = 0;
      for (uint32_t a = 0; a < active; a++) {
        uint32_t i = sel[a];
        ptr[i] = ht->next[ptr[i]];
        if (ptr[i] >= 0) sel[na++] = i;
      }
      active = na;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* k-hop MATCH as the reference executes it: a chain of hash joins                            */
/*   person p0 JOIN knows k1 ON p0.id = k1.src JOIN person p1 ON k1.dst = p1.id JOIN knows k2 */
/*   ON p1.id = k2.src ...   (2-hop idiom: benchmark/ldbc/queries/interactive-complex-3.sql:11)*/
/* ------------------------------------------------------------------------------------------ */
int orc_khop_join(const int64_t *vid, uint64_t V, const int64_t *esrc, const int64_t *edst, uint64_t E,
                  const int64_t *sources, uint64_t n_src, int k_min, int k_max, orc_rows *out) {
  if (k_min < 1 || k_max < k_min || k_max > ORC_MAX_HOPS) return -1;
  for (int h = 0; h <= ORC_MAX_HOPS; h++) rows_init(&out[h], h + 1);
  orc_jht person, knows;
  int rc;
  if ((rc = orc_jht_build(&person, vid, V))) return rc;
  if ((rc = orc_jht_build(&knows, esrc, E))) return rc;

  /* seed: all persons, or sources JOIN person */
  orc_rows cur;
  rows_init(&cur, 1);
  if (!sources) {
    if (rows_reserve(&cur, V)) return -3;
    for (uint64_t i = 0; i < V; i++) cur.data[i] = (int64_t)i;
    cur.n = V;
  } else {
    orc_rows pr;
    rows_init(&pr, 2);
    if ((rc = orc_jht_probe(&person, sources, n_src, &pr))) return rc;
    if (rows_reserve(&cur, pr.n)) return -3;
    for (uint64_t i = 0; i < pr.n; i++) cur.data[i] = pr.data[i * 2 + 1];
    cur.n = pr.n;
    orc_rows_free(&pr);
  }

  for (int h = 1; h <= k_max; h++) {
    /* probe knows (build side keyed on k_person1id) with the id of the path's last vertex */
    int pc = cur.ncols;
    int64_t *keys = (int64_t *)malloc((size_t)(cur.n ? cur.n : 1) * sizeof(int64_t));
    if (!keys) return -3;
    for (uint64_t r = 0; r < cur.n; r++) keys[r] = vid[cur.data[r * pc + (pc - 1)]];
    orc_rows pe;
    rows_init(&pe, 2);
    if ((rc = orc_jht_probe(&knows, keys, cur.n, &pe))) return rc;
    free(keys);
    /* join the edge's destination with person (k.dst = p.id) */
    int64_t *dkeys = (int64_t *)malloc((size_t)(pe.n ? pe.n : 1) * sizeof(int64_t));
    if (!dkeys) return -3;
    for (uint64_t r = 0; r < pe.n; r++) dkeys[r] = edst[pe.data[r * 2 + 1]];
    orc_rows pp;
    rows_init(&pp, 2);
    if ((rc = orc_jht_probe(&person, dkeys, pe.n, &pp))) return rc;
    free(dkeys);
    orc_rows nxt;
    rows_init(&nxt, pc + 1);
    if (rows_reserve(&nxt, pp.n)) return -3;
    for (uint64_t r = 0; r < pp.n; r++) {
      uint64_t pe_idx = (uint64_t)pp.data[r * 2];
      uint64_t row = (uint64_t)pe.data[pe_idx * 2];
      memcpy(&nxt.data[r * (pc + 1)], &cur.data[row * pc], (size_t)pc * sizeof(int64_t));
      nxt.data[r * (pc + 1) + pc] = pp.data[r * 2 + 1];
    }
    nxt.n = pp.n;
    orc_rows_free(&pe);
    orc_rows_free(&pp);
    orc_rows_free(&cur);
    cur = nxt;
    if (h >= k_min) {
      if (rows_reserve(&out[h], cur.n)) return -3;
      memcpy(out[h].data, cur.data, (size_t)cur.n * (size_t)(h + 1) * sizeof(int64_t));
      out[h].n = cur.n;
    }
  }
  orc_rows_free(&cur);
  orc_jht_free(&person);
  orc_jht_free(&knows);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Recursive CTE (UNION) + min(hop) GROUP BY                                                  */
/* ------------------------------------------------------------------------------------------ */
/* Tuple set with whole-tuple identity, standing in for GroupedAggregateHashTable::FindOrCreateGroups
 * (src/execution/aggregate_hashtable.cpp:367-504: linear probing; a tuple is "new" iff no equal
 * tuple is already present).  Only the new/not-new answer matters for the result relation. */
typedef struct tset {
  uint64_t cap, n;
  int64_t *t; /* cap*3 */
  uint8_t *used;
} tset;
static int tset_init(tset *s, uint64_t cap) {
  s->cap = next_pow2(cap < 1024 ? 1024 : cap);
  s->n = 0;
  s->t = (int64_t *)malloc((size_t)s->cap * 3 * sizeof(int64_t));
  s->used = (uint8_t *)calloc((size_t)s->cap, 1);
  return (s->t && s->used) ? 0 : -3;
}
static void tset_free(tset *s) {
  free(s->t);
  free(s->used);
}
static inline uint64_t tup_hash(int64_t a, int64_t b, int64_t c) {
  /* CombineHash-style mix (vector_hash.cpp:96-125 combines per-column hashes); any hash works */
  uint64_t h = ref_hash64(a);
  h = (h * 0xbf58476d1ce4e5b9ULL) ^ ref_hash64(b);
  h = (h * 0xbf58476d1ce4e5b9ULL) ^ ref_hash64(c);
  return h ^ (h >> 29);
}
static int tset_insert(tset *s, int64_t a, int64_t b, int64_t c, int *is_new);
static int tset_grow(tset *s) {
  tset n;
  if (tset_init(&n, s->cap * 2)) return -3;
  for (uint64_t i = 0; i < s->cap; i++)
    if (s->used[i]) {
      int nw;
      tset_insert(&n, s->t[i * 3], s->t[i * 3 + 1], s->t[i * 3 + 2], &nw);
    }
  tset_free(s);
  *s = n;
  return 0;
}
static int tset_insert(tset *s, int64_t a, int64_t b, int64_t c, int *is_new) {
  if ((s->n + 1) * 2 > s->cap)
    if (tset_grow(s)) return -3;
  uint64_t i = tup_hash(a, b, c) & (s->cap - 1);
  while (s->used[i]) {
    if (s->t[i * 3] == a && s->t[i * 3 + 1] == b && s->t[i * 3 + 2] == c) {
      *is_new = 0;
      return 0;
    }
    i = (i + 1) & (s->cap - 1);
  }
  s->used[i] = 1;
  s->t[i * 3] = a;
  s->t[i * 3 + 1] = b;
  s->t[i * 3 + 2] = c;
  s->n++;
  *is_new = 1;
  return 0;
}

/* PhysicalRecursiveCTE::{Sink,ProbeHT,GetData,ExecuteRecursivePipelines}
 *                                     src/execution/operator/set/physical_recursive_cte.cpp:48-139
 * Level loop: the working table (tuples new in the previous iteration) is joined with knows
 * (hash join probing on friend = k_person1id) under the filter hopCount < max_hops; produced
 * tuples (start, hop+1, k_person2id) pass the UNION dedupe on the WHOLE tuple (:48-58, :60-72);
 * recursion stops when an iteration adds nothing (:100-102).  The reference rebuilds the knows
 * hash table every level (:112-119); the table is identical each time, so it is built once here.
 * Then friends_shortest = min(hopCount) GROUP BY startPerson, friend
 * (bi-10-shortestpath.sql:26-31; PhysicalHashAggregate physical_hash_aggregate.cpp:152-266).
 * As in gg.h, the destination must be a vertex (k_person2id = p.p_personid). */
int orc_cte_shortest(const int64_t *vid, uint64_t V, const int64_t *esrc, const int64_t *edst, uint64_t E,
                     const int64_t *sources, uint64_t n_src, int max_hops, orc_rows *out) {
  if (max_hops < 0) return -1; /* the reference recursion does not terminate on cycles without the bound */
  rows_init(out, 3);
  orc_jht person, knows;
  int rc;
  if ((rc = orc_jht_build(&person, vid, V))) return rc;
  if ((rc = orc_jht_build(&knows, esrc, E))) return rc;
  tset seen;
  if (tset_init(&seen, 4096)) return -3;

  orc_rows all, work, inter;
  rows_init(&all, 3);
  rows_init(&work, 3);
  rows_init(&inter, 3);

  /* seed arm: SELECT p_personid, 0, p_personid FROM person WHERE p_personid IN (sources) */
  {
    orc_rows pr;
    rows_init(&pr, 2);
    if ((rc = orc_jht_probe(&person, sources, n_src, &pr))) return rc;
    for (uint64_t i = 0; i < pr.n; i++) {
      int64_t id = vid[pr.data[i * 2 + 1]];
      int nw;
      if (tset_insert(&seen, id, 0, id, &nw)) return -3;
      if (nw) {
        if (rows_reserve(&inter, 1)) return -3;
        int64_t *d = &inter.data[inter.n * 3];
        d[0] = id;
        d[1] = 0;
        d[2] = id;
        inter.n++;
      }
    }
    orc_rows_free(&pr);
  }

  while (inter.n) {
    /* emit intermediate_table, then working_table <- intermediate_table (:81-94) */
    if (rows_reserve(&all, inter.n)) return -3;
    memcpy(&all.data[all.n * 3], inter.data, (size_t)inter.n * 3 * sizeof(int64_t));
    all.n += inter.n;
    orc_rows_free(&work);
    work = inter;
    rows_init(&inter, 3);

    /* recursive arm: FILTER hopCount < max_hops -> HASH_JOIN(friend = k_person1id) -> person join */
    uint64_t nk = 0;
    int64_t *keys = (int64_t *)malloc((size_t)(work.n ? work.n : 1) * sizeof(int64_t));
    uint64_t *rowof = (uint64_t *)malloc((size_t)(work.n ? work.n : 1) * sizeof(uint64_t));
    if (!keys || !rowof) return -3;
    for (uint64_t r = 0; r < work.n; r++)
      if (work.data[r * 3 + 1] < max_hops) {
        keys[nk] = work.data[r * 3 + 2];
        rowof[nk++] = r;
      }
    orc_rows pe;
    rows_init(&pe, 2);
    if ((rc = orc_jht_probe(&knows, keys, nk, &pe))) return rc;
    int64_t *dkeys = (int64_t *)malloc((size_t)(pe.n ? pe.n : 1) * sizeof(int64_t));
    if (!dkeys) return -3;
    for (uint64_t r = 0; r < pe.n; r++) dkeys[r] = edst[pe.data[r * 2 + 1]];
    orc_rows pp;
    rows_init(&pp, 2);
    if ((rc = orc_jht_probe(&person, dkeys, pe.n, &pp))) return rc;
    for (uint64_t r = 0; r < pp.n; r++) {
      uint64_t pe_idx = (uint64_t)pp.data[r * 2];
      uint64_t wr = rowof[pe.data[pe_idx * 2]];
      int64_t start = work.data[wr * 3], hop = work.data[wr * 3 + 1] + 1, fr = dkeys[pe_idx];
      int nw;
      if (tset_insert(&seen, start, hop, fr, &nw)) return -3;
      if (nw) {
        if (rows_reserve(&inter, 1)) return -3;
        int64_t *d = &inter.data[inter.n * 3];
        d[0] = start;
        d[1] = hop;
        d[2] = fr;
        inter.n++;
      }
    }
    free(keys);
    free(rowof);
    free(dkeys);
    orc_rows_free(&pe);
    orc_rows_free(&pp);
  }
  orc_rows_free(&work);
  orc_rows_free(&inter);
  tset_free(&seen);

  /* min(hopCount) GROUP BY startPerson, friend: sort by (start, friend, hop), keep first of each group */
  for (uint64_t r = 0; r < all.n; r++) { /* reorder columns to (start, friend, hop) */
    int64_t hop = all.data[r * 3 + 1];
    all.data[r * 3 + 1] = all.data[r * 3 + 2];
    all.data[r * 3 + 2] = hop;
  }
  orc_rows_sort(&all);
  if (rows_reserve(out, all.n)) return -3;
  for (uint64_t r = 0; r < all.n; r++) {
    if (r && all.data[r * 3] == all.data[(r - 1) * 3] && all.data[r * 3 + 1] == all.data[(r - 1) * 3 + 1]) continue;
    memcpy(&out->data[out->n * 3], &all.data[r * 3], 3 * sizeof(int64_t));
    out->n++;
  }
  orc_rows_free(&all);
  orc_jht_free(&person);
  orc_jht_free(&knows);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Direct CSR formulation                                                                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct idpair {
  int64_t id;
  int64_t idx;
} idpair;
static int idpair_cmp(const void *a, const void *b) {
  const idpair *x = (const idpair *)a, *y = (const idpair *)b;
  if (x->id < y->id) return -1;
  if (x->id > y->id) return 1;
  return (x->idx < y->idx) ? -1 : (x->idx > y->idx);
}
/* the sorted (id, idx) table's pointer is stored in the spare slot vid[V] */
static idpair *g_sorted_of(const orc_csr *g) { return *(idpair **)(g->vid + g->V); }

int64_t orc_csr_lookup(const orc_csr *g, int64_t id) {
  const idpair *s = g_sorted_of(g);
  uint64_t lo = 0, hi = g->V;
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (s[mid].id < id)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (lo < g->V && s[lo].id == id) ? s[lo].idx : -1;
}

/* The adjacency index the reference builds is the join hash table keyed on the source vertex
 * (PhysicalHashJoin::Sink/Finalize, physical_hash_join.cpp:128-185); the CSR is the same
 * information laid out densely: row u lists, in edge-rowid order, the dense index of every
 * k_person2id with k_person1id = vid[u] whose two endpoints are both vertices. */
int orc_csr_build(orc_csr *g, const int64_t *vid, uint64_t V, const int64_t *esrc, const int64_t *edst,
                  const int64_t *rowid, uint64_t E) {
  memset(g, 0, sizeof(*g));
  g->V = V;
  /* vid array carries one extra slot that stores the pointer to the sorted (id, idx) table */
  g->vid = (int64_t *)malloc((size_t)(V + 2) * sizeof(int64_t));
  idpair *sorted = (idpair *)malloc((size_t)(V ? V : 1) * sizeof(idpair));
  if (!g->vid || !sorted) return -3;
  memcpy(g->vid, vid, (size_t)V * sizeof(int64_t));
  *(idpair **)(g->vid + V) = sorted;
  for (uint64_t i = 0; i < V; i++) {
    sorted[i].id = vid[i];
    sorted[i].idx = (int64_t)i;
  }
  qsort(sorted, (size_t)V, sizeof(idpair), idpair_cmp);
  for (uint64_t i = 1; i < V; i++)
    if (sorted[i].id == sorted[i - 1].id) return -4;

  uint32_t *su = (uint32_t *)malloc((size_t)(E ? E : 1) * sizeof(uint32_t));
  uint32_t *dv = (uint32_t *)malloc((size_t)(E ? E : 1) * sizeof(uint32_t));
  g->off = (int64_t *)calloc((size_t)V + 2, sizeof(int64_t));
  if (!su || !dv || !g->off) return -3;
  /* The lookups and the counting sort run on up to 32 threads, each owning a contiguous chunk of the edge rows;
   * per-thread degree counts turn into per-thread start positions, so the placement is the same stable one
   * (ascending edge position within a row) whatever the thread count. */
  int T = 1;
#ifdef _OPENMP
  T = omp_get_max_threads();
  if (T > 32) T = 32;
  if (T < 1) T = 1;
  if (E < 100000) T = 1;
#endif
  int64_t *h = (int64_t *)calloc((size_t)T * (size_t)(V + 1), sizeof(int64_t));
  if (!h) return -3;
  uint64_t kept = 0;
#pragma omp parallel num_threads(T) reduction(+ : kept)
  {
    int t = 0;
#ifdef _OPENMP
    t = omp_get_thread_num();
#endif
    const uint64_t e0 = E * (uint64_t)t / (uint64_t)T, e1 = E * (uint64_t)(t + 1) / (uint64_t)T;
    int64_t *ht = h + (size_t)t * (size_t)(V + 1);
    for (uint64_t e = e0; e < e1; e++) {
      int64_t u = orc_csr_lookup(g, esrc[e]), v = orc_csr_lookup(g, edst[e]);
      if (u < 0 || v < 0) {
        su[e] = 0xFFFFFFFFu;
        continue;
      }
      su[e] = (uint32_t)u;
      dv[e] = (uint32_t)v;
      ht[u]++;
      kept++;
    }
  }
  g->E = kept;
  g->dropped = E - kept;
  for (uint64_t u = 0; u < V; u++) {
    int64_t d = 0;
    for (int t = 0; t < T; t++) d += h[(size_t)t * (size_t)(V + 1) + u];
    g->off[u + 1] = g->off[u] + d;
  }
  g->nbr = (uint32_t *)malloc((size_t)(kept ? kept : 1) * sizeof(uint32_t));
  g->eid = (int64_t *)malloc((size_t)(kept ? kept : 1) * sizeof(int64_t));
  if (!g->nbr || !g->eid) return -3;
#pragma omp parallel for num_threads(T) schedule(static)
  for (uint64_t u = 0; u < V; u++) { /* per-thread counts -> per-thread first positions */
    int64_t run = g->off[u];
    for (int t = 0; t < T; t++) {
      int64_t c = h[(size_t)t * (size_t)(V + 1) + u];
      h[(size_t)t * (size_t)(V + 1) + u] = run;
      run += c;
    }
  }
#pragma omp parallel num_threads(T)
  {
    int t = 0;
#ifdef _OPENMP
    t = omp_get_thread_num();
#endif
    const uint64_t e0 = E * (uint64_t)t / (uint64_t)T, e1 = E * (uint64_t)(t + 1) / (uint64_t)T;
    int64_t *cur = h + (size_t)t * (size_t)(V + 1);
    for (uint64_t e = e0; e < e1; e++) { /* stable: ascending edge position within a row */
      if (su[e] == 0xFFFFFFFFu) continue;
      int64_t p = cur[su[e]]++;
      g->nbr[p] = dv[e];
      g->eid[p] = rowid ? rowid[e] : (int64_t)e;
    }
  }
  free(h);
  free(su);
  free(dv);
  return 0;
}
void orc_csr_free(orc_csr *g) {
  if (!g) return;
  if (g->vid) free(*(idpair **)(g->vid + g->V));
  free(g->vid);
  free(g->off);
  free(g->nbr);
  free(g->eid);
  memset(g, 0, sizeof(*g));
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Depth-first enumeration of the walks the join chain produces: the children of a path prefix
 * ending in v are exactly the probe matches of vid[v] in the knows hash table = CSR row v. */
static void khop_rec(const orc_csr *g, uint32_t v, uint64_t q, int j, int k_min, int k_max, orc_khop_stats *st) {
  const int64_t b = g->off[v], e = g->off[v + 1];
  const int h = j + 1;
  st->traversed_edges += (uint64_t)(e - b);
  st->frontier_entries += 1;
  if (h >= k_min) st->rows[h] += (uint64_t)(e - b);
  if (h == k_max) {
    uint32_t slo = 0;
    const uint32_t *nb = g->nbr;
    for (int64_t i = b; i < e; i++) slo += (uint32_t)orc_leaf(q, nb[i]);
    st->digest[h] = orc_dsum_add(st->digest[h], slo);
    return;
  }
  for (int64_t i = b; i < e; i++) {
    uint32_t w = g->nbr[i];
    uint64_t p = orc_leaf(q, w);
    if (h >= k_min) st->digest[h] = orc_dsum_add(st->digest[h], p);
    khop_rec(g, w, orc_q(p, h), h, k_min, k_max, st);
  }
}

int orc_khop_csr(const orc_csr *g, const uint32_t *src_dense, uint64_t n_src, uint64_t lo, uint64_t hi, int k_min,
                 int k_max, int threads, orc_khop_stats *st) {
  if (k_min < 1 || k_max < k_min || k_max > ORC_MAX_HOPS) return -1;
  memset(st, 0, sizeof(*st));
  uint64_t n = src_dense ? n_src : (hi > lo ? hi - lo : 0);
#ifdef _OPENMP
  int nt = threads > 0 ? threads : omp_get_max_threads();
#else
  int nt = 1;
  (void)threads;
#endif
  orc_khop_stats *part = (orc_khop_stats *)calloc((size_t)nt, sizeof(orc_khop_stats));
  if (!part) return -3;
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
  {
#ifdef _OPENMP
    int t = omp_get_thread_num();
#else
    int t = 0;
#endif
    orc_khop_stats *ps = &part[t];
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
    for (uint64_t i = 0; i < n; i++) {
      uint32_t u = src_dense ? src_dense[i] : (uint32_t)(lo + i);
      if (u >= g->V) continue;
      khop_rec(g, u, orc_q((uint64_t)u, 0), 0, k_min, k_max, ps);
    }
  }
  for (int t = 0; t < nt; t++) {
    for (int h = 0; h <= ORC_MAX_HOPS; h++) {
      st->rows[h] += part[t].rows[h];
      st->digest[h] = orc_dsum_add(st->digest[h], part[t].digest[h]);
    }
    st->traversed_edges += part[t].traversed_edges;
    st->frontier_entries += part[t].frontier_entries;
  }
  free(part);
  return 0;
}

static int khop_rows_rec(const orc_csr *g, uint32_t *path, int j, int k_min, int k_max, orc_rows *out) {
  uint32_t v = path[j];
  int h = j + 1;
  for (int64_t i = g->off[v]; i < g->off[v + 1]; i++) {
    path[h] = g->nbr[i];
    if (h >= k_min) {
      if (rows_reserve(&out[h], 1)) return -3;
      int64_t *d = &out[h].data[out[h].n * (uint64_t)(h + 1)];
      for (int c = 0; c <= h; c++) d[c] = g->vid[path[c]];
      out[h].n++;
    }
    if (h < k_max) {
      int rc = khop_rows_rec(g, path, h, k_min, k_max, out);
      if (rc) return rc;
    }
  }
  return 0;
}
int orc_khop_csr_rows(const orc_csr *g, const uint32_t *src_dense, uint64_t n_src, uint64_t lo, uint64_t hi,
                      int k_min, int k_max, orc_rows *out) {
  if (k_min < 1 || k_max < k_min || k_max > ORC_MAX_HOPS) return -1;
  for (int h = 0; h <= ORC_MAX_HOPS; h++) rows_init(&out[h], h + 1);
  uint64_t n = src_dense ? n_src : (hi > lo ? hi - lo : 0);
  uint32_t path[ORC_MAX_HOPS + 1];
  for (uint64_t i = 0; i < n; i++) {
    uint32_t u = src_dense ? src_dense[i] : (uint32_t)(lo + i);
    if (u >= g->V) continue;
    path[0] = u;
    int rc = khop_rows_rec(g, path, 0, k_min, k_max, out);
    if (rc) return rc;
  }
  return 0;
}

/* 64-lane bitset BFS: lane i's bit travels along every edge once per level; the first level at
 * which bit i reaches v is min(hopCount) of (start_i, v) in the recursive-CTE relation above. */
int orc_bfs64_csr(const orc_csr *g, const int64_t *src_dense, int n_src, int max_hops, int32_t *dist,
                  orc_bfs_stats *st) {
  if (n_src < 0 || n_src > 64) return -1;
  const uint64_t V = g->V;
  uint64_t *front = (uint64_t *)calloc((size_t)(V ? V : 1), 8);
  uint64_t *seen = (uint64_t *)calloc((size_t)(V ? V : 1), 8);
  uint64_t *next = (uint64_t *)calloc((size_t)(V ? V : 1), 8);
  if (!front || !seen || !next) return -3;
  memset(st, 0, sizeof(*st));
  for (uint64_t i = 0; i < (uint64_t)n_src * V; i++) dist[i] = -1;
  int any = 0;
  for (int i = 0; i < n_src; i++) {
    if (src_dense[i] < 0 || (uint64_t)src_dense[i] >= V) continue;
    front[src_dense[i]] |= 1ULL << i;
    seen[src_dense[i]] |= 1ULL << i;
    dist[(uint64_t)i * V + (uint64_t)src_dense[i]] = 0;
    st->reached_pairs++;
    any = 1;
  }
  int level = 0;
  while (any && (max_hops < 0 || level < max_hops)) {
    st->levels++;
    for (uint64_t v = 0; v < V; v++) {
      uint64_t f = front[v];
      if (!f) continue;
      st->active_vertices++;
      st->traversed_edges += (uint64_t)(g->off[v + 1] - g->off[v]);
      for (int64_t i = g->off[v]; i < g->off[v + 1]; i++) next[g->nbr[i]] |= f;
    }
    level++;
    any = 0;
    for (uint64_t v = 0; v < V; v++) {
      uint64_t nw = next[v] & ~seen[v];
      next[v] = 0;
      front[v] = nw;
      if (!nw) continue;
      any = 1;
      seen[v] |= nw;
      while (nw) {
        int b = __builtin_ctzll(nw);
        nw &= nw - 1;
        dist[(uint64_t)b * V + v] = level;
        st->reached_pairs++;
      }
    }
  }
  free(front);
  free(seen);
  free(next);
  return 0;
}
