"""Generate golden vectors by running the COMPILED REFERENCE (oracle/_ref/libduckdb.so, built from
/root/reference by oracle/Makefile `ref`) on seeded synthetic LDBC-shaped tables.

    python tests/golden/make_golden.py          # rewrites tests/golden/ldbc_*.npz

The reference cannot travel to the GPU box, so the vectors are committed: each .npz holds the INPUT
tables (person ids, knows src/dst) and the EXPECTED OUTPUT relations of the reference's SQL
formulations of the hot path (oracle/ref_duckdb.py: sql_khop_rows, sql_khop, sql_shortest — the
1-/2-hop join chains after benchmark/ldbc/queries/interactive-complex-3.sql:11 and the
friends/friends_shortest CTE pair of benchmark/ldbc/queries/bi-10-shortestpath.sql:8-31).
Outputs are stored sorted (hash-join output order is not a stable property of the reference).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from duckdb_pgq_amd import datagen  # noqa: E402
from oracle import ref_duckdb as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sort_rows(a):
    return a[np.lexsort(a.T[::-1])] if a.shape[0] else a


def make(name, vid, src, dst, hops_rows, hops_count, bfs):
    db = R.RefDuckDB(threads=4)
    db.load_ldbc(vid, src, dst)
    out = {"vid": vid, "src": src, "dst": dst}
    for h in hops_rows:
        out[f"rows{h}"] = sort_rows(db.execute(R.sql_khop_rows(h)))
    for h in hops_count:
        out[f"count{h}"] = db.execute(R.sql_khop(h))[0]
    for i, (n_src, seed, max_hops) in enumerate(bfs):
        sources = datagen.pick_sources(vid, n_src, seed)
        sources = np.concatenate([sources[:-1], np.array([-424242], np.int64)])  # one id that is not a person
        out[f"bfs{i}_sources"] = sources
        out[f"bfs{i}_max_hops"] = np.array([max_hops], np.int64)
        out[f"bfs{i}_rel"] = sort_rows(db.execute(R.sql_shortest(sources, max_hops)))
    db.close()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    # tiny multigraph with self-loops, duplicate edge rows and dangling endpoints
    vid, src, dst = datagen.small_graph(40, 160, 1234, dangling=6, dup_edges=12)
    make("ldbc_tiny", vid, src, dst, hops_rows=[1, 2, 3], hops_count=[1, 2, 3, 4], bfs=[(9, 1, 3), (9, 2, 6)])
    # LDBC-shaped (power-law, mirrored) small graph
    vid, src, dst = datagen.ldbc_knows(400, 4000, 0x5EED)
    make("ldbc_small", vid, src, dst, hops_rows=[1, 2], hops_count=[1, 2, 3], bfs=[(64, 3, 2), (64, 4, 5)])
    # SF0.1-sized: counts and BFS relation only
    vid, src, dst = datagen.ldbc("sf0.1")
    make("ldbc_sf0_1", vid, src, dst, hops_rows=[], hops_count=[1, 2], bfs=[(64, 5, 3)])
