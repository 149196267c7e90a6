#!/usr/bin/env python3
"""Many vertices: V = 80 M (64 x V past 2^32: the BFS words and distance matrix), E = 400 M rows — CSR arrays, 2-hop
count + digest and a 64-source BFS to fixpoint against the CPU oracle, and the edge-only build's vertex table against
the ids drawn.  Diagnostic; ~80 GB of host memory.     usage: check_big_v.py [V] [E]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402
from tests import oracle_lib  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000_000
rng = np.random.default_rng(17)
t0 = time.perf_counter()
vid = (np.arange(V, dtype=np.int64) * np.int64(137438953) - np.int64(1 << 52))  # sparse, both signs, sorted
s = rng.integers(0, V, E, dtype=np.int64)
d = rng.integers(0, V, E, dtype=np.int64)
src, dst = vid[s], vid[d]
print(f"tables drawn, {time.perf_counter() - t0:.0f} s", flush=True)
out = {"V": V, "E": E}
orc = oracle_lib.load()
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_vertices(vid)
g.append_edges(src, dst)
t = time.perf_counter()
c = g.build_csr()
out["build_s"] = round(time.perf_counter() - t, 3)
t = time.perf_counter()
rc, og = orc.csr_build(vid, src, dst, None)
assert rc == 0
out["oracle_build_s"] = round(time.perf_counter() - t, 1)
print("built", out, flush=True)
off, nbr, _, v2 = c.export()
o_off, o_nbr, _, o_vid = og.arrays()
out["offsets_equal"] = bool(np.array_equal(off, o_off))
out["neighbours_equal"] = bool(np.array_equal(nbr, o_nbr))
out["vertex_ids_equal"] = bool(np.array_equal(v2, o_vid))
del off, nbr, o_off, o_nbr
st = g.expand_khop(c, 1, 2)
ref = og.khop(1, 2)
out["khop_equal"] = st == ref
out["rows_2hop"] = int(st["rows"][2])
print("khop", out, flush=True)
sources = vid[rng.integers(0, V, 64)]
t = time.perf_counter()
dist, bst = g.bfs64(c, sources, -1)
out["bfs_s"] = round(time.perf_counter() - t, 2)
dense = np.searchsorted(vid, sources).astype(np.int64)
o_dist, o_st = og.bfs64(dense, -1)
out["bfs_distances_equal"] = bool(np.array_equal(dist, o_dist))
out["bfs_stats_equal"] = bst == o_st
out["bfs_levels"] = int(bst["levels"])
out["bfs_reached_pairs"] = int(bst["reached_pairs"])
del dist, o_dist
c.close()
og.close()
print("bfs", out, flush=True)
# edge-only: the vertex table is the set of ids that occur
seen = np.zeros(V, bool)
seen[s] = True
seen[d] = True
want = vid[seen]
g.staging_clear()
g.append_edges(src, dst)
n = g.vertices_from_edges()
c = g.build_csr()
_, _, _, v3 = c.export()
out["edge_only_vertices"] = int(n)
out["edge_only_vertex_table_ok"] = bool(n == want.size and np.array_equal(v3, want))
out["edge_only_rows_equal"] = g.expand_khop(c, 1, 2)["rows"] == st["rows"]
c.close()
g.close()
print(json.dumps(out))
ok = all(out[k] for k in ("offsets_equal", "neighbours_equal", "vertex_ids_equal", "khop_equal", "bfs_distances_equal",
                          "bfs_stats_equal", "edge_only_vertex_table_ok", "edge_only_rows_equal"))
sys.exit(0 if ok else 1)
