#!/usr/bin/env python3
"""More than 2^30 edge rows through the EDGE-ONLY build (gg_vertices_from_edges: the pair-probed endpoint set filled in
one pass + bucket sort of the ids, then gg_csr_build): the derived vertex table against the ids that were drawn, and the
arrays and the 2-hop count against the general path (CAS set + LSD rounds, then the multi-pass build) on the same
staged rows.  Diagnostic; ~45 GB of host memory.     usage: check_big_e_edge_only.py [V] [E]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1_150_000_000
rng = np.random.default_rng(9)
vid = np.arange(V, dtype=np.int64) * np.int64(4398046511) - np.int64(1 << 50)  # sparse ids on both sides of zero
vid = vid[rng.permutation(V)]
g = pkg.GG(0)
g.set_edge_rowid(False)
seen = np.zeros(V, bool)
t0 = time.perf_counter()
CH = 100_000_000
for a in range(0, E, CH):
    n = min(CH, E - a)
    s = np.minimum((rng.pareto(1.5, n) * (V / 200)).astype(np.int64), V - 1)  # skewed sources, uniform destinations
    d = rng.integers(0, V, n, dtype=np.int64)
    seen[s] = True
    seen[d] = True
    g.append_edges(vid[s], vid[d])
    print(f"staged {a + n} rows, {time.perf_counter() - t0:.0f} s", flush=True)
want = np.sort(vid[seen])
out = {"V_drawn": V, "V_seen": int(want.size), "E": E}
res = {}
for name, legacy in (("fast", False), ("general", True)):
    g.force_legacy_build(legacy)
    t = time.perf_counter()
    n = g.vertices_from_edges()
    c = g.build_csr()
    out[name + "_s"] = round(time.perf_counter() - t, 3)
    st = g.expand_khop(c, 1, 2)
    off, nbr, _, v2 = c.export()
    res[name] = (st, off, nbr, v2, c.E, n)
    c.close()
    print(name, "built and exported", flush=True)
g.force_legacy_build(False)
a, b = res["fast"], res["general"]
out["edges_kept"] = int(a[4])
for name in ("fast", "general"):
    r = res[name]
    out[name] = {"vertices": int(r[5]), "vertex_table_ok": bool(np.array_equal(r[3], want)), "E": int(r[4]),
                 "offsets_last": int(r[1][-1]), "rows": [int(x) for x in r[0]["rows"][:3]],
                 "degree_sum_ok": bool(int(np.diff(r[1]).sum()) == E)}
out["vertex_count_ok"] = bool(a[5] == b[5] == want.size)
out["vertex_table_is_the_sorted_distinct_ids"] = bool(np.array_equal(a[3], want) and np.array_equal(b[3], want))
out["offsets_equal"] = bool(np.array_equal(a[1], b[1]))
out["neighbours_equal"] = bool(np.array_equal(a[2], b[2]))
out["khop_equal"] = a[0] == b[0]
out["rows_2hop"] = int(a[0]["rows"][2])
out["row_offsets_sum_to_E"] = bool(int(a[1][-1]) == E)
print(json.dumps(out))
ok = all(out[k] for k in ("vertex_count_ok", "vertex_table_is_the_sorted_distinct_ids", "offsets_equal", "neighbours_equal",
                          "khop_equal", "row_offsets_sum_to_E"))
sys.exit(0 if ok else 1)
