#!/usr/bin/env python3
"""More than 2^30 edge rows through the bucketed build (byte offsets past 4 GB, u32 positions past 2^30): its arrays
and 2-hop result against the multi-pass build's on the same staged tables (diagnostic; ~40 GB of host memory).
usage: check_big_e.py [V] [E]"""
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
rng = np.random.default_rng(5)
vid = np.arange(V, dtype=np.int64) * np.int64(4398046511) + 11  # sparse ids: packed dictionary
vid = vid[rng.permutation(V)]
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_vertices(vid)
t0 = time.perf_counter()
CH = 100_000_000
for a in range(0, E, CH):
    n = min(CH, E - a)
    # skewed sources (a few thousand hubs), uniform destinations
    s = np.minimum((rng.pareto(1.5, n) * (V / 200)).astype(np.int64), V - 1)
    d = rng.integers(0, V, n, dtype=np.int64)
    g.append_edges(vid[s], vid[d])
    print(f"staged {a + n} rows, {time.perf_counter() - t0:.0f} s", flush=True)
out = {"V": V, "E": E}
res = {}
for name, legacy in (("bucketed", False), ("multipass", True)):
    g.force_legacy_build(legacy)
    t = time.perf_counter()
    c = g.build_csr()
    out[name + "_build_s"] = round(time.perf_counter() - t, 3)
    st = g.expand_khop(c, 1, 2)
    off, nbr, _, v2 = c.export()
    res[name] = (st, off, nbr, v2, c.E)
    c.close()
    print(name, "built and exported", flush=True)
g.force_legacy_build(False)
a, b = res["bucketed"], res["multipass"]
out["edges_kept"] = int(a[4])
for name in ("bucketed", "multipass"):
    r = res[name]
    out[name] = {"E": int(r[4]), "offsets_last": int(r[1][-1]), "rows": [int(x) for x in r[0]["rows"][:3]]}
out["offsets_equal"] = bool(np.array_equal(a[1], b[1]))
out["neighbours_equal"] = bool(np.array_equal(a[2], b[2]))
out["vertex_ids_equal"] = bool(np.array_equal(a[3], b[3]))
out["khop_equal"] = a[0] == b[0]
out["rows_2hop"] = int(a[0]["rows"][2])
print(json.dumps(out))
sys.exit(0 if out["offsets_equal"] and out["neighbours_equal"] and out["khop_equal"] else 1)
