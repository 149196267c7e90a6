#!/usr/bin/env python3
"""All-sources 1..3-hop count + digest: the product kernels against the frontier kernels (diagnostic).
usage: bench_k3.py sf1|sf10 [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402

scale = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
vid, src, dst = pkg.datagen.ldbc(scale)
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_vertices(vid)
g.append_edges(src, dst)
c = g.build_csr()
out = {"workload": scale, "V": int(vid.size), "E": int(src.size)}
res = {}
for name, frontier in (("product", False), ("frontier", True)):
    g.force_frontier(frontier)
    st = g.expand_khop(c, 1, 3)  # warm-up
    g.profile_reset()
    g.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        st = g.expand_khop(c, 1, 3)
    dt = (time.perf_counter() - t0) / reps
    g.profile(False)
    res[name] = st
    out[name] = {"ms": round(dt * 1e3, 3), "walks_3hop": st["rows"][3], "traversed_edges": st["traversed_edges"],
                 "edges_per_s": st["traversed_edges"] / dt,
                 "kernels_us": {k: round(v[1] / reps * 1e3, 1) for k, v in g.profile_get().items() if v[0]}}
g.force_frontier(False)
out["equal"] = res["product"] == res["frontier"]
out["speedup"] = out["frontier"]["ms"] / out["product"]["ms"]
print(json.dumps(out))
