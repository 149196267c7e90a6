#!/usr/bin/env python3
"""Diagnostic: counts from an EXPLICIT frontier — every vertex passed as a source LIST — 1..2 and 1..3 hops: the
product form of the last hop (k_expand_pairs + sort + k_expand_front) against the frontier kernels (knob 1) and, for
reference, the all-sources product kernels (source list omitted).   usage: bench_front.py sf1|sf10 [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import torch  # noqa: E402

import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd.gg import KhopStats  # noqa: E402

torch.cuda.init()

scale = sys.argv[1] if len(sys.argv) > 1 else "sf10"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
vid, src, dst = pkg.datagen.ldbc(scale)
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_vertices(vid)
g.append_edges(src, dst)
c = g.build_csr()
out = {"workload": scale, "V": int(vid.size), "E": int(src.size)}
for kmax in (2, 3):
    res = {}
    for name, knob, srcs in (("list_product", 0, vid), ("list_pairs", 2, vid), ("list_frontier", 1, vid),
                             ("all_sources_product", 0, None)):
        g.force_frontier(knob)
        st = g.expand_khop(c, 1, kmax, sources=srcs)
        g.profile_reset()
        g.profile_select(None)
        g.profile(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            st = g.expand_khop(c, 1, kmax, sources=srcs)
        dt = (time.perf_counter() - t0) / reps
        g.profile(False)
        res[name] = (st["rows"], st["digest"])
        out[f"k{kmax}_{name}"] = {"ms": round(dt * 1e3, 3), "walks": st["rows"][kmax],
                                  "kernels_us": {k: round(v[1] / reps * 1e3, 1) for k, v in g.profile_get().items() if v[1] > 0.005}}
    g.force_frontier(0)
    out[f"k{kmax}_equal"] = res["list_product"] == res["list_pairs"] == res["list_frontier"] == res["all_sources_product"]
# the same source list, 2-hop rows MATERIALISED: k_mat_front (level 1 sorted by last vertex) against k_mat_last
for name, knob in (("mat_list_product", 0), ("mat_list_frontier", 1)):
    g.force_frontier(knob)
    best = None
    for rep in range(3):
        g.profile_reset()
        g.profile_select(None)
        g.profile(True)
        st, res = KhopStats(), C.c_void_p()
        a = vid.ctypes.data_as(C.POINTER(C.c_int64))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g._chk(g.lib.gg_expand_khop_result(g.ctx, c.handle, a, vid.size, 2, 2, C.byref(st), C.byref(res)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g.profile(False)
        g.lib.gg_result_destroy(res)
        best = dt if best is None or dt < best else best
    prof = {k: round(v[1] * 1e3, 1) for k, v in g.profile_get().items() if v[1] > 0.005}
    kname = "mat_front" if "mat_front" in prof else "mat_last"
    out[name] = {"ms": round(best * 1e3, 3), "rows": st.rows[2], "kernels_us": prof,
                 "store_TBps_of_" + kname: st.rows[2] * 24 / (prof[kname] * 1e-6) / 1e12}
g.force_frontier(0)
print(json.dumps(out))
