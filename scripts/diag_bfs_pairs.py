#!/usr/bin/env python3
"""Diagnostic: gg_bfs64_pairs_packed on SF100 (64 seeds, hopCount < 5: what the shortest-path statement's scan does when
it opens) — wall time of the call without the fetch, and its kernels.  usage: diag_bfs_pairs.py [sf100] [max_hops]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd.gg import BfsStats, _i64  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf100"
hops = int(sys.argv[2]) if len(sys.argv) > 2 else 5
vid, src, dst = pkg.datagen.ldbc(scale)
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_edges(src, dst)
g.vertices_from_edges()
c = g.build_csr()
sources = pkg.datagen.pick_sources(vid, 64, 1)
s, ps = _i64(sources)
for rep in range(5):
    if rep == 3:
        g.profile_reset()
        g.profile_select(None)
        g.profile(True)
    st, res = BfsStats(), C.c_void_p()
    t = time.perf_counter()
    g._chk(g.lib.gg_bfs64_pairs_packed(g.ctx, c.handle, ps, s.size, hops, C.byref(st), C.byref(res)))
    dt = time.perf_counter() - t
    n = C.c_uint64()
    g._chk(g.lib.gg_result_rows(res, 0, C.byref(n)))
    g.lib.gg_result_destroy(res)
    print(f"rep {rep}: call {dt * 1e3:.3f} ms, {n.value} pairs, levels {st.levels}", flush=True)
g.profile(False)
for k, v in sorted(g.profile_get().items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} {v[0]:4d} launches {v[1] / 2:8.3f} ms per call")
c.close()
g.close()
