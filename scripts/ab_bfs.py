#!/usr/bin/env python3
"""A/B timing of libgg variants: 64-source BFS batches at a scale (diagnostic).
usage: ab_bfs.py sf100 libA.so libB.so ..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd import gg as ggmod  # noqa: E402

scale = sys.argv[1]
vid, src, dst = pkg.datagen.ldbc(scale)
batches = [pkg.datagen.pick_sources(vid, 64, 100 + b) for b in range(8)]
ref = None
for rep in range(2):
    for lib in sys.argv[2:]:
        ggmod._lib = ggmod.load_library(os.path.abspath(lib))
        g = pkg.GG(0)
        g.set_edge_rowid(False)
        g.append_vertices(vid)
        g.append_edges(src, dst)
        c = g.build_csr()
        g.bfs64(c, batches[0], -1, fetch=False)  # warm-up
        g.profile_reset()
        g.profile(True)
        t0 = time.perf_counter()
        out = []
        for b in batches:
            out.append(g.bfs64(c, b, -1, fetch=False)[1])
        dt = (time.perf_counter() - t0) / len(batches)
        g.profile(False)
        if ref is None:
            ref = out
        print(lib, json.dumps({"ok": out == ref, "ms_per_batch": round(dt * 1e3, 4),
                               **{k: round(v[1] / len(batches) * 1e3, 1) for k, v in g.profile_get().items() if v[0]}}))
        c.close()
        g.close()
