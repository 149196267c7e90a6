#!/usr/bin/env python3
"""A/B timing of libgg variants on one box (diagnostic; variants may compute wrong results).
usage: ab_expand.py sf100 libA.so libB.so ...   -> per-variant expand_mid2 / build kernel times"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd import gg as ggmod  # noqa: E402

scale = sys.argv[1]
libs = sys.argv[2:]
vid, src, dst = pkg.datagen.ldbc(scale)
out = {}
for rep in range(2):
    for lib in libs:
        ggmod._lib = ggmod.load_library(os.path.abspath(lib))
        g = pkg.GG(0)
        g.append_vertices(vid)
        g.append_edges(src, dst)
        c = g.build_csr()
        g.expand_khop(c, 1, 2)
        g.profile_reset()
        g.profile(True)
        for _ in range(10):
            c2 = g.build_csr()
            g.expand_khop(c2, 1, 2)
            c2.close()
        g.profile(False)
        prof = g.profile_get()
        out.setdefault(lib, []).append({k: round(v[1] / v[0] * 1e3, 1) for k, v in prof.items() if v[0] and v[1] / v[0] > 0.05})
        c.close()
        g.close()
for lib, runs in out.items():
    print(lib)
    for r in runs:
        print("   ", json.dumps(r))
