#!/usr/bin/env python3
"""N CSR builds (+ one 2-hop expansion each) of a workload with a given libgg: the program rocprofv3 profiles.
usage: build_once.py sf100 [lib.so] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd import gg as ggmod  # noqa: E402

scale = sys.argv[1]
if len(sys.argv) > 2 and sys.argv[2] != "-":
    ggmod._lib = ggmod.load_library(os.path.abspath(sys.argv[2]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
vid, src, dst = pkg.datagen.ldbc(scale)
g = pkg.GG(0)
g.set_edge_rowid(os.environ.get("AB_ROWID", "0") == "1")
g.append_vertices(vid)
g.append_edges(src, dst)
for _ in range(n):
    c = g.build_csr()
    g.expand_khop(c, 1, 2)
    c.close()
g.close()
