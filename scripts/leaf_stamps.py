#!/usr/bin/env python3
"""Diagnostic: where a leaf of k_leaf_rows spends its time (needs a -DGG_FB_LEAF_STAMPS=1 library).
usage: GG_LEAF_STAMPS_FILE=/tmp/stamps.bin leaf_stamps.py build/libgg_stamps.so [scale]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd import gg as ggmod  # noqa: E402

ggmod._lib = ggmod.load_library(os.path.abspath(sys.argv[1]))
vid, src, dst = pkg.datagen.ldbc(sys.argv[2] if len(sys.argv) > 2 else "sf100")
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_vertices(vid)
g.append_edges(src, dst)
for _ in range(3):  # the first build allocates the stamp buffer, the later ones fill and dump it
    g.build_csr().close()
st = np.fromfile(os.environ["GG_LEAF_STAMPS_FILE"], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
st = st[st[:, 0] != 0]
names = ["header->run table", "run table->entries", "entries->counted", "counted->run table (pass 1)",
         "run table->entries (pass 1)", "entries->ranked", "ranked->rows written"]
d = np.diff(st, axis=1)
life = st[:, 7] - st[:, 0]
print("leaves", st.shape[0], "mean life (cycles)", life.mean(), "median", np.median(life), "p90", np.percentile(life, 90),
      "max", life.max())
for k, nm in enumerate(names):
    x = d[:, k]
    print(f"{nm:32s} mean {x.mean():9.0f}  median {np.median(x):9.0f}  p90 {np.percentile(x, 90):9.0f}")
span = st[:, 7].max() - st[:, 0].min()
print("kernel span (cycles)", span, "sum of lives / span = average resident leaves", life.sum() / span)
