#!/usr/bin/env python3
"""Diagnostic: the build the reference's own 2-hop SQL takes (k1.k_person2id = k2.k_person1id: no vertex table in the
pattern) — gg_vertices_from_edges + gg_csr_build over the staged edge table, per-kernel times from the library's events.
    python3 scripts/bench_edge_only.py [scale] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import duckdb_pgq_amd as pkg  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf100"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
vid, src, dst = pkg.datagen.ldbc(scale)
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_edges(src, dst)
g.staging_sync()


def step():
    n = g.vertices_from_edges()
    c = g.build_csr()
    return n, c


for _ in range(3):
    n, c = step()
    c.close()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    n, c = step()
    c.close()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps
g.profile_reset()
g.profile_select(None)
g.profile(True)
for _ in range(3):
    n, c = step()
    c.close()
g.profile(False)
prof = g.profile_get()
n, c = step()
E, V = src.size, n
st = g.expand_khop(c, 2, 2)
c.close()
set_us = sum(v[1] for k, v in prof.items() if k.startswith("set_")) / 3 * 1e3
all_us = sum(v[1] for v in prof.values()) / 3 * 1e3
alg_set = 16 * E + 8 * V  # both id columns read once, the vertex table written
alg_build = (32 * E + 8 * V) + (32 * E + 16 * V)
print(json.dumps({
    "workload": f"LDBC SNB {scale.upper()} knows table alone: distinct endpoint ids (sorted) + CSR build, per statement",
    "vertices": int(V), "edge_rows": int(E), "rows_2hop": st["rows"][2],
    "vertex_table_equals_np_unique": bool(np.array_equal(np.unique(np.concatenate([src, dst])).size, V)),
    "ms_per_step_wall": wall * 1e3, "kernels_us_per_step": {k: v[1] / 3 * 1e3 for k, v in prof.items()},
    "endpoint_set_us": set_us, "all_kernels_us": all_us,
    "hbm": {"endpoint_set": {"algorithmic_bytes": alg_set, "frac": alg_set / (set_us * 1e-6) / 8e12 if set_us else None},
            "whole": {"algorithmic_bytes": alg_set + alg_build, "frac": (alg_set + alg_build) / (all_us * 1e-6) / 8e12}}}))
g.close()
