#!/usr/bin/env python3
"""Diagnostic: rate of k_mat_mid2 by part size — SF10 and SF100 2-hop rows materialised in middle-vertex parts of
several budgets (is SF100's 5.5 TB/s against SF10's 6.9 a matter of the part's size or of the graph?).
    python3 scripts/ab_mat.py [scales] [budgets GiB]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import duckdb_pgq_amd as pkg  # noqa: E402

scales = (sys.argv[1] if len(sys.argv) > 1 else "sf10,sf100").split(",")
budgets = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,16,26,48").split(",")]
for scale in scales:
    vid, src, dst = pkg.datagen.ldbc(scale)
    g = pkg.GG(0)
    g.set_edge_rowid(False)
    g.append_vertices(vid)
    g.append_edges(src, dst)
    csr = g.build_csr()
    total = g.khop_count(csr, 2, 2)[2]
    for budget in budgets:
        n_parts = max(1, int(np.ceil(total * 24 / (budget * 2**30))))
        bounds = g.khop_partition_mid(csr, n_parts)
        for rep in range(2):
            g.profile_reset()
            g.profile_select(["mat_mid2"])
            g.profile(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rows = 0
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                res = g.expand_khop_mid_result(csr, lo, hi, k_min=2)
                rows += res.rows(2)
                res.close()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            g.profile(False)
            k = g.profile_get().get("mat_mid2", (0, 0.0))
        print(f"{scale} budget {budget:5.1f} GiB: {n_parts:3d} parts, {rows} rows, wall {dt*1e3:7.2f} ms ({rows*24/dt/1e12:.2f} TB/s), "
              f"mat_mid2 {k[1]:7.2f} ms ({rows*24/(k[1]*1e-3)/1e12:.2f} TB/s)", flush=True)
    csr.close()
    g.close()
