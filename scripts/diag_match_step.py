#!/usr/bin/env python3
"""Diagnostic: where one step of bench.py's region B (build + every 2-hop row written in middle-vertex parts) spends
its wall time outside k_mat_mid2 — build, count from degrees, partition, and per part: call, close.
    python3 scripts/diag_match_step.py [sf100] [budget GiB]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import duckdb_pgq_amd as pkg  # noqa: E402

torch.cuda.init()
scale = sys.argv[1] if len(sys.argv) > 1 else "sf100"
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
vid, src, dst = pkg.datagen.ldbc(scale)
g = pkg.GG(0)
g.set_edge_rowid(False)
g.chunk_rows = 122_880
g.append_vertices(vid)
g.append_edges(src, dst)
g.staging_sync()


def now():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(4):
    for with_stats in (True, False):
        t = [now()]
        csr = g.build_csr()
        t.append(now())
        total = g.khop_count(csr, 2, 2)[2]
        t.append(now())
        n_parts = max(1, int(-(-total * 24 // int(budget * 2**30))))
        bounds = g.khop_partition_mid(csr, n_parts) if n_parts > 1 else [0, csr.V]
        t.append(now())
        parts = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            a = now()
            res = g.expand_khop_mid_result(csr, lo, hi, k_min=2, with_stats=with_stats)
            b = now()
            rows = res.rows(2)
            res.close()
            c = now()
            parts.append((b - a, c - b, rows))
        csr.close()
        t.append(now())
        ms = lambda x: f"{x * 1e3:7.3f}"  # noqa: E731
        print(f"rep {rep} stats={int(with_stats)}: step {ms(t[-1] - t[0])} ms = build {ms(t[1] - t[0])} + count {ms(t[2] - t[1])} + partition "
              f"{ms(t[3] - t[2])} + parts {ms(sum(p[0] for p in parts))} (ideal at 8 TB/s {ms(total * 24 / 8e12)}) + closes "
              f"{ms(sum(p[1] for p in parts))} + rest {ms(t[-1] - t[3] - sum(p[0] + p[1] for p in parts))}", flush=True)
        if rep == 3:
            print("   per part ms:", " ".join(f"{p[0] * 1e3:.3f}/{p[2] * 24 / p[0] / 1e12:.2f}TB/s" for p in parts), flush=True)
g.close()
