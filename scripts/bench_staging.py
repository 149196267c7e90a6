#!/usr/bin/env python3
"""Diagnostic: the staging path alone — gg_edges_append from T host threads (65 536-row calls, as a Sink thread's
batches) + gg_staging_sync, SF100's edge table (39.8 M rows, 637 MB over PCIe).  usage: bench_staging.py [sf100]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import duckdb_pgq_amd as pkg  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf100"
if len(sys.argv) > 2:  # a libgg variant (scripts/build_variants.py)
    from duckdb_pgq_amd import gg as ggmod
    ggmod._lib = ggmod.load_library(os.path.abspath(sys.argv[2]))
vid, src, dst = pkg.datagen.ldbc(scale)
n = src.size
CH = 65536
g = pkg.GG(0)
g.set_edge_rowid(False)
for T in (1, 4, 8, 16, 64):
    best = 1e9
    for rep in range(3):
        g.staging_clear()
        chunks = list(range(0, n, CH))
        nxt = [0]
        lock = threading.Lock()

        def work():
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= len(chunks):
                    return
                a = chunks[i]
                g.append_edges(src[a:a + CH], dst[a:a + CH])

        t0 = time.perf_counter()
        th = [threading.Thread(target=work) for _ in range(T)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        g.staging_sync()
        best = min(best, time.perf_counter() - t0)
    print(f"{T:3d} threads: {best * 1e3:7.1f} ms  {n * 16 / best / 1e9:6.1f} GB/s", flush=True)
g.close()
