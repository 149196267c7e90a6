#!/usr/bin/env python3
"""bench_bfs.py — secondary benchmark: LDBC SNB shortest_path(Person, Person), 64-source bitset BFS
(BASELINE.json configs[2]).  Not the driver's bench line (that is bench.py); results go to profiles/.

A step = one 64-source batch run to fixpoint on a prebuilt CSR (distances stay in HBM).  With N ranks,
each rank takes different source batches (source-batch sharding: no communication, weak scaling).
CPU baseline: the C oracle's bitset BFS on one batch; and, where oracle/_ref exists, the reference's
recursive CTE (bi-10 friends/friends_shortest) on a bounded hop count.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sf100")
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--max-hops", type=int, default=-1)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import duckdb_pgq_amd as pkg

    vid, src, dst = pkg.datagen.ldbc(args.workload)
    gg = pkg.GG(0)
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    csr = gg.build_csr()
    batches = [pkg.datagen.pick_sources(vid, 64, 0x5EED, batch=b) for b in range(args.batches)]
    gg.bfs64(csr, batches[0], args.max_hops, fetch=False)  # warm-up
    gg.profile_reset()
    gg.profile(True)
    t0 = time.perf_counter()
    te = act = lv = 0
    for b in batches:
        _, st = gg.bfs64(csr, b, args.max_hops, fetch=False)
        te += st["traversed_edges"]
        act += st["active_vertices"]
        lv += st["levels"]
    dt = time.perf_counter() - t0
    gg.profile(False)
    prof = gg.profile_get()
    V = csr.V
    alg = 8 * V * lv + 16 * act + 24 * te + 24 * V * lv  # SURVEY.md §8d per-level formula summed over levels
    kern_ms = sum(v[1] for k, v in prof.items() if k.startswith("bfs_"))
    line = {
        "metric": "traversed edges/sec, 64-source bitset BFS (shortest path)", "workload": args.workload,
        "batches": args.batches, "max_hops": args.max_hops, "value": te / dt, "unit": "traversed edges/s",
        "ms_per_batch": dt / args.batches * 1e3, "levels_per_batch": lv / args.batches,
        "roofline": {"bound": "hbm", "achieved": alg / (kern_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (kern_ms * 1e-3) / 8e12, "kernel_ms_per_batch": kern_ms / args.batches},
        "kernels": {k: {"launches": v[0], "us_per_batch": v[1] * 1e3 / args.batches} for k, v in prof.items()},
    }
    if not args.no_cpu:
        from tests import oracle_lib

        orc = oracle_lib.load()
        rc, g = orc.csr_build(vid, src, dst)
        t0 = time.perf_counter()
        d, ost = g.bfs64(g.lookup(batches[0]), args.max_hops)
        cdt = time.perf_counter() - t0
        _, gst = gg.bfs64(csr, batches[0], args.max_hops, fetch=False)
        dist, _ = gg.bfs64(csr, batches[0], args.max_hops)
        line["parity_vs_oracle"] = bool(np.array_equal(d, dist) and ost == gst)
        line["cpu_port"] = {"value": ost["traversed_edges"] / cdt, "unit": "traversed edges/s", "cores": 1,
                            "sample": f"one 64-source batch, C oracle bitset BFS, {cdt:.2f}s"}
        g.close()
    print(json.dumps(line))
    csr.close()
    gg.close()


if __name__ == "__main__":
    main()
