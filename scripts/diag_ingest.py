#!/usr/bin/env python3
"""Diagnostic: the SF100 `count(*)` 2-hop statement inside the reference by PRAGMA threads — where the statement's
time goes when nearly all of it is reading `knows` and staging it (GG_TIMING=1 / GG_STAGING_TRACE=1 print the phases).
usage: diag_ingest.py [sf100] [threads,threads,...]      (env: GG_SINK_BATCH_ROWS, GG_NO_PIPELINE_SINKS, GG_INGEST_TASKS)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import ref_duckdb as R  # noqa: E402
from duckdb_pgq_amd import datagen  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf100"
threads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16, 32, 64, 128, 256]
EXT = os.path.join(ROOT, "duckdb_pgq_amd", "gg_duckdb.duckdb_extension")
vid, src, dst = datagen.ldbc(scale)
d = R.RefDuckDB(threads=threads[0])
d.load_ldbc(vid, src, dst)
d.execute(f"LOAD '{EXT}'")
d.execute("PRAGMA enable_gpu_graph")
sql = "SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id"
out = {}
for t in threads:
    d.execute(f"PRAGMA threads={t}")
    times = []
    for _ in range(7):
        t0 = time.perf_counter()
        r = d.execute(sql)
        times.append(time.perf_counter() - t0)
    times.sort()
    out[t] = {"best_ms": round(times[0] * 1e3, 2), "median_ms": round(times[len(times) // 2] * 1e3, 2)}
    print(f"threads {t:4d}: best {times[0] * 1e3:7.2f} ms, median {times[len(times) // 2] * 1e3:7.2f} ms, result {r.tolist()}",
          file=sys.stderr, flush=True)
print(json.dumps({"scale": scale, "env": {k: v for k, v in os.environ.items() if k.startswith("GG_")}, "by_threads": out}))
