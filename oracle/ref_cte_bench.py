#!/usr/bin/env python3
"""Time the reference's recursive-CTE shortest path (bi-10 friends/friends_shortest) on seeded
LDBC-shaped tables; TEST INFRASTRUCTURE ONLY.  Run once per reference variant (GG_REF_VARIANT= / hoisted):
    python oracle/ref_cte_bench.py sf1 [n_sources] [max_hops] [threads]   ->  one JSON line
The relation is summarised as (rows, sum of all fields mod 2^61) so that two variants can be compared."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from duckdb_pgq_amd import datagen  # noqa: E402
from oracle import ref_duckdb as R  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf1"
n_src = int(sys.argv[2]) if len(sys.argv) > 2 else 64
max_hops = int(sys.argv[3]) if len(sys.argv) > 3 else 5
threads = int(sys.argv[4]) if len(sys.argv) > 4 else (os.cpu_count() or 1)
vid, src, dst = datagen.ldbc(scale) if scale in datagen.LDBC_SIZES else datagen.ldbc_knows(*map(int, scale.split(",")))
db = R.RefDuckDB(threads=threads)
db.load_ldbc(vid, src, dst)
sources = datagen.pick_sources(vid, n_src, 0x5EED)
sql = R.sql_shortest(sources, max_hops)
times = []
for _ in range(3):  # 1 cold + 2 hot
    t = time.perf_counter()
    rel = db.execute(sql)
    times.append(time.perf_counter() - t)
db.close()
print(json.dumps({"variant": R.VARIANT or "stock", "workload": scale, "sources": n_src, "max_hops": max_hops,
                  "threads": threads, "rows": int(rel.shape[0]),
                  "checksum": int(rel.astype(np.uint64).sum(dtype=np.uint64) % (1 << 61)),
                  "cold_s": times[0], "hot_s": times[1:], "median_hot_s": float(np.median(times[1:]))}))
