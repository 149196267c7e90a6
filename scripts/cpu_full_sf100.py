#!/usr/bin/env python3
"""The reference's CPU time for the WHOLE headline workload, once (bench.py's cpu_baseline is a bounded sample plus an
estimate): count(*) of the 1-hop and of the 2-hop join chain over the SF100 tables on oracle/_ref/libduckdb.so with all
host threads, one cold run each (and a hot 2-hop run if it fits), counts checked against the C oracle.
    python3 scripts/cpu_full_sf100.py [scale] > profiles/r04_cpu_full_sf100.json      (TEST INFRASTRUCTURE: uses oracle/)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from duckdb_pgq_amd import datagen  # noqa: E402
from oracle import ref_duckdb as R  # noqa: E402
from tests import oracle_lib  # noqa: E402


def note(*a):
    print("[cpu_full]", *a, file=sys.stderr, flush=True)


scale = sys.argv[1] if len(sys.argv) > 1 else "sf100"
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 900.0
t_start = time.perf_counter()
vid, src, dst = datagen.ldbc(scale)
note(f"{scale}: V={vid.size} rows={src.size}")
cores = os.cpu_count() or 1
db = R.RefDuckDB(threads=cores)
t0 = time.perf_counter()
db.load_ldbc(vid, src, dst)
t_load = time.perf_counter() - t0
note(f"tables loaded in {t_load:.1f}s; threads={cores}")
rc, g = oracle_lib.load().csr_build(vid, src, dst)
assert rc == 0
ost = g.khop(1, 2)
note("oracle:", ost["rows"][1:3], "TE", ost["traversed_edges"])
runs = []
for h in (1, 2):
    c, dt = db.timed(R.sql_khop(h))
    runs.append({"statement": f"{h}-hop count(*)", "seconds": dt, "count": int(c[0, 0]), "run": "cold"})
    note(runs[-1])
left = budget_s - (time.perf_counter() - t_start)
if left > 1.3 * runs[1]["seconds"]:  # one hot run of the 2-hop statement, if the box's time allows
    c, dt = db.timed(R.sql_khop(2))
    runs.append({"statement": "2-hop count(*)", "seconds": dt, "count": int(c[0, 0]), "run": "hot"})
    note(runs[-1])
db.close()
two = min(r["seconds"] for r in runs if r["statement"].startswith("2-hop"))
total = runs[0]["seconds"] + two
cpu = "unknown"
for line in open("/proc/cpuinfo"):
    if line.startswith("model name"):
        cpu = line.split(":", 1)[1].strip()
        break
print(json.dumps({
    "what": f"reference DuckDB (oracle/_ref/libduckdb.so), LDBC {scale} synthetic tables, count(*) of the 1-hop and 2-hop "
            "join chains over ALL knows rows (the whole bench.py workload, not a sample)",
    "threads": cores, "cpu_model": cpu, "table_load_s": t_load, "runs": runs,
    "counts_match_oracle": bool(runs[0]["count"] == ost["rows"][1] and all(
        r["count"] == ost["rows"][2] for r in runs if r["statement"].startswith("2-hop"))),
    "traversed_edges": int(ost["traversed_edges"]), "seconds_1hop_plus_2hop": total,
    "value": ost["traversed_edges"] / total, "unit": "traversed edges/s"}, indent=1))
