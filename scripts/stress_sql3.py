#!/usr/bin/env python3
"""Diagnostic: the shortest-path statement with several 64-source batches (200 seeds), drained by the reference's
pipeline threads, over and over; every run must return the first run's relation.  usage: stress_sql3.py [sf10] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GG_CRASH_TRACE", "1")
from duckdb_pgq_amd import datagen  # noqa: E402
from oracle import ref_duckdb as R  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
vid, src, dst = datagen.ldbc(scale)
d = R.RefDuckDB(threads=os.cpu_count())
d.load_ldbc(vid, src, dst)
d.execute(f"LOAD '{R.EXTENSION}'")
seeds = [int(v) for v in datagen.pick_sources(vid, 64, 3)] + [int(v) for v in vid[:136]] + [-7, -8]
sql = R.sql_shortest(seeds, 4).replace(", person p", "").replace("AND k.k_person2id = p.p_personid ", "").replace(
    "SELECT startPerson, friend, min(hopCount) AS hopCount FROM friends GROUP BY startPerson, friend",
    "SELECT count(*), sum(hopCount), sum(startPerson % 1000), sum(friend % 1000) FROM (SELECT startPerson, friend, "
    "min(hopCount) AS hopCount FROM friends GROUP BY startPerson, friend) t")
d.execute("PRAGMA enable_gpu_graph")
assert "GG_SHORTEST_PATH" in d.explain(sql), d.explain(sql)
want = d.execute(sql).tolist()
print("relation:", want, flush=True)
for r in range(rounds):
    got = d.execute(sql).tolist()
    assert got == want, (r, got, want)
    if r % 10 == 9:
        print("round", r, "ok", flush=True)
d.close()
