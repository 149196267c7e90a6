#!/usr/bin/env python3
"""Diagnostic: several substituted statements in a loop, through pipeline sinks and through the scan-function route
(GG_NO_PIPELINE_SINKS), results compared with the reference's own plan every time.  usage: stress_sql2.py [sf10] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GG_CRASH_TRACE", "1")
import numpy as np  # noqa: E402
from duckdb_pgq_amd import datagen  # noqa: E402
from oracle import ref_duckdb as R  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
vid, src, dst = datagen.ldbc(scale)
d = R.RefDuckDB(threads=os.cpu_count())
d.load_table("person", {"p_personid": vid})
d.load_table("knows", {"k_person1id": src, "k_person2id": dst, "w": src % 1000 + dst % 7})  # (a payload column)
d.execute(f"LOAD '{R.EXTENSION}'")
s0, s1 = int(vid[7]), int(vid[11])
sources = datagen.pick_sources(vid, 64, 1)
stmts = {
    "count2": "SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id",
    "rows_from_source": f"SELECT k1.k_person1id, k2.k_person2id FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id AND k1.k_person1id = {s0}",
    "payload": f"SELECT k1.w, k2.w, k2.k_person2id FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id AND k1.k_person1id = {s1}",
    "ic3": f"select count(*) from (select k_person2id from knows where k_person1id = {s0} union select k2.k_person2id from knows k1, "
           f"knows k2 where k1.k_person1id = {s0} and k1.k_person2id = k2.k_person1id and k2.k_person2id <> {s0}) f",
    "shortest": R.sql_shortest(sources, 4).replace(", person p", "").replace("AND k.k_person2id = p.p_personid ", "").replace(
        "SELECT startPerson, friend, min(hopCount) AS hopCount FROM friends GROUP BY startPerson, friend",
        "SELECT count(*), sum(hopCount) FROM (SELECT startPerson, friend, min(hopCount) AS hopCount FROM friends GROUP BY startPerson, friend) t"),
}


def norm(a):
    return sorted(map(tuple, a.tolist()))


want = {k: norm(d.execute(q)) for k, q in stmts.items() if k != "shortest"}
d.execute("PRAGMA enable_gpu_graph")
want["shortest"] = norm(d.execute(stmts["shortest"]))  # (the reference's recursive CTE takes minutes at this size)
for k, q in stmts.items():
    assert "GG_" in d.explain(q), (k, d.explain(q))
for r in range(rounds):
    for route in ("sinks", "scan"):
        if route == "scan":
            os.environ["GG_NO_PIPELINE_SINKS"] = "1"
        else:
            os.environ.pop("GG_NO_PIPELINE_SINKS", None)
        for k, q in stmts.items():
            got = norm(d.execute(q))
            assert got == want[k], (r, route, k)
    print("round", r, "ok", flush=True)
d.close()
