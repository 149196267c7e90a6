#!/usr/bin/env python3
"""Diagnostic: one statement over and over inside the compiled reference, plain and on a pinned graph (looking for
intermittent faults under the reference's many pipeline threads).  usage: stress_sql.py [sf10] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from duckdb_pgq_amd import datagen  # noqa: E402
from oracle import ref_duckdb as R  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
vid, src, dst = datagen.ldbc(scale)
d = R.RefDuckDB(threads=os.cpu_count())
d.load_ldbc(vid, src, dst)
d.execute(f"LOAD '{R.EXTENSION}'")
sources = datagen.pick_sources(vid, 64, 1)
sql = R.sql_shortest(sources, 5).replace(", person p", "").replace("AND k.k_person2id = p.p_personid ", "").replace(
    "SELECT startPerson, friend, min(hopCount) AS hopCount FROM friends GROUP BY startPerson, friend",
    "SELECT count(*), sum(hopCount) FROM (SELECT startPerson, friend, min(hopCount) AS hopCount "
    "FROM friends GROUP BY startPerson, friend) t")
d.execute("PRAGMA enable_gpu_graph")
want = d.execute(sql).tolist()
for r in range(rounds):
    for _ in range(3):
        assert d.execute(sql).tolist() == want
    d.execute("PRAGMA gg_use_pinned_graphs")
    d.execute("SELECT * FROM gg_graph_pin('', '', 'knows', 'k_person1id', 'k_person2id')")
    for _ in range(7):
        assert d.execute(sql).tolist() == want
    d.execute("SELECT * FROM gg_graph_unpin()")
    d.execute("PRAGMA gg_ignore_pinned_graphs")
    print("round", r, "ok", flush=True)
d.close()
