"""End-to-end SQL timing inside the compiled reference: the same statement planned by the reference
(hash joins / recursive CTE) and by the planner rules (GPU operators), tables resident in the
reference's storage.  Unlike bench.py this INCLUDES reading the base tables out of DuckDB and staging
them over PCIe, i.e. what a user of the drop-in sees per query.

    python scripts/bench_sql.py [--scale sf10] [--threads N] [--cpu-runs 1] [--gpu-runs 3]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from duckdb_pgq_amd import datagen  # noqa: E402
from oracle import ref_duckdb as R  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = os.path.join(ROOT, "duckdb_pgq_amd", "gg_duckdb.duckdb_extension")
os.environ.setdefault("GG_CRASH_TRACE", "1")  # a backtrace on stderr if a worker thread faults


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", default="sf10")
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--cpu-runs", type=int, default=1)
    ap.add_argument("--gpu-runs", type=int, default=3)
    ap.add_argument("--skip-cpu", action="store_true")
    a = ap.parse_args()

    vid, src, dst = datagen.ldbc(a.scale)
    d = R.RefDuckDB(threads=a.threads)
    t = time.perf_counter()
    d.load_ldbc(vid, src, dst)
    load_s = time.perf_counter() - t
    d.execute(f"LOAD '{EXT}'")
    sources = datagen.pick_sources(vid, 64, 1)
    queries = {
        "count_2hop_edge_only": "SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id",
        "shortest_64src_5hops": R.sql_shortest(sources, 5).replace(
            ", person p", "").replace("AND k.k_person2id = p.p_personid ", "").replace(
            "SELECT startPerson, friend, min(hopCount) AS hopCount FROM friends GROUP BY startPerson, friend",
            "SELECT count(*), sum(hopCount) FROM (SELECT startPerson, friend, min(hopCount) AS hopCount "
            "FROM friends GROUP BY startPerson, friend) t"),
    }
    s0 = int(vid[7])
    # the friends + friends-of-friends derived table of interactive-complex-3.sql:3-12 (one source)
    queries["ic3_friends_of_friends"] = (
        f"select count(*) from (select k_person2id from knows where k_person1id = {s0} union "
        f"select k2.k_person2id from knows k1, knows k2 where k1.k_person1id = {s0} "
        f"and k1.k_person2id = k2.k_person1id and k2.k_person2id <> {s0}) f")
    out = {"scale": a.scale, "threads": a.threads, "rows": int(src.size), "load_s": round(load_s, 2), "queries": {}}
    for name, sql in queries.items():
        rec = {}
        d.execute("PRAGMA enable_gpu_graph")
        assert "GG_" in d.explain(sql), d.explain(sql)
        gpu, best = d.timed(sql, runs=a.gpu_runs)
        rec["gpu_s"] = round(best, 4)
        rec["result"] = gpu.tolist()
        # the same statement with the graph pinned on the device (gg_graph_pin): no ingest, no build
        d.execute("PRAGMA gg_use_pinned_graphs")  # pinned graphs are opt-in per connection (snapshot semantics)
        d.execute("SELECT * FROM gg_graph_pin('', '', 'knows', 'k_person1id', 'k_person2id')")
        pinned, best = d.timed(sql, runs=a.gpu_runs)
        rec["gpu_pinned_s"] = round(best, 5)
        assert np.array_equal(pinned, gpu)
        d.execute("SELECT * FROM gg_graph_unpin()")
        d.execute("PRAGMA gg_ignore_pinned_graphs")
        d.execute("PRAGMA disable_gpu_graph")
        if not a.skip_cpu:
            cpu, best = d.timed(sql, runs=a.cpu_runs)
            rec["cpu_s"] = round(best, 4)
            rec["equal"] = bool(np.array_equal(cpu, gpu))
            rec["speedup"] = round(rec["cpu_s"] / rec["gpu_s"], 1)
        out["queries"][name] = rec
        print(json.dumps({name: rec}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
