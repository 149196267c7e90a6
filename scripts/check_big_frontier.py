#!/usr/bin/env python3
"""Frontier levels around 2^32 entries: V = 80 M vertices of out-degree ~8, every vertex listed as a source, walks of
3..5 edges in count mode — the explicit-frontier forms (two materialised levels of 0.64 G and 5.1 G entries under the
product kernel) against the all-sources forms and the counts from degrees.  Diagnostic.  usage: check_big_frontier.py [V] [E]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 640_000_000
rng = np.random.default_rng(23)
vid = np.arange(V, dtype=np.int64) * 3 + 5
s = np.repeat(np.arange(V, dtype=np.int64), E // V)  # every vertex exactly E/V out-edges: row counts are V (E/V)^h
d = rng.integers(0, V, s.size, dtype=np.int64)
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_vertices(vid)
g.append_edges(vid[s], vid[d])
c = g.build_csr()
out = {"V": V, "E": int(s.size), "cases": {}}
ok = True
for k_min, k_max in ((3, 3), (4, 4), (5, 5)):
    t = time.perf_counter()
    try:
        a = g.expand_khop(c, k_min, k_max)
    except pkg.GGError as e:  # a level past 2^32 entries is refused (GG_ERR_TOO_LARGE), never wrapped
        rec = {"refused": str(e), "count_from_degrees": g.khop_count(c, k_min, k_max)[k_max], "expected": V * (E // V) ** k_max}
        ok = ok and e.code == -5 and rec["count_from_degrees"] == rec["expected"]
        out["cases"][f"{k_min}..{k_max}"] = rec
        print(rec, flush=True)
        continue
    ta = time.perf_counter() - t
    t = time.perf_counter()
    try:
        b = g.expand_khop(c, k_min, k_max, sources=vid)
        err = None
    except pkg.GGError as e:
        b, err = None, str(e)
    tb = time.perf_counter() - t
    cnt = g.khop_count(c, k_min, k_max)
    rec = {"all_sources_s": round(ta, 2), "listed_s": round(tb, 2), "rows": int(a["rows"][k_max]),
           "frontier_entries_listed": None if b is None else int(b["frontier_entries"]),
           "count_from_degrees_equal": cnt[k_max] == a["rows"][k_max], "error": err,
           "equal": None if b is None else (a["rows"] == b["rows"] and a["digest"] == b["digest"])}
    ok = ok and rec["count_from_degrees_equal"] and (rec["equal"] or err is not None)
    out["cases"][f"{k_min}..{k_max}"] = rec
    print(rec, flush=True)
c.close()
g.close()
print(json.dumps(out))
sys.exit(0 if ok else 1)
