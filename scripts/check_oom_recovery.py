#!/usr/bin/env python3
"""A result that cannot fit: every 2-hop row of SF100 (306 GB) asked for in ONE part.  The call must fail with
GG_ERR_OOM (-3) — no crash, no partial result — and the context must go on working: the same rows in parts afterwards,
digest equal to the counting expansion.  Diagnostic.  usage: check_oom_recovery.py [sf100]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf100"
vid, src, dst = pkg.datagen.ldbc(scale)
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_vertices(vid)
g.append_edges(src, dst)
c = g.build_csr()
counted = g.expand_khop(c, 2, 2)
out = {"rows": counted["rows"][2], "bytes_wanted": counted["rows"][2] * 24}
try:
    res = g.expand_khop_mid_result(c, 0, c.V, k_min=2)
    out["one_part"] = f"succeeded with {res.rows(2)} rows"
    res.close()
except pkg.GGError as e:
    out["one_part"] = {"code": e.code, "message": str(e)}
rows = dig = 0
bounds = g.khop_partition_mid(c, 10)
for lo, hi in zip(bounds[:-1], bounds[1:]):
    res = g.expand_khop_mid_result(c, lo, hi, k_min=2, with_stats=False)
    n, d = res.digest(c, 2)
    rows += n
    dig = (dig + d) & 0xFFFFFFFF
    res.close()
out["parts_after_the_failure"] = {"rows": rows, "digest": dig, "equal_to_count": [rows, dig] == [counted["rows"][2], counted["digest"][2]]}
c.close()
g.close()
print(json.dumps(out))
ok = isinstance(out["one_part"], dict) and out["one_part"]["code"] == -3 and out["parts_after_the_failure"]["equal_to_count"]
sys.exit(0 if ok else 1)
