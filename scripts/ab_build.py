#!/usr/bin/env python3
"""A/B timing of libgg variants on one box: per-kernel times of CSR build + 2-hop expansion (diagnostic).
usage: ab_build.py sf100 libA.so libB.so ...   (results of a variant are checked against the first one)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd import gg as ggmod  # noqa: E402

scale = sys.argv[1]
libs = sys.argv[2:]
rowid = os.environ.get("AB_ROWID", "0") == "1"
legacy = os.environ.get("AB_LEGACY", "0") == "1"
vid, src, dst = pkg.datagen.ldbc(scale)
ids_mode = os.environ.get("AB_IDS", "")
if ids_mode:  # same graph, other vertex ids: "dense" (direct-address array) or "mulN" (ids = shuffled rank * N)
    import numpy as np
    order = np.argsort(vid, kind="stable")
    svid = vid[order]
    ds, dd = order[np.searchsorted(svid, src)], order[np.searchsorted(svid, dst)]
    rng = np.random.default_rng(1)
    mul = 1 if ids_mode == "dense" else int(ids_mode[3:])
    vid = (rng.permutation(vid.size).astype(np.int64)) * mul + 12345
    src, dst = vid[ds], vid[dd]
if os.environ.get("AB_SORT", "0") == "1":  # an edge table sorted by source id (as LDBC ships knows)
    import numpy as np
    order = np.argsort(src, kind="stable")
    src, dst = src[order], dst[order]
out = {}
ref = None
for rep in range(2):
    for lib in libs:
        ggmod._lib = ggmod.load_library(os.path.abspath(lib))
        g = pkg.GG(0)
        g.set_edge_rowid(rowid)
        if legacy:
            g.force_legacy_build(True)
        g.append_vertices(vid)
        g.append_edges(src, dst)
        c = g.build_csr()
        st = g.expand_khop(c, 1, 2)
        if ref is None:
            ref = st
        ok = st == ref
        g.profile_reset()
        g.profile(True)
        n = 10
        for _ in range(n):
            c2 = g.build_csr()
            g.expand_khop(c2, 1, 2)
            c2.close()
        g.profile(False)
        prof = g.profile_get()
        row = {k: round(v[1] / n * 1e3, 1) for k, v in prof.items() if v[0] and v[1] / n > 0.002}
        build_us = sum(v for k, v in row.items() if not k.startswith("expand") and k not in ("reduce_partials", "tile_partition"))
        out.setdefault(lib, []).append({"ok": ok, "build_us": round(build_us, 1), **row})
        c.close()
        g.close()
for lib, runs in out.items():
    print(lib)
    for r in runs:
        print("   ", json.dumps(r))
