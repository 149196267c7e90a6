#!/usr/bin/env python3
"""Diagnostic: time materialised 2-hop expansion (rows written to HBM as int64 ids) at SF10."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd.gg import KhopStats  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf10"
if len(sys.argv) > 2:  # a libgg variant (scripts/build_variants.py)
    from duckdb_pgq_amd import gg as ggmod
    ggmod._lib = ggmod.load_library(os.path.abspath(sys.argv[2]))
vid, src, dst = pkg.datagen.ldbc(scale)
g = pkg.GG(0)
g.append_vertices(vid)
g.append_edges(src, dst)
csr = g.build_csr()
if len(sys.argv) > 3 and sys.argv[3] == "frontier":  # the per-parent form (k_mat_last) instead of the product form
    g.force_frontier(True)
for rep in range(3):
    g.profile_reset()
    g.profile(True)
    st = KhopStats()
    res = C.c_void_p()
    t = time.perf_counter()
    g._chk(g.lib.gg_expand_khop_result(g.ctx, csr.handle, None, 0, 2, 2, C.byref(st), C.byref(res)))
    dt = time.perf_counter() - t
    g.profile(False)
    rows = st.rows[2]
    print(f"{scale}: {rows} rows x 3 cols int64 = {rows*24/1e9:.1f} GB in {dt*1e3:.1f} ms -> {rows*24/dt/1e12:.2f} TB/s written;",
          {k: round(v[1], 2) for k, v in g.profile_get().items() if v[1] > 0.05})
    g.lib.gg_result_destroy(res)
csr.close()
g.close()
