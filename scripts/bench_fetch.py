#!/usr/bin/env python3
"""Diagnostic: draining a device-resident result from T host threads in 128 Ki-row slabs (what the pipeline's threads
do: gg_result_fetch into page-locked slabs) — the reached pairs of a 64-source BFS on SF100 (28.7 M packed rows, 230 MB)
and the rows of a 2-hop expansion from a source list (three id columns).  Env GG_FETCH_LANES = 1..4.
usage: bench_fetch.py [sf100]"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd.gg import BfsStats, KhopStats, _i64  # noqa: E402

scale = sys.argv[1] if len(sys.argv) > 1 else "sf100"
vid, src, dst = pkg.datagen.ldbc(scale)
g = pkg.GG(0)
g.set_edge_rowid(False)
g.append_edges(src, dst)
g.vertices_from_edges()
c = g.build_csr()
SLAB = int(os.environ.get("SLAB_ROWS", 131072))
i64p = C.POINTER(C.c_int64)


def drain(res, table, n_cols, rows, T):
    slabs = [[g.host_buffer(SLAB) for _ in range(n_cols)] for _ in range(T)]
    nxt = [0]
    lock = threading.Lock()

    def work(t):
        ptrs = (i64p * n_cols)(*[b.ctypes.data_as(i64p) for b in slabs[t]])
        got = C.c_uint32()
        while True:
            with lock:
                o = nxt[0]
                nxt[0] += SLAB
            if o >= rows:
                return
            g._chk(g.lib.gg_result_fetch(res, table, o, SLAB, ptrs, C.byref(got)))

    best = 1e9
    for rep in range(4):
        nxt[0] = 0
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        best = min(best, time.perf_counter() - t0)
    return best


sources = pkg.datagen.pick_sources(vid, 64, 1)
s, ps = _i64(sources)
st, res = BfsStats(), C.c_void_p()
g._chk(g.lib.gg_bfs64_pairs_packed(g.ctx, c.handle, ps, s.size, 5, C.byref(st), C.byref(res)))
n = C.c_uint64()
g._chk(g.lib.gg_result_rows(res, 0, C.byref(n)))
for T in ((1, 2, 4, 8, 16, 32) if "SLAB_ROWS" not in os.environ else (4, 8, 32)):
    dt = drain(res, 0, 1, n.value, T)
    print(f"BFS pairs, {n.value} rows x 1 column, {T:2d} threads: {dt * 1e3:7.2f} ms  {n.value * 8 / dt / 1e9:5.1f} GB/s", flush=True)
g.lib.gg_result_destroy(res)
many = pkg.datagen.pick_sources(vid, 2000, 3)
r = g.expand_khop_result(c, 2, sources=many)
rows = r.rows(2)
for T in (1, 4, 8, 32):
    dt = drain(r.handle, 2, 3, rows, T)
    print(f"2-hop rows, {rows} rows x 3 columns, {T:2d} threads: {dt * 1e3:7.2f} ms  {rows * 24 / dt / 1e9:5.1f} GB/s", flush=True)
r.close()
c.close()
g.close()
