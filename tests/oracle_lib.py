"""ctypes binding of oracle/liboracle.so (the CPU oracle; test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
MAX_HOPS = 8


class Rows(C.Structure):
    _fields_ = [("ncols", C.c_int), ("n", C.c_uint64), ("cap", C.c_uint64), ("data", C.POINTER(C.c_int64))]


class Jht(C.Structure):
    _fields_ = [("count", C.c_uint64), ("capacity", C.c_uint64), ("bitmask", C.c_uint64),
                ("heads", C.POINTER(C.c_int64)), ("next", C.POINTER(C.c_int64)), ("keys", C.POINTER(C.c_int64))]


class CsrS(C.Structure):
    _fields_ = [("V", C.c_uint64), ("E", C.c_uint64), ("dropped", C.c_uint64), ("off", C.POINTER(C.c_int64)),
                ("nbr", C.POINTER(C.c_uint32)), ("eid", C.POINTER(C.c_int64)), ("vid", C.POINTER(C.c_int64))]


class KhopStats(C.Structure):
    _fields_ = [("rows", C.c_uint64 * (MAX_HOPS + 1)), ("digest", C.c_uint64 * (MAX_HOPS + 1)),
                ("traversed_edges", C.c_uint64), ("frontier_entries", C.c_uint64)]


class BfsStats(C.Structure):
    _fields_ = [("levels", C.c_uint32), ("traversed_edges", C.c_uint64), ("active_vertices", C.c_uint64),
                ("reached_pairs", C.c_uint64)]


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _rows_to_np(r: Rows) -> np.ndarray:
    if r.n == 0:
        return np.zeros((0, r.ncols), np.int64)
    return np.ctypeslib.as_array(r.data, shape=(r.n, r.ncols)).copy()


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_fmix64.restype = C.c_uint64
        L.orc_fmix64.argtypes = [C.c_uint64]
        L.orc_row_hash.restype = C.c_uint64
        L.orc_row_hash.argtypes = [C.POINTER(C.c_uint32), C.c_int]
        L.orc_rows_free.argtypes = [C.POINTER(Rows)]
        L.orc_rows_free.restype = None
        L.orc_rows_sort.argtypes = [C.POINTER(Rows)]
        L.orc_rows_sort.restype = None
        L.orc_jht_build.argtypes = [C.POINTER(Jht), C.POINTER(C.c_int64), C.c_uint64]
        L.orc_jht_free.argtypes = [C.POINTER(Jht)]
        L.orc_jht_free.restype = None
        L.orc_jht_probe.argtypes = [C.POINTER(Jht), C.POINTER(C.c_int64), C.c_uint64, C.POINTER(Rows)]
        L.orc_khop_join.argtypes = [C.POINTER(C.c_int64), C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.c_uint64, C.POINTER(C.c_int64), C.c_uint64, C.c_int, C.c_int, C.POINTER(Rows)]
        L.orc_cte_shortest.argtypes = [C.POINTER(C.c_int64), C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                       C.c_uint64, C.POINTER(C.c_int64), C.c_uint64, C.c_int, C.POINTER(Rows)]
        L.orc_csr_build.argtypes = [C.POINTER(CsrS), C.POINTER(C.c_int64), C.c_uint64, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_uint64]
        L.orc_csr_free.argtypes = [C.POINTER(CsrS)]
        L.orc_csr_free.restype = None
        L.orc_csr_lookup.argtypes = [C.POINTER(CsrS), C.c_int64]
        L.orc_csr_lookup.restype = C.c_int64
        L.orc_khop_csr.argtypes = [C.POINTER(CsrS), C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.c_uint64,
                                   C.c_int, C.c_int, C.c_int, C.POINTER(KhopStats)]
        L.orc_khop_csr_rows.argtypes = [C.POINTER(CsrS), C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.c_uint64,
                                        C.c_int, C.c_int, C.POINTER(Rows)]
        L.orc_bfs64_csr.argtypes = [C.POINTER(CsrS), C.POINTER(C.c_int64), C.c_int, C.c_int, C.POINTER(C.c_int32),
                                    C.POINTER(BfsStats)]
        L.orc_num_threads.restype = C.c_int

    # ---- digest
    def row_hash(self, dense_row) -> int:
        a = np.ascontiguousarray(dense_row, dtype=np.uint32)
        return int(self.lib.orc_row_hash(a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size - 1))

    def digest_rows(self, dense_rows: np.ndarray) -> int:
        s = 0  # 32-bit digest: low halves of the row hashes, summed mod 2^32 (DESIGN.md "Row digest")
        for r in dense_rows:
            s = (s + (self.row_hash(r) & 0xFFFFFFFF)) & 0xFFFFFFFF
        return s

    # ---- join restatement
    def hash_join(self, build_keys, probe_keys) -> np.ndarray:
        b = np.ascontiguousarray(build_keys, np.int64)
        p = np.ascontiguousarray(probe_keys, np.int64)
        ht = Jht()
        assert self.lib.orc_jht_build(C.byref(ht), _p64(b), b.size) == 0
        out = Rows(2, 0, 0, None)
        assert self.lib.orc_jht_probe(C.byref(ht), _p64(p), p.size, C.byref(out)) == 0
        r = _rows_to_np(out)
        self.lib.orc_rows_free(C.byref(out))
        self.lib.orc_jht_free(C.byref(ht))
        return r

    def khop_join(self, vid, src, dst, k_min, k_max, sources=None):
        """{h: rows of dense indices} through the hash-join chain (reference operators)."""
        vid = np.ascontiguousarray(vid, np.int64)
        src = np.ascontiguousarray(src, np.int64)
        dst = np.ascontiguousarray(dst, np.int64)
        out = (Rows * (MAX_HOPS + 1))()
        if sources is None:
            sp, ns = None, 0
        else:
            s = np.ascontiguousarray(sources, np.int64)
            sp, ns = _p64(s), s.size
        rc = self.lib.orc_khop_join(_p64(vid), vid.size, _p64(src), _p64(dst), src.size, sp, ns, k_min, k_max, out)
        assert rc == 0, rc
        res = {}
        for h in range(k_min, k_max + 1):
            res[h] = _rows_to_np(out[h])
        for h in range(MAX_HOPS + 1):
            self.lib.orc_rows_free(C.byref(out[h]))
        return res

    def cte_shortest(self, vid, src, dst, sources, max_hops) -> np.ndarray:
        vid = np.ascontiguousarray(vid, np.int64)
        src = np.ascontiguousarray(src, np.int64)
        dst = np.ascontiguousarray(dst, np.int64)
        s = np.ascontiguousarray(sources, np.int64)
        out = Rows(3, 0, 0, None)
        rc = self.lib.orc_cte_shortest(_p64(vid), vid.size, _p64(src), _p64(dst), src.size, _p64(s), s.size, max_hops, C.byref(out))
        assert rc == 0, rc
        r = _rows_to_np(out)
        self.lib.orc_rows_free(C.byref(out))
        return r

    # ---- CSR formulation
    def csr_build(self, vid, src, dst, rowid=None):
        vid = np.ascontiguousarray(vid, np.int64)
        src = np.ascontiguousarray(src, np.int64)
        dst = np.ascontiguousarray(dst, np.int64)
        g = CsrS()
        rp = None
        if rowid is not None:
            rowid = np.ascontiguousarray(rowid, np.int64)
            rp = _p64(rowid)
        rc = self.lib.orc_csr_build(C.byref(g), _p64(vid), vid.size, _p64(src), _p64(dst), rp, src.size)
        return rc, OracleCsr(self, g)

    def num_threads(self):
        return int(self.lib.orc_num_threads())


class OracleCsr:
    def __init__(self, orc: Oracle, g: CsrS):
        self.orc, self.g = orc, g

    @property
    def V(self):
        return self.g.V

    @property
    def E(self):
        return self.g.E

    @property
    def dropped(self):
        return self.g.dropped

    def arrays(self):
        V, E = self.g.V, self.g.E
        off = np.ctypeslib.as_array(self.g.off, shape=(V + 1,)).copy()
        nbr = np.ctypeslib.as_array(self.g.nbr, shape=(max(E, 1),))[:E].astype(np.int64)
        eid = np.ctypeslib.as_array(self.g.eid, shape=(max(E, 1),))[:E].copy()
        vid = np.ctypeslib.as_array(self.g.vid, shape=(max(V, 1),))[:V].copy()
        return off, nbr, eid, vid

    def lookup(self, ids):
        return np.array([self.orc.lib.orc_csr_lookup(C.byref(self.g), int(i)) for i in ids], np.int64)

    def khop(self, k_min, k_max, sources_dense=None, lo=0, hi=None, threads=0):
        st = KhopStats()
        if sources_dense is None:
            hi = self.g.V if hi is None else hi
            rc = self.orc.lib.orc_khop_csr(C.byref(self.g), None, 0, lo, hi, k_min, k_max, threads, C.byref(st))
        else:
            a = np.ascontiguousarray(sources_dense, np.uint32)
            rc = self.orc.lib.orc_khop_csr(C.byref(self.g), a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, 0, 0, k_min, k_max, threads, C.byref(st))
        assert rc == 0, rc
        return {"rows": list(st.rows), "digest": list(st.digest), "traversed_edges": st.traversed_edges,
                "frontier_entries": st.frontier_entries}

    def khop_rows(self, k_min, k_max, sources_dense=None, lo=0, hi=None):
        out = (Rows * (MAX_HOPS + 1))()
        if sources_dense is None:
            hi = self.g.V if hi is None else hi
            rc = self.orc.lib.orc_khop_csr_rows(C.byref(self.g), None, 0, lo, hi, k_min, k_max, out)
        else:
            a = np.ascontiguousarray(sources_dense, np.uint32)
            rc = self.orc.lib.orc_khop_csr_rows(C.byref(self.g), a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, 0, 0, k_min, k_max, out)
        assert rc == 0, rc
        res = {h: _rows_to_np(out[h]) for h in range(k_min, k_max + 1)}
        for h in range(MAX_HOPS + 1):
            self.orc.lib.orc_rows_free(C.byref(out[h]))
        return res

    def bfs64(self, sources_dense, max_hops):
        s = np.ascontiguousarray(sources_dense, np.int64)
        dist = np.empty((s.size, self.g.V), np.int32)
        st = BfsStats()
        rc = self.orc.lib.orc_bfs64_csr(C.byref(self.g), _p64(s), s.size, max_hops, dist.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(st))
        assert rc == 0, rc
        return dist, {"levels": st.levels, "traversed_edges": st.traversed_edges, "active_vertices": st.active_vertices,
                      "reached_pairs": st.reached_pairs}

    def close(self):
        self.orc.lib.orc_csr_free(C.byref(self.g))


_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        src = os.path.join(ORACLE_DIR, "gg_oracle.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
        _cached = Oracle(C.CDLL(LIB))
    return _cached


def sort_rows(a: np.ndarray) -> np.ndarray:
    if a.shape[0] == 0:
        return a
    return a[np.lexsort(a.T[::-1])]
