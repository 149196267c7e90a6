"""ctypes binding of include/gg.h (test/bench harness; the C-ABI itself is the product boundary)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgg.so")

GG_MAX_HOPS = 8
GG_BFS_LANES = 64
GG_CHUNK_ROWS = 1024

# every symbol include/gg.h declares (tests/test_abi.py checks the library exports exactly these)
SYMBOLS = [
    "gg_version", "gg_last_error", "gg_device_count", "gg_ctx_create", "gg_ctx_destroy",
    "gg_vertices_append", "gg_edges_append", "gg_staging_sync", "gg_staging_counts", "gg_staging_clear",
    "gg_ctx_set_edge_rowid", "gg_csr_build", "gg_csr_build_shard", "gg_csr_destroy", "gg_csr_info", "gg_csr_export",
    "gg_expand_khop", "gg_expand_khop_range", "gg_khop_count", "gg_expand_khop_dev", "gg_stream_wait", "gg_join_probe", "gg_khop_partition", "gg_expand_khop_mid", "gg_khop_partition_mid", "gg_expand_khop_mid_result",
    "gg_debug_force_frontier", "gg_debug_force_legacy_build", "gg_debug_scan_fault", "gg_debug_rank_mode",
    "gg_debug_max_grid_tiles", "gg_debug_reset", "gg_debug_placement",
    "gg_result_rows", "gg_result_fetch", "gg_result_destroy", "gg_expand_khop_result", "gg_result_digest",
    "gg_expand_khop_edges", "gg_result_fetch_edges",
    "gg_result_filter_common_neighbour", "gg_staging_clear_edges", "gg_vertices_from_edges",
    "gg_bfs64", "gg_bfs64_pairs", "gg_bfs64_pairs_packed", "gg_walk_endpoints", "gg_host_alloc", "gg_host_free", "gg_csr_lookup",
    "gg_bfs_sharded_begin", "gg_bfs_sharded_expand", "gg_bfs_sharded_words", "gg_bfs_sharded_commit",
    "gg_bfs_sharded_pairs", "gg_bfs_sharded_end", "gg_bfs_sharded_levels",
    "gg_profile_enable", "gg_profile_select", "gg_profile_reset", "gg_profile_count", "gg_profile_get",
]


class GGError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"gg error {code}: {msg}")
        self.code = code


class KhopStats(C.Structure):
    _fields_ = [
        ("rows", C.c_uint64 * (GG_MAX_HOPS + 1)),
        ("digest", C.c_uint64 * (GG_MAX_HOPS + 1)),
        ("traversed_edges", C.c_uint64),
        ("frontier_entries", C.c_uint64),
    ]


class BfsStats(C.Structure):
    _fields_ = [
        ("levels", C.c_uint32),
        ("traversed_edges", C.c_uint64),
        ("active_vertices", C.c_uint64),
        ("reached_pairs", C.c_uint64),
    ]


_lib = None


def load_library(path: str | None = None):
    """Load libgg.so; raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the gg hot path."
        )
    lib = C.CDLL(p)
    P, u64, i64p = C.c_void_p, C.c_uint64, C.POINTER(C.c_int64)
    lib.gg_version.restype = C.c_char_p
    lib.gg_last_error.restype = C.c_char_p
    lib.gg_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.gg_ctx_create.argtypes = [C.c_int, C.POINTER(P)]
    lib.gg_ctx_destroy.argtypes = [P]
    lib.gg_ctx_destroy.restype = None
    lib.gg_vertices_append.argtypes = [P, i64p, u64]
    lib.gg_edges_append.argtypes = [P, i64p, i64p, i64p, u64]
    lib.gg_staging_sync.argtypes = [P]
    lib.gg_staging_counts.argtypes = [P, C.POINTER(u64), C.POINTER(u64)]
    lib.gg_staging_clear.argtypes = [P]
    lib.gg_csr_build.argtypes = [P, C.POINTER(P)]
    lib.gg_csr_build_shard.argtypes = [P, C.c_int, C.c_int, C.POINTER(P)]
    lib.gg_ctx_set_edge_rowid.argtypes = [P, C.c_int]
    lib.gg_csr_destroy.argtypes = [P]
    lib.gg_csr_destroy.restype = None
    lib.gg_csr_info.argtypes = [P, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.gg_csr_export.argtypes = [P, i64p, i64p, i64p, i64p]
    lib.gg_expand_khop.argtypes = [P, P, i64p, u64, C.c_int, C.c_int, C.c_int, C.POINTER(KhopStats), C.POINTER(P)]
    lib.gg_expand_khop_range.argtypes = [P, P, u64, u64, C.c_int, C.c_int, C.c_int, C.POINTER(KhopStats), C.POINTER(P)]
    lib.gg_khop_partition.argtypes = [P, P, C.c_int, C.POINTER(u64)]
    lib.gg_khop_count.argtypes = [P, P, i64p, u64, C.c_int, C.c_int, C.POINTER(u64)]
    lib.gg_expand_khop_dev.argtypes = [P, P, C.c_int, C.POINTER(C.c_void_p)]
    lib.gg_stream_wait.argtypes = [P, C.c_void_p, C.c_int]
    lib.gg_join_probe.argtypes = [P, P, i64p, u64, C.POINTER(u64), C.POINTER(P)]
    lib.gg_expand_khop_mid.argtypes = [P, P, u64, u64, C.c_int, C.c_int, C.POINTER(KhopStats)]
    lib.gg_khop_partition_mid.argtypes = [P, P, C.c_int, C.POINTER(u64)]
    lib.gg_expand_khop_mid_result.argtypes = [P, P, u64, u64, C.c_int, C.POINTER(KhopStats), C.POINTER(P)]
    lib.gg_debug_force_frontier.argtypes = [P, C.c_int]
    lib.gg_debug_force_legacy_build.argtypes = [P, C.c_int]
    lib.gg_debug_rank_mode.argtypes = [P, C.c_int]
    lib.gg_debug_scan_fault.argtypes = [P, C.c_uint32, u64]
    lib.gg_debug_max_grid_tiles.argtypes = [P, u64]
    lib.gg_debug_reset.argtypes = [P]
    lib.gg_debug_placement.argtypes = [P, C.POINTER(u64), C.POINTER(u64)]
    lib.gg_result_rows.argtypes = [P, C.c_int, C.POINTER(u64)]
    lib.gg_result_fetch.argtypes = [P, C.c_int, u64, C.c_uint32, C.POINTER(i64p), C.POINTER(C.c_uint32)]
    lib.gg_expand_khop_result.argtypes = [P, P, i64p, u64, C.c_int, C.c_int, C.POINTER(KhopStats), C.POINTER(P)]
    lib.gg_result_filter_common_neighbour.argtypes = [P, P, C.c_int, P, C.POINTER(P)]
    lib.gg_staging_clear_edges.argtypes = [P]
    lib.gg_vertices_from_edges.argtypes = [P, C.c_int, C.POINTER(u64)]
    lib.gg_result_destroy.argtypes = [P]
    lib.gg_result_destroy.restype = None
    lib.gg_result_digest.argtypes = [P, P, P, C.c_int, C.POINTER(u64), C.POINTER(u64)]
    lib.gg_expand_khop_edges.argtypes = [P, P, i64p, u64, C.c_int, C.POINTER(KhopStats), C.POINTER(P)]
    lib.gg_result_fetch_edges.argtypes = [P, C.c_int, u64, C.c_uint32, C.POINTER(i64p), C.POINTER(C.c_uint32)]
    lib.gg_bfs64.argtypes = [P, P, i64p, C.c_int, C.c_int, i64p, u64, C.POINTER(C.c_int32), C.POINTER(BfsStats)]
    lib.gg_host_alloc.argtypes = [P, u64, C.POINTER(C.c_void_p)]
    lib.gg_host_free.argtypes = [P, C.c_void_p]
    lib.gg_host_free.restype = None
    lib.gg_csr_lookup.argtypes = [P, P, i64p, u64, C.POINTER(C.c_uint32)]
    lib.gg_bfs64_pairs.argtypes = [P, P, i64p, C.c_int, C.c_int, C.POINTER(BfsStats), C.POINTER(P)]
    lib.gg_bfs64_pairs_packed.argtypes = [P, P, i64p, C.c_int, C.c_int, C.POINTER(BfsStats), C.POINTER(P)]
    lib.gg_walk_endpoints.argtypes = [P, P, i64p, u64, C.c_int, C.POINTER(P)]
    lib.gg_bfs_sharded_begin.argtypes = [P, P, i64p, C.c_int, C.POINTER(P)]
    lib.gg_bfs_sharded_expand.argtypes = [P, C.POINTER(C.c_void_p), C.POINTER(u64), C.POINTER(u64)]
    lib.gg_bfs_sharded_words.argtypes = [P, C.POINTER(C.c_uint64), C.c_int]
    lib.gg_bfs_sharded_commit.argtypes = [P]
    lib.gg_bfs_sharded_pairs.argtypes = [P, C.POINTER(P)]
    lib.gg_bfs_sharded_levels.argtypes = [P, C.POINTER(u64), C.POINTER(u64)]
    lib.gg_bfs_sharded_end.argtypes = [P]
    lib.gg_bfs_sharded_end.restype = None
    lib.gg_profile_enable.argtypes = [P, C.c_int]
    lib.gg_profile_select.argtypes = [P, C.c_char_p]
    lib.gg_profile_reset.argtypes = [P]
    lib.gg_profile_count.argtypes = [P, C.POINTER(C.c_int)]
    lib.gg_profile_get.argtypes = [P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(u64), C.POINTER(C.c_double)]
    if path is None:
        _lib = lib
    return lib


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(C.POINTER(C.c_int64))


class ShardedBfs:
    """One rank's state of a graph-sharded BFS (gg_bfs_sharded_*)."""

    def __init__(self, gg: "GG", handle, V: int):
        self.gg, self.handle, self.V = gg, handle, V
        self.next_ptr = 0

    def expand(self) -> int:
        """Pull the next frontier words of the owned vertices; returns the rank's newly reached pairs."""
        ptr, n, new = C.c_void_p(), C.c_uint64(), C.c_uint64()
        self.gg._chk(self.gg.lib.gg_bfs_sharded_expand(self.handle, C.byref(ptr), C.byref(n), C.byref(new)))
        self.next_ptr = ptr.value or 0
        return int(new.value)

    @property
    def __cuda_array_interface__(self):
        """The rank's next-frontier words in HBM, for a device-side collective: torch.as_tensor(run, device="cuda")
        views them as int64[V] (a SUM all-reduce over disjoint supports is the OR of the ranks' words)."""
        return {"shape": (self.V,), "typestr": "<i8", "data": (self.next_ptr, False), "version": 2}

    def words(self) -> np.ndarray:
        out = np.empty(self.V, np.uint64)
        self.gg._chk(self.gg.lib.gg_bfs_sharded_words(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64)), 0))
        return out

    def set_words(self, words: np.ndarray):
        w = np.ascontiguousarray(words, np.uint64)
        self.gg._chk(self.gg.lib.gg_bfs_sharded_words(self.handle, w.ctypes.data_as(C.POINTER(C.c_uint64)), 1))

    def commit(self):
        self.gg._chk(self.gg.lib.gg_bfs_sharded_commit(self.handle))

    def levels(self):
        """(levels this rank pushed, levels it pulled) so far."""
        push, pull = C.c_uint64(), C.c_uint64()
        self.gg._chk(self.gg.lib.gg_bfs_sharded_levels(self.handle, C.byref(push), C.byref(pull)))
        return int(push.value), int(pull.value)

    def pairs(self) -> np.ndarray:
        i64p = C.POINTER(C.c_int64)
        res = C.c_void_p()
        self.gg._chk(self.gg.lib.gg_bfs_sharded_pairs(self.handle, C.byref(res)))
        try:
            n = C.c_uint64()
            self.gg._chk(self.gg.lib.gg_result_rows(res, 2, C.byref(n)))
            out = np.empty((3, n.value), np.int64)
            if n.value:
                ptrs = (i64p * 3)(*[out[c].ctypes.data_as(i64p) for c in range(3)])
                got = C.c_uint32()
                self.gg._chk(self.gg.lib.gg_result_fetch(res, 2, 0, n.value, ptrs, C.byref(got)))
        finally:
            self.gg.lib.gg_result_destroy(res)
        return out.T.copy()

    def close(self):
        if self.handle:
            self.gg.lib.gg_bfs_sharded_end(self.handle)
            self.handle = None


class DeviceWords:
    """n uint64 words in device memory owned by the library, viewable as a tensor (torch.as_tensor(words, device=...)):
    what a device-side collective reduces in place."""

    def __init__(self, ptr: int, n: int):
        self.ptr, self.n = ptr, n

    @property
    def __cuda_array_interface__(self):
        # (int64: torch has no uint64 arithmetic; sums of counts stay far below 2^63, digests are masked by the reader)
        return {"shape": (self.n,), "typestr": "<i8", "data": (self.ptr, False), "version": 2}


class KhopResult:
    """Materialised walks left in HBM (gg_expand_khop_result): row counts and <=1024-row slices on demand."""

    def __init__(self, gg: "GG", handle, stats):
        self.gg, self.handle, self.stats = gg, handle, stats

    def rows(self, h: int) -> int:
        n = C.c_uint64()
        self.gg._chk(self.gg.lib.gg_result_rows(self.handle, h, C.byref(n)))
        return int(n.value)

    def fetch_edges(self, h: int, offset: int, max_rows: int = GG_CHUNK_ROWS) -> np.ndarray:
        """Edge rowids e1..eh of rows [offset, offset + max_rows) (results of expand_khop_edges)."""
        bufs = [np.empty(GG_CHUNK_ROWS, np.int64) for _ in range(h)]
        ptrs = (C.POINTER(C.c_int64) * h)(*[b.ctypes.data_as(C.POINTER(C.c_int64)) for b in bufs])
        got = C.c_uint32()
        self.gg._chk(self.gg.lib.gg_result_fetch_edges(self.handle, h, offset, min(max_rows, GG_CHUNK_ROWS), ptrs, C.byref(got)))
        return np.stack([b[: got.value] for b in bufs], axis=1)

    def digest(self, csr, h: int):
        """(rows, digest) of the h-hop rows as they stand in HBM — the figures a count-mode expansion reports."""
        n, d = C.c_uint64(), C.c_uint64()
        self.gg._chk(self.gg.lib.gg_result_digest(self.gg.ctx, csr.handle, self.handle, h, C.byref(n), C.byref(d)))
        return int(n.value), int(d.value)

    def fetch(self, h: int, offset: int, max_rows: int = GG_CHUNK_ROWS) -> np.ndarray:
        bufs = [np.empty(GG_CHUNK_ROWS, np.int64) for _ in range(h + 1)]
        ptrs = (C.POINTER(C.c_int64) * (h + 1))(*[b.ctypes.data_as(C.POINTER(C.c_int64)) for b in bufs])
        got = C.c_uint32()
        self.gg._chk(self.gg.lib.gg_result_fetch(self.handle, h, offset, min(max_rows, GG_CHUNK_ROWS), ptrs, C.byref(got)))
        return np.stack([b[: got.value] for b in bufs], axis=1)

    def close(self):
        if self.handle:
            self.gg.lib.gg_result_destroy(self.handle)
            self.handle = None


class Csr:
    def __init__(self, gg: "GG", handle):
        self.gg, self.handle = gg, handle
        V, E, D = C.c_uint64(), C.c_uint64(), C.c_uint64()
        gg._chk(gg.lib.gg_csr_info(handle, C.byref(V), C.byref(E), C.byref(D)))
        self.V, self.E, self.dropped = V.value, E.value, D.value

    def export(self):
        off = np.empty(self.V + 1, np.int64)
        nbr = np.empty(self.E, np.int64)
        eid = np.empty(self.E, np.int64)
        vid = np.empty(self.V, np.int64)
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))  # noqa: E731
        self.gg._chk(self.gg.lib.gg_csr_export(self.handle, p(off), p(nbr), p(eid), p(vid)))
        return off, nbr, eid, vid

    def close(self):
        if self.handle:
            self.gg.lib.gg_csr_destroy(self.handle)
            self.handle = None


class GG:
    """One gg_ctx.  Mirrors the call sequence of the C++ operators: append (Sink) -> build (Finalize)
    -> expand / bfs (GetData)."""

    def __init__(self, device: int = 0, chunk_rows: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        self._chk(self.lib.gg_ctx_create(device, C.byref(h)))
        self.ctx = h
        self.chunk_rows = chunk_rows  # >0: append in DataChunk-sized pieces like the Sink would

    def _chk(self, rc: int):
        if rc != 0:
            raise GGError(rc, self.lib.gg_last_error().decode())

    def close(self):
        if self.ctx:
            self.lib.gg_ctx_destroy(self.ctx)
            self.ctx = None

    # ---- staging
    def append_vertices(self, ids):
        a, p = _i64(ids)
        step = self.chunk_rows or max(1, a.size)
        for o in range(0, a.size, step):
            n = min(step, a.size - o)
            self._chk(self.lib.gg_vertices_append(self.ctx, C.cast(C.addressof(p.contents) + 8 * o, C.POINTER(C.c_int64)), n))

    def append_edges(self, src, dst, rowid=None):
        s, ps = _i64(src)
        d, pd = _i64(dst)
        assert s.size == d.size
        if rowid is not None:
            r, pr = _i64(rowid)
        step = self.chunk_rows or max(1, s.size)
        i64p = C.POINTER(C.c_int64)
        for o in range(0, s.size, step):
            n = min(step, s.size - o)
            a = C.cast(C.addressof(ps.contents) + 8 * o, i64p)
            b = C.cast(C.addressof(pd.contents) + 8 * o, i64p)
            c = C.cast(C.addressof(pr.contents) + 8 * o, i64p) if rowid is not None else None
            self._chk(self.lib.gg_edges_append(self.ctx, a, b, c, n))

    def staging_sync(self):
        self._chk(self.lib.gg_staging_sync(self.ctx))

    def staging_clear(self):
        self._chk(self.lib.gg_staging_clear(self.ctx))

    def staging_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.gg_staging_counts(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- build
    def build_csr(self) -> Csr:
        h = C.c_void_p()
        self._chk(self.lib.gg_csr_build(self.ctx, C.byref(h)))
        return Csr(self, h)

    def set_edge_rowid(self, keep: bool):
        self._chk(self.lib.gg_ctx_set_edge_rowid(self.ctx, int(keep)))

    def build_csr_shard(self, part: int, n_parts: int) -> Csr:
        h = C.c_void_p()
        self._chk(self.lib.gg_csr_build_shard(self.ctx, part, n_parts, C.byref(h)))
        return Csr(self, h)

    # ---- k-hop
    @staticmethod
    def _stats_dict(st: KhopStats):
        return {
            "rows": list(st.rows),
            "digest": list(st.digest),
            "traversed_edges": st.traversed_edges,
            "frontier_entries": st.frontier_entries,
        }

    def _collect(self, res, k_min, k_max):
        out = {}
        try:
            for h in range(k_min, k_max + 1):
                n = C.c_uint64()
                self._chk(self.lib.gg_result_rows(res, h, C.byref(n)))
                table = np.empty((n.value, h + 1), np.int64)
                bufs = [np.empty(GG_CHUNK_ROWS, np.int64) for _ in range(h + 1)]
                ptrs = (C.POINTER(C.c_int64) * (h + 1))(*[b.ctypes.data_as(C.POINTER(C.c_int64)) for b in bufs])
                got = C.c_uint32()
                o = 0
                while o < n.value:  # <=1024-row slices: one DataChunk per fetch
                    self._chk(self.lib.gg_result_fetch(res, h, o, GG_CHUNK_ROWS, ptrs, C.byref(got)))
                    for c in range(h + 1):
                        table[o : o + got.value, c] = bufs[c][: got.value]
                    o += got.value
                out[h] = table
        finally:
            self.lib.gg_result_destroy(res)
        return out

    def expand_khop(self, csr: Csr, k_min: int, k_max: int, sources=None, materialise=False):
        st = KhopStats()
        res = C.c_void_p()
        if sources is None:
            rc = self.lib.gg_expand_khop(self.ctx, csr.handle, None, 0, k_min, k_max, int(materialise), C.byref(st), C.byref(res))
        else:
            a, p = _i64(sources)
            rc = self.lib.gg_expand_khop(self.ctx, csr.handle, p, a.size, k_min, k_max, int(materialise), C.byref(st), C.byref(res))
        self._chk(rc)
        d = self._stats_dict(st)
        if materialise:
            d["tables"] = self._collect(res, k_min, k_max)
        return d

    def expand_khop_dev(self, csr: Csr, k_min: int = 1) -> "DeviceWords":
        """All-sources 1..2-hop count + digest with the six result words left on the device (sharding.FIELDS order);
        nothing waits.  The returned view is valid until the next such call on this context."""
        ptr = C.c_void_p()
        self._chk(self.lib.gg_expand_khop_dev(self.ctx, csr.handle, k_min, C.byref(ptr)))
        return DeviceWords(ptr.value, 6)

    def stream_wait(self, other_stream: int, direction: int = 0):
        """direction 0: `other_stream` (raw hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) waits for the
        library's stream; 1: the library's stream waits for it.  No host synchronisation."""
        self._chk(self.lib.gg_stream_wait(self.ctx, C.c_void_p(other_stream), direction))

    def join_probe(self, csr: Csr, keys) -> np.ndarray:
        """(position in `keys`, rowid) for every edge row whose source equals keys[position] (gg_join_probe)."""
        a, p = _i64(keys)
        m, res = C.c_uint64(), C.c_void_p()
        self._chk(self.lib.gg_join_probe(self.ctx, csr.handle, p, a.size, C.byref(m), C.byref(res)))
        out = np.empty((m.value, 2), np.int64)
        bufs = [np.empty(GG_CHUNK_ROWS, np.int64) for _ in range(2)]
        ptrs = (C.POINTER(C.c_int64) * 2)(*[b.ctypes.data_as(C.POINTER(C.c_int64)) for b in bufs])
        got, o = C.c_uint32(), 0
        try:
            while o < m.value:
                self._chk(self.lib.gg_result_fetch(res, 1, o, GG_CHUNK_ROWS, ptrs, C.byref(got)))
                out[o:o + got.value, 0] = bufs[0][:got.value]
                out[o:o + got.value, 1] = bufs[1][:got.value]
                o += got.value
        finally:
            self.lib.gg_result_destroy(res)
        return out

    def khop_count(self, csr: Csr, k_min: int, k_max: int, sources=None) -> list:
        """Number of h-hop walks per length (index h), from degrees: no row, no digest (gg_khop_count)."""
        rows = (C.c_uint64 * (GG_MAX_HOPS + 1))()
        if sources is None:
            self._chk(self.lib.gg_khop_count(self.ctx, csr.handle, None, 0, k_min, k_max, rows))
        else:
            a, p = _i64(sources)
            self._chk(self.lib.gg_khop_count(self.ctx, csr.handle, p, a.size, k_min, k_max, rows))
        return list(rows)

    def expand_khop_result(self, csr: Csr, k: int, sources=None) -> KhopResult:
        """k-hop walks from `sources` (None: all vertices) materialised in HBM; nothing crosses PCIe."""
        st = KhopStats()
        res = C.c_void_p()
        if sources is None:
            sp, ns = None, 0
        else:
            a, sp = _i64(sources)
            ns = a.size
        self._chk(self.lib.gg_expand_khop_result(self.ctx, csr.handle, sp, ns, k, k, C.byref(st), C.byref(res)))
        return KhopResult(self, res, self._stats_dict(st))

    def expand_khop_edges(self, csr: Csr, k: int, sources=None) -> "KhopResult":
        """k-hop walks with the rowid of every edge taken (sources None: from every vertex)."""
        st = KhopStats()
        res = C.c_void_p()
        if sources is None:
            sp, ns = None, 0
        else:
            a, sp = _i64(sources)
            ns = a.size
        self._chk(self.lib.gg_expand_khop_edges(self.ctx, csr.handle, sp, ns, k, C.byref(st), C.byref(res)))
        return KhopResult(self, res, self._stats_dict(st))

    def staging_clear_edges(self):
        self._chk(self.lib.gg_staging_clear_edges(self.ctx))

    def vertices_from_edges(self, keep_staged: bool = False) -> int:
        """Vertex table := distinct endpoint ids of the staged edges (keep_staged: united with the ids
        already staged as vertices), ascending; returns their number."""
        n = C.c_uint64()
        self._chk(self.lib.gg_vertices_from_edges(self.ctx, int(keep_staged), C.byref(n)))
        return int(n.value)

    def connected_paths_same_neighbour(self, path_csr: Csr, filter_csr: Csr, hops: int, sources=None):
        """`hops`-hop walks over path_csr whose vertices all share a neighbour in filter_csr; rows (w, v0..vh)."""
        st = KhopStats()
        res, out = C.c_void_p(), C.c_void_p()
        if sources is None:
            sp, ns = None, 0
        else:
            a, sp = _i64(sources)
            ns = a.size
        self._chk(self.lib.gg_expand_khop_result(self.ctx, path_csr.handle, sp, ns, hops, hops, C.byref(st), C.byref(res)))
        try:
            self._chk(self.lib.gg_result_filter_common_neighbour(self.ctx, res, hops, filter_csr.handle, C.byref(out)))
        finally:
            self.lib.gg_result_destroy(res)
        return self._collect(out, hops + 1, hops + 1)[hops + 1]

    def expand_khop_range(self, csr: Csr, lo: int, hi: int, k_min: int, k_max: int, materialise=False):
        st = KhopStats()
        res = C.c_void_p()
        self._chk(self.lib.gg_expand_khop_range(self.ctx, csr.handle, lo, hi, k_min, k_max, int(materialise), C.byref(st), C.byref(res)))
        d = self._stats_dict(st)
        if materialise:
            d["tables"] = self._collect(res, k_min, k_max)
        return d

    def expand_khop_mid(self, csr: Csr, lo: int, hi: int, k_min: int = 1, k_max: int = 2):
        st = KhopStats()
        self._chk(self.lib.gg_expand_khop_mid(self.ctx, csr.handle, lo, hi, k_min, k_max, C.byref(st)))
        return self._stats_dict(st)

    def expand_khop_mid_result(self, csr: Csr, lo: int, hi: int, k_min: int = 2, with_stats: bool = True) -> "KhopResult":
        """The 2-hop rows (k_min == 1: also the 1-hop rows) with the middle vertex in [lo, hi), materialised in HBM.
        with_stats=False: no counting expansion (count + digest) in front of the rows; KhopResult.rows() still works."""
        st = KhopStats()
        res = C.c_void_p()
        self._chk(self.lib.gg_expand_khop_mid_result(self.ctx, csr.handle, lo, hi, k_min,
                                                     C.byref(st) if with_stats else None, C.byref(res)))
        return KhopResult(self, res, self._stats_dict(st) if with_stats else None)

    def khop_partition_mid(self, csr: Csr, n_parts: int):
        b = (C.c_uint64 * (n_parts + 1))()
        self._chk(self.lib.gg_khop_partition_mid(self.ctx, csr.handle, n_parts, b))
        return list(b)

    def force_frontier(self, on):
        """False / 0: normal; True / 1: frontier kernels only; 2 / 3: the product form of an explicit frontier's last hop /
        last two hops at any size (and no all-sources product kernels)."""
        self._chk(self.lib.gg_debug_force_frontier(self.ctx, int(on)))

    def scan_fault(self, spin_limit: int = 0, mute_tile: int = (1 << 64) - 1):
        self._chk(self.lib.gg_debug_scan_fault(self.ctx, spin_limit, mute_tile))

    def rank_mode(self, mode: int):
        """0: probe the LDS atomic order once (default), 1: ranks from ds_add_rtn, 2: ranks from match masks."""
        self._chk(self.lib.gg_debug_rank_mode(self.ctx, int(mode)))

    def force_legacy_build(self, on: bool):
        self._chk(self.lib.gg_debug_force_legacy_build(self.ctx, int(on)))

    def max_grid_tiles(self, n: int = 0):
        """Expansion launches of at most n workgroups (0: the hardware bound)."""
        self._chk(self.lib.gg_debug_max_grid_tiles(self.ctx, int(n)))

    def debug_reset(self):
        """Every testing knob and the edge-rowid switch back to its default."""
        self._chk(self.lib.gg_debug_reset(self.ctx))

    def placement(self):
        """(column sets placed by probing, fast pairs of the last set: 3 = each result column in its own memory rank)."""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.gg_debug_placement(self.ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def khop_partition(self, csr: Csr, n_parts: int):
        b = (C.c_uint64 * (n_parts + 1))()
        self._chk(self.lib.gg_khop_partition(self.ctx, csr.handle, n_parts, b))
        return list(b)

    # ---- bfs
    def bfs64(self, csr: Csr, sources, max_hops: int, targets=None, fetch=True):
        s, ps = _i64(sources)
        n_out = csr.V if targets is None else len(targets)
        st = BfsStats()
        if not fetch:  # stats only: distances stay on the device
            self._chk(self.lib.gg_bfs64(self.ctx, csr.handle, ps, s.size, max_hops, None, 0, None, C.byref(st)))
            return None, {"levels": st.levels, "traversed_edges": st.traversed_edges,
                          "active_vertices": st.active_vertices, "reached_pairs": st.reached_pairs}
        out = np.empty((s.size, n_out), np.int32)
        if targets is None:
            rc = self.lib.gg_bfs64(self.ctx, csr.handle, ps, s.size, max_hops, None, 0, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(st))
        else:
            t, pt = _i64(targets)
            rc = self.lib.gg_bfs64(self.ctx, csr.handle, ps, s.size, max_hops, pt, t.size, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(st))
        self._chk(rc)
        return out, {"levels": st.levels, "traversed_edges": st.traversed_edges, "active_vertices": st.active_vertices, "reached_pairs": st.reached_pairs}

    def lookup(self, csr: Csr, ids) -> np.ndarray:
        """Dense index of each id (uint32, 0xFFFFFFFF = not a vertex)."""
        a, pa = _i64(ids)
        out = np.empty(a.size, np.uint32)
        self._chk(self.lib.gg_csr_lookup(self.ctx, csr.handle, pa, a.size, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def host_buffer(self, n_int64: int) -> np.ndarray:
        """An int64 array in page-locked memory owned by the context (for gg_result_fetch at PCIe rate)."""
        p = C.c_void_p()
        self._chk(self.lib.gg_host_alloc(self.ctx, 8 * max(1, n_int64), C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int64)), shape=(max(1, n_int64),))[:n_int64]

    def bfs64_pairs_packed(self, csr: Csr, sources, max_hops: int) -> np.ndarray:
        """gg_bfs64_pairs_packed: one word per reached pair, lane << 58 | distance << 32 | dense vertex index."""
        i64p = C.POINTER(C.c_int64)
        s, ps = _i64(sources)
        st, res = BfsStats(), C.c_void_p()
        self._chk(self.lib.gg_bfs64_pairs_packed(self.ctx, csr.handle, ps, s.size, max_hops, C.byref(st), C.byref(res)))
        try:
            n = C.c_uint64()
            self._chk(self.lib.gg_result_rows(res, 0, C.byref(n)))
            out = np.empty(n.value, np.int64)
            if n.value:
                ptrs = (i64p * 1)(out.ctypes.data_as(i64p))
                got = C.c_uint32()
                self._chk(self.lib.gg_result_fetch(res, 0, 0, n.value, ptrs, C.byref(got)))
        finally:
            self.lib.gg_result_destroy(res)
        return out.view(np.uint64)

    def bfs64_pairs(self, csr: Csr, sources, max_hops: int):
        """Reached (source id, vertex id, distance) rows of one <=64-source batch, compacted on the device."""
        i64p = C.POINTER(C.c_int64)
        s, ps = _i64(sources)
        st, res = BfsStats(), C.c_void_p()
        self._chk(self.lib.gg_bfs64_pairs(self.ctx, csr.handle, ps, s.size, max_hops, C.byref(st), C.byref(res)))
        try:
            n = C.c_uint64()
            self._chk(self.lib.gg_result_rows(res, 2, C.byref(n)))
            out = np.empty((3, n.value), np.int64)
            if n.value:
                ptrs = (i64p * 3)(*[out[c].ctypes.data_as(i64p) for c in range(3)])
                got = C.c_uint32()
                self._chk(self.lib.gg_result_fetch(res, 2, 0, n.value, ptrs, C.byref(got)))  # one bulk copy
                assert got.value == n.value
        finally:
            self.lib.gg_result_destroy(res)
        return out.T.copy(), {"levels": st.levels, "traversed_edges": st.traversed_edges,
                              "active_vertices": st.active_vertices, "reached_pairs": st.reached_pairs}

    def walk_endpoints(self, csr: Csr, sources, k_max: int):
        """gg_walk_endpoints: (vertex ids, masks) — bit h of a mask: the vertex ends a walk of exactly h edges
        from one of the sources."""
        i64p = C.POINTER(C.c_int64)
        s, ps = _i64(sources)
        res = C.c_void_p()
        self._chk(self.lib.gg_walk_endpoints(self.ctx, csr.handle, ps, s.size, k_max, C.byref(res)))
        try:
            n = C.c_uint64()
            self._chk(self.lib.gg_result_rows(res, 1, C.byref(n)))
            out = np.empty((2, n.value), np.int64)
            if n.value:
                ptrs = (i64p * 2)(*[out[c].ctypes.data_as(i64p) for c in range(2)])
                got = C.c_uint32()
                self._chk(self.lib.gg_result_fetch(res, 1, 0, n.value, ptrs, C.byref(got)))
                assert got.value == n.value
        finally:
            self.lib.gg_result_destroy(res)
        return out[0].copy(), out[1].copy()

    # ---- graph-sharded BFS (one shard per GPU; see include/gg.h)
    def bfs_sharded_begin(self, shard: Csr, sources) -> "ShardedBfs":
        s, ps = _i64(sources)
        h = C.c_void_p()
        self._chk(self.lib.gg_bfs_sharded_begin(self.ctx, shard.handle, ps, s.size, C.byref(h)))
        return ShardedBfs(self, h, shard.V)

    def bfs_sharded_emulated(self, shards, sources, max_hops: int):
        """All ranks of a graph-sharded BFS in ONE process (the shards live on this context): the exchange is
        a host-side sum of the ranks' words.  Returns the union of the ranks' (source, vertex, distance) rows
        and the number of levels expanded — what N GPUs with an RCCL all-reduce between them produce."""
        runs = [self.bfs_sharded_begin(c, sources) for c in shards]
        try:
            level = 0
            while max_hops < 0 or level < max_hops:
                words = np.zeros(runs[0].V, np.uint64)
                new = 0
                for r in runs:
                    new += r.expand()
                    words += r.words()  # disjoint supports: the sum is the OR
                if new == 0:
                    break
                for r in runs:
                    r.set_words(words)
                    r.commit()
                level += 1
            rows = [r.pairs() for r in runs]
            self.last_sharded_levels = [r.levels() for r in runs]  # (pushed, pulled) per rank
            return np.concatenate(rows, axis=0), level
        finally:
            for r in runs:
                r.close()

    # ---- profiling
    def profile(self, on: bool):
        self._chk(self.lib.gg_profile_enable(self.ctx, int(on)))

    def profile_select(self, names=None):
        """Time only these kernels (list of names; None: all)."""
        self._chk(self.lib.gg_profile_select(self.ctx, None if names is None else ",".join(names).encode()))

    def profile_reset(self):
        self._chk(self.lib.gg_profile_reset(self.ctx))

    def profile_get(self):
        n = C.c_int()
        self._chk(self.lib.gg_profile_count(self.ctx, C.byref(n)))
        out = {}
        for i in range(n.value):
            name, cnt, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            self._chk(self.lib.gg_profile_get(self.ctx, i, C.byref(name), C.byref(cnt), C.byref(ms)))
            out[name.value.decode()] = (cnt.value, ms.value)
        return out
