"""Driver for the compiled reference (oracle/_ref/libduckdb.so) — TEST INFRASTRUCTURE ONLY.

Used by tests/golden/make_golden.py (fixtures), tests/test_reference_ref.py (pins the C oracle to the
real reference where the .so is present) and bench.py's cpu_baseline leg (kind "reference").  Talks to
the reference exclusively through its public C API (src/include/duckdb.h: duckdb_open :170,
duckdb_connect :203, duckdb_query :286, duckdb_value_int64 :428) plus oracle/ref_driver.cpp for bulk
appends.  Never imported by the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# GG_REF_VARIANT=hoisted: the reference with oracle/hoisted_build.patch (hash tables of a recursive CTE's invariant
# builds kept across iterations: the "best CPU" baseline).  One variant per process: both export the same symbols.
# GG_REF_VARIANT=patched: the reference with oracle/callout.patch (planner call-outs and the BuildPipelines case the
# product's operators need, as the maintainers would add them): the extension registers its rules there, the
# interposition shim is NOT loaded, and the extension is the build without access to private members.
VARIANT = os.environ.get("GG_REF_VARIANT", "")
_DIR = {"hoisted": "_ref_hoisted", "patched": "_ref_patched"}.get(VARIANT, "_ref")
LIBDUCKDB = os.path.join(HERE, _DIR, "libduckdb.so")
LIBGGREF = os.path.join(HERE, _DIR, "libggref.so")
_PKG = os.path.join(os.path.dirname(HERE), "duckdb_pgq_amd")
# interposition shim of the product's planner rules (duckdb_pgq_amd/host/gg_plan_hook.c): a pass-through
# until the extension registers its rules, but it has to be in the global scope BEFORE libduckdb
PLAN_HOOK = os.path.join(_PKG, "libgg_plan_hook.so")
# the loadable extension that goes with this variant of the reference
EXTENSION = os.path.join(_PKG, "callouts" if VARIANT == "patched" else "", "gg_duckdb.duckdb_extension")


def rules_route() -> str:
    """How the planner rules reach this reference: "callouts" (patched reference), "shim", or "" (table functions only)."""
    if VARIANT == "patched":
        return "callouts"
    return "shim" if os.path.exists(PLAN_HOOK) else ""


def available() -> bool:
    return os.path.exists(LIBDUCKDB) and os.path.exists(LIBGGREF)


class _Column(C.Structure):  # duckdb_column, src/include/duckdb.h
    _fields_ = [("data", C.c_void_p), ("nullmask", C.c_void_p), ("type", C.c_int), ("name", C.c_char_p),
                ("internal_data", C.c_void_p)]


class _Result(C.Structure):  # duckdb_result
    _fields_ = [("column_count", C.c_uint64), ("row_count", C.c_uint64), ("rows_changed", C.c_uint64),
                ("columns", C.POINTER(_Column)), ("error_message", C.c_char_p), ("internal_data", C.c_void_p)]


class RefDuckDB:
    def __init__(self, threads: int | None = None):
        if not available():
            raise RuntimeError("oracle/_ref is not built (make -C oracle ref, needs /root/reference)")
        self.hook = C.CDLL(PLAN_HOOK, mode=C.RTLD_GLOBAL) if rules_route() == "shim" else None
        self.L = C.CDLL(LIBDUCKDB, mode=C.RTLD_GLOBAL)
        self.G = C.CDLL(LIBGGREF)
        self.L.duckdb_value_int64.restype = C.c_int64
        self.L.duckdb_value_int64.argtypes = [C.POINTER(_Result), C.c_uint64, C.c_uint64]
        self.L.duckdb_column_data.restype = C.c_void_p
        self.L.duckdb_column_data.argtypes = [C.POINTER(_Result), C.c_uint64]
        self.L.duckdb_column_type.restype = C.c_int
        self.L.duckdb_column_type.argtypes = [C.POINTER(_Result), C.c_uint64]
        self.L.duckdb_nullmask_data.restype = C.c_void_p
        self.L.duckdb_nullmask_data.argtypes = [C.POINTER(_Result), C.c_uint64]
        self.G.ggref_append_int64_columns.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.POINTER(C.c_int64)),
                                                      C.c_uint64, C.c_char_p, C.c_int]
        self.db, self.con = C.c_void_p(), C.c_void_p()
        assert self.L.duckdb_open(None, C.byref(self.db)) == 0
        assert self.L.duckdb_connect(self.db, C.byref(self.con)) == 0
        self.threads = threads or os.cpu_count() or 1
        self.execute(f"PRAGMA threads={self.threads}")

    def connect(self) -> "RefDuckDB":
        """A second connection to the same database (for concurrent statements)."""
        other = object.__new__(RefDuckDB)
        other.hook, other.L, other.G, other.db, other.threads = self.hook, self.L, self.G, self.db, self.threads
        other.con = C.c_void_p()
        assert self.L.duckdb_connect(self.db, C.byref(other.con)) == 0
        other._owns_db = False
        return other

    def close(self):
        if self.con:
            self.L.duckdb_disconnect(C.byref(self.con))
            if getattr(self, "_owns_db", True):
                self.L.duckdb_close(C.byref(self.db))
            self.con = None

    def execute(self, sql: str) -> np.ndarray:
        """Run sql; return the (all-integer) result as an int64 array [rows, cols]."""
        r = _Result()
        st = self.L.duckdb_query(self.con, sql.encode(), C.byref(r))
        if st != 0:
            msg = r.error_message.decode() if r.error_message else "?"
            self.L.duckdb_destroy_result(C.byref(r))
            raise RuntimeError(f"reference query failed: {msg}")
        out = np.empty((r.row_count, r.column_count), np.int64)
        INT64, INT32 = 5, 4  # DUCKDB_TYPE_BIGINT / DUCKDB_TYPE_INTEGER
        for c in range(r.column_count):
            typ = self.L.duckdb_column_type(C.byref(r), C.c_uint64(c))      # duckdb.h:316
            data = self.L.duckdb_column_data(C.byref(r), C.c_uint64(c))     # duckdb.h:362
            if r.row_count and typ == INT64 and data:
                out[:, c] = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_int64)), shape=(r.row_count,))
            elif r.row_count and typ == INT32 and data:
                out[:, c] = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_int32)), shape=(r.row_count,))
            else:
                for i in range(r.row_count):
                    out[i, c] = self.L.duckdb_value_int64(C.byref(r), c, i)
            # a NULL reads as 0: the data slot under a NULL is whatever the producing operator left there (the two
            # plans of a differential test need not leave the same bytes), duckdb.h:365-372
            mask = self.L.duckdb_nullmask_data(C.byref(r), C.c_uint64(c))
            if r.row_count and mask and typ in (INT64, INT32):
                nulls = np.ctypeslib.as_array(C.cast(mask, C.POINTER(C.c_bool)), shape=(r.row_count,))
                if nulls.any():
                    out[nulls, c] = 0
        self.L.duckdb_destroy_result(C.byref(r))
        return out

    def execute_text(self, sql: str) -> list:
        """Run sql; return every row as a tuple of strings (None for NULL), whatever the column types."""
        r = _Result()
        self.L.duckdb_value_varchar.restype = C.c_void_p
        self.L.duckdb_value_varchar.argtypes = [C.POINTER(_Result), C.c_uint64, C.c_uint64]
        if self.L.duckdb_query(self.con, sql.encode(), C.byref(r)) != 0:
            msg = r.error_message.decode() if r.error_message else "?"
            self.L.duckdb_destroy_result(C.byref(r))
            raise RuntimeError(f"reference query failed: {msg}")
        rows = []
        for i in range(r.row_count):
            row = []
            for c in range(r.column_count):
                p = self.L.duckdb_value_varchar(C.byref(r), c, i)
                row.append(C.string_at(p).decode() if p else None)
                if p:
                    self.L.duckdb_free(C.c_void_p(p))
            rows.append(tuple(row))
        self.L.duckdb_destroy_result(C.byref(r))
        return rows

    def query_text(self, sql: str) -> list:
        """Run sql through the reference's C++ API (oracle/ref_driver.cpp: every column type, DECIMAL and HUGEINT
        included, which its C API of this vintage cannot return); rows as tuples of strings, None for NULL,
        booleans as 1 / 0 — the value conversion of the reference's own sqllogictest runner."""
        out, n, rows, cols = C.c_char_p(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        err = C.create_string_buffer(1024)
        self.G.ggref_query_text.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_int]
        self.G.ggref_free.argtypes = [C.c_void_p]
        raw = C.c_void_p()
        rc = self.G.ggref_query_text(self.con, sql.encode(), C.cast(C.byref(raw), C.POINTER(C.c_char_p)), C.byref(n),
                                     C.byref(rows), C.byref(cols), err, 1024)
        if rc != 0:
            raise RuntimeError(f"reference query failed: {err.value.decode(errors='replace')}")
        data = C.string_at(raw, n.value)
        self.G.ggref_free(raw)
        result = []
        if rows.value:
            for line in data.split(b"\x1e")[:-1]:
                result.append(tuple(None if f == b"\x00" else f.decode(errors="replace") for f in line.split(b"\x1f")))
        return result

    def explain(self, sql: str) -> str:
        """Physical plan of sql as text (EXPLAIN's second column)."""
        r = _Result()
        self.L.duckdb_value_varchar.restype = C.c_void_p
        self.L.duckdb_value_varchar.argtypes = [C.POINTER(_Result), C.c_uint64, C.c_uint64]
        if self.L.duckdb_query(self.con, ("EXPLAIN " + sql).encode(), C.byref(r)) != 0:
            msg = r.error_message.decode() if r.error_message else "?"
            self.L.duckdb_destroy_result(C.byref(r))
            raise RuntimeError(f"reference query failed: {msg}")
        parts = []
        for i in range(r.row_count):
            p = self.L.duckdb_value_varchar(C.byref(r), r.column_count - 1, i)
            parts.append(C.string_at(p).decode())
            self.L.duckdb_free(C.c_void_p(p))
        self.L.duckdb_destroy_result(C.byref(r))
        return "\n".join(parts)

    def load_table(self, name: str, columns: dict):
        """CREATE TABLE name(col BIGINT NOT NULL, ...) and append the given int64 arrays."""
        cols = ", ".join(f"{k} BIGINT NOT NULL" for k in columns)
        self.execute(f"CREATE TABLE {name} ({cols})")
        arrs = [np.ascontiguousarray(v, np.int64) for v in columns.values()]
        n = arrs[0].size
        ptrs = (C.POINTER(C.c_int64) * len(arrs))(*[a.ctypes.data_as(C.POINTER(C.c_int64)) for a in arrs])
        err = C.create_string_buffer(512)
        rc = self.G.ggref_append_int64_columns(self.con, name.encode(), len(arrs), ptrs, n, err, 512)
        if rc != 0:
            raise RuntimeError(f"reference append failed: {err.value.decode()}")

    def load_ldbc(self, vid, src, dst):
        # benchmark/ldbc/schema.sql:72-105 (only the key columns the path touches)
        self.load_table("person", {"p_personid": vid})
        self.load_table("knows", {"k_person1id": src, "k_person2id": dst})

    def timed(self, sql: str, runs: int = 1):
        """(result, best wall seconds) over `runs` executions (benchmark_runner protocol: hot runs)."""
        best, out = None, None
        for _ in range(runs):
            t = time.perf_counter()
            out = self.execute(sql)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        return out, best


# ---- the SQL formulations of the hot path -------------------------------------------
def sql_khop(h: int, select: str = "count(*)", where_extra: str = "") -> str:
    """h-hop walks: person p0, knows k1, person p1, ..., knows kh, person ph (every path vertex a person)."""
    frm = ["person p0"]
    cond = []
    for i in range(1, h + 1):
        frm += [f"knows k{i}", f"person p{i}"]
        cond += [f"p{i-1}.p_personid = k{i}.k_person1id", f"k{i}.k_person2id = p{i}.p_personid"]
    w = " AND ".join(cond) + (f" AND {where_extra}" if where_extra else "")
    return f"SELECT {select} FROM {', '.join(frm)} WHERE {w}"


def sql_khop_rows(h: int, where_extra: str = "") -> str:
    return sql_khop(h, ", ".join(f"p{i}.p_personid" for i in range(h + 1)), where_extra)


def sql_shortest(sources, max_hops: int) -> str:
    """friends / friends_shortest of benchmark/ldbc/queries/bi-10-shortestpath.sql:8-31, seed widened to
    a source list, destination joined with person (gg.h edge semantics)."""
    src = ", ".join(str(int(s)) for s in sources)
    return f"""
WITH RECURSIVE friends(startPerson, hopCount, friend) AS (
    SELECT p_personid, 0, p_personid FROM person WHERE p_personid IN ({src})
  UNION
    SELECT f.startPerson, f.hopCount+1, k.k_person2id
      FROM friends f, knows k, person p
     WHERE f.friend = k.k_person1id AND k.k_person2id = p.p_personid AND f.hopCount < {int(max_hops)}
)
SELECT startPerson, friend, min(hopCount) AS hopCount FROM friends GROUP BY startPerson, friend"""
