"""The reference's own operator test vectors through the substituted engine.

tests/golden/ref_sql_vectors.json holds every statement of the reference's sqllogictest files for the hot path
(test/sql/join/inner/*.test, test/sql/cte/*.test) with the outcome its runner demands (made by
tests/golden/make_sqllogic_vectors.py; comparison rules of test/sqlite/test_sqllogictest.cpp:306-560,880-960 restated
here).  They are replayed inside the compiled reference (oracle/_ref):

  * rules OFF (CPU, no GPU needed): every vector must hold — this pins the reader and the replay, not the product;
  * rules ON (`-m gpu`): the extension is loaded and `PRAGMA enable_gpu_graph` is set on the connection before the
    file's first statement.  Every vector must hold again — what the planner rules decline still runs on the
    reference's operators, what they take over must give the reference's answer — and EXPLAIN records which statements
    got a GG operator; the duplicate-chain join and the joins of four more files must be among them (the cte files
    hold as well, but none of their statements is a shape the operators take — see _assert_substituted).  The same on the
    maintainers' route (GG_REF_VARIANT=patched: call-outs instead of the interposition shim) in a child process.
"""
import hashlib
import json
import os
import subprocess
import sys

import pytest

from oracle import ref_duckdb as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VECTORS = os.path.join(ROOT, "tests", "golden", "ref_sql_vectors.json")
pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")


def _vectors():
    return json.load(open(VECTORS))["files"]


def _same_value(got: str, want: str) -> bool:
    """compare_values of the reference's runner: equal strings, or equal as numbers, or equal as booleans."""
    if got == want:
        return True
    booleans = {"true": 1, "false": 0, "1": 1, "0": 0}
    if got.lower() in booleans and want.lower() in booleans:
        return booleans[got.lower()] == booleans[want.lower()]
    try:
        from decimal import Decimal

        return Decimal(got) == Decimal(want)
    except Exception:
        pass
    try:
        return abs(float(got) - float(want)) <= 1e-9 * max(1.0, abs(float(want)))
    except Exception:
        return False


def _check_query(rec, rows):
    ncols = len(rec["columns"])
    hashed = len(rec["expected"]) == 1 and " values hashing to " in rec["expected"][0]
    if rows and len(rows[0]) != ncols and not hashed:  # (a hash covers the values, however many columns they came in)
        return f"column count {len(rows[0])} != {ncols}"
    ncols = len(rows[0]) if rows else ncols
    values = [("NULL" if v is None else ("(empty)" if v == "" else v)) for row in rows for v in row]
    if rec["sort"] == "rowsort":
        srt = sorted([values[r * ncols:(r + 1) * ncols] for r in range(len(rows))])
        values = [v for row in srt for v in row]
    elif rec["sort"] == "valuesort":
        values = sorted(values)
    expected = rec["expected"]
    if len(expected) == 1 and " values hashing to " in expected[0]:
        n, digest = expected[0].split(" values hashing to ")
        md5 = hashlib.md5("".join(v + "\n" for v in values).encode()).hexdigest()
        return None if (int(n) == len(values) and md5 == digest) else f"hash mismatch: {len(values)} values, {md5}"
    # row-wise if every expected line splits into exactly ncols values, else one value per line
    split = [line.split("\t") for line in expected]
    if expected and all(len(s) == ncols for s in split):
        want = [v for s in split for v in s]
    else:
        want = list(expected)
    if len(want) != len(values):
        return f"{len(values)} values, expected {len(want)}"
    for i, (g, w) in enumerate(zip(values, want)):
        if not _same_value(g, w):
            return f"value {i}: {g!r} != {w!r}"
    return None


def replay(name, records, gpu_rules: bool):
    """Run one file's records in a fresh database; returns (failures, statements taken over by a GG operator)."""
    d = R.RefDuckDB(threads=4)
    failures, taken = [], []
    try:
        if gpu_rules:
            d.execute(f"LOAD '{R.EXTENSION}'")
            d.execute("PRAGMA enable_gpu_graph")
            d.execute("PRAGMA enable_gpu_joins")  # (any single-key inner join over a table scan: GG_KEY_JOIN)
        for rec in records:
            if rec["kind"] == "directive":
                continue
            sql = rec["sql"]
            if rec["kind"] == "statement":
                try:
                    d.query_text(sql)
                    ok = True
                except RuntimeError:
                    ok = False
                if ok != (rec["expect"] == "ok"):
                    failures.append((name, rec["line"], "statement " + rec["expect"], sql[:200]))
                continue
            if gpu_rules and sql.lstrip().lower().startswith(("select", "with")):
                try:
                    if "GG_" in d.explain(sql):
                        taken.append((name, rec["line"], sql))
                except RuntimeError:
                    pass
            try:
                rows = d.query_text(sql)
            except RuntimeError as e:
                failures.append((name, rec["line"], "query failed: " + str(e)[:200], sql[:200]))
                continue
            why = _check_query(rec, rows)
            if why:
                failures.append((name, rec["line"], why, sql[:200]))
    finally:
        d.close()
    return failures, taken


def replay_all(gpu_rules: bool):
    failures, taken, n = [], [], 0
    for name, records in sorted(_vectors().items()):
        f, t = replay(name, records, gpu_rules)
        failures += f
        taken += t
        n += sum(r["kind"] != "directive" for r in records)
    return n, failures, taken


def test_vectors_hold_on_the_reference_itself():
    n, failures, _ = replay_all(False)
    assert n > 600 and not failures, failures[:5]


def _assert_substituted(n, failures, taken):
    assert n > 600 and not failures, failures[:5]
    files = {t[0] for t in taken}
    # the 10 240-long duplicate chain (test_join_duplicates.test:14-24) ran as GG_JOIN_COUNT, and joins of
    # join_cache / test_using_join / test_join_invisible_probe / test_join_perfect_hash as GG_KEY_JOIN (build side sunk
    # into the device index, probe side streamed through it)
    for name in ("test_join_duplicates.test", "join_cache.test", "test_using_join.test", "test_join_invisible_probe.test"):
        assert "test/sql/join/inner/" + name in files, sorted(files)
    assert len(taken) >= 10, taken
    # The cte files hold too (above), but none of their statements has a shape the operators take: the recursions are
    # arithmetic over the working table, and the one hash join in a recursive arm
    # (recursive_cte_complex_pipelines.test:67-81) compares an INTEGER-then-HUGEINT column with a BIGINT one.  That
    # shape — a key join with a base table inside a recursive arm — is covered with the reference's own plan as the
    # oracle by tests/test_duckdb_extension.py::test_generic_key_joins_stream_through_the_device_index.


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(R.EXTENSION), reason="extension not built")
def test_vectors_hold_with_the_planner_rules_on():
    n, failures, taken = replay_all(True)
    _assert_substituted(n, failures, taken)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):  # (a record of what was taken over, for profiles/)
        json.dump({"route": R.rules_route(), "statements": n, "taken_over": [{"file": f, "line": l, "sql": s} for f, l, s in taken]},
                  open(os.path.join(out, f"ref_vectors_taken_{R.rules_route()}.json"), "w"), indent=1)


@pytest.mark.gpu
def test_vectors_hold_on_the_call_out_route():
    patched = os.path.join(ROOT, "oracle", "_ref_patched", "libduckdb.so")
    ext = os.path.join(ROOT, "duckdb_pgq_amd", "callouts", "gg_duckdb.duckdb_extension")
    if not (os.path.exists(patched) and os.path.exists(ext)):
        pytest.skip("oracle/_ref_patched / callouts extension not built")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "gpu",
                        "tests/test_reference_vectors.py::test_vectors_hold_with_the_planner_rules_on"], cwd=ROOT,
                       env=dict(os.environ, GG_REF_VARIANT="patched"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
