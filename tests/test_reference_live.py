"""Where the compiled reference is present (oracle/_ref, built from /root/reference by
`make -C oracle ref`), run it live against the C oracle on a fresh seeded graph.  Skipped on boxes
that only carry the committed fixtures."""
import os

import numpy as np
import pytest

from duckdb_pgq_amd import datagen
from oracle import ref_duckdb as R
from tests.oracle_lib import sort_rows

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def refdb():
    vid, src, dst = datagen.ldbc_knows(1200, 30_000, 0xABCD)
    db = R.RefDuckDB(threads=4)
    db.load_ldbc(vid, src, dst)
    yield db, vid, src, dst
    db.close()


def test_khop_counts_and_rows(orc, refdb):
    db, vid, src, dst = refdb
    rc, c = orc.csr_build(vid, src, dst)
    assert rc == 0
    st = c.khop(1, 3)
    for h in (1, 2, 3):
        assert int(db.execute(R.sql_khop(h))[0, 0]) == st["rows"][h]
    rows = c.khop_rows(1, 2)
    for h in (1, 2):
        assert np.array_equal(sort_rows(db.execute(R.sql_khop_rows(h))), sort_rows(rows[h]))
    # a source sample by rowid, as bench.py's cpu_baseline leg uses
    lim = 100
    got = int(db.execute(R.sql_khop(2, where_extra=f"p0.rowid < {lim}"))[0, 0])
    assert got == c.khop(2, 2, lo=0, hi=lim)["rows"][2]
    c.close()


def test_shortest_path_relation(orc, refdb):
    db, vid, src, dst = refdb
    sources = datagen.pick_sources(vid, 64, 9)
    for max_hops in (0, 1, 4):
        ref = sort_rows(db.execute(R.sql_shortest(sources, max_hops)))
        assert np.array_equal(sort_rows(orc.cte_shortest(vid, src, dst, sources, max_hops)), ref)


def test_connectedsegments_sql_on_reference():
    """The reference itself reproduces its golden rows through our driver (sanity of the driver)."""
    from tests import trainbenchmark as tb

    db = R.RefDuckDB(threads=1)
    t = tb.tables()
    db.load_table("Segment", {"id": t["Segment"][:, 0], "length": t["Segment"][:, 1]})
    db.load_table("connectsTo", {"TrackElement1_id": t["connectsTo"][:, 0], "TrackElement2_id": t["connectsTo"][:, 1]})
    db.load_table("monitoredBy", {"TrackElement_id": t["monitoredBy"][:, 0], "Sensor_id": t["monitoredBy"][:, 1]})
    # the statement is generated (tests/trainbenchmark.py), or read where the reference's own file is present
    path = "/root/reference/benchmark/trainbenchmark/queries/connectedsegments.sql"
    sql = open(path).read() if os.path.exists(path) else tb.connectedsegments_sql()
    assert np.array_equal(sort_rows(db.execute(sql)), sort_rows(tb.CONNECTEDSEGMENTS_GOLDEN))
    db.close()


def test_hoisted_build_variant_of_the_reference_returns_the_same_relation():
    """oracle/hoisted_build.patch keeps the hash table over the edge table across the iterations of
    the recursive CTE instead of rebuilding it per level (physical_recursive_cte.cpp:112-119).  It is only a faster
    baseline: the shortest-path relation must be the stock reference's.  One variant per process (same symbols)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref_hoisted", "libduckdb.so")):
        pytest.skip("oracle/_ref_hoisted not built (make -C oracle ref_hoisted)")
    out = {}
    for variant in ("", "hoisted"):
        env = dict(os.environ, GG_REF_VARIANT=variant)
        r = subprocess.run([sys.executable, os.path.join(root, "oracle", "ref_cte_bench.py"), "1500,40000,9", "24", "4", "2"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[variant or "stock"] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["stock"]["variant"] == "stock" and out["hoisted"]["variant"] == "hoisted"
    assert out["stock"]["rows"] == out["hoisted"]["rows"] > 0
    assert out["stock"]["checksum"] == out["hoisted"]["checksum"]


@pytest.mark.skipif(not os.path.exists(R.EXTENSION), reason="extension not built")
def test_statement_generator_of_the_sql_fuzzer_is_valid_sql_for_the_reference():
    """scripts/fuzz_sql.py --cpu-only: every statement shape the differential fuzzer draws (join chains, friends unions,
    shortest-path CTEs, key joins, with writes in between) parses, binds and runs on the reference's own plans — so a
    failure of the GPU run is never the generator's."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_sql.py"), "--cpu-only", "--writes", "--seed", "9",
                          "--iterations", "150"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fuzz_sql ok: 150 statements" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
