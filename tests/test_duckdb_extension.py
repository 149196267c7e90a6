"""Drop-in test: the compiled reference (oracle/_ref/libduckdb.so) LOADs our extension and runs the GPU
operators next to its own CPU operators on the SAME tables in the SAME database; results must be equal
as sorted relations.  Needs the prebuilt extension (built where /root/reference exists) and a GPU."""
import os
import re

import numpy as np
import pytest

from duckdb_pgq_amd import datagen
from oracle import ref_duckdb as R
from tests.oracle_lib import sort_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = R.EXTENSION  # (the build that goes with the reference variant under test: oracle/ref_duckdb.py)

pytestmark = [
    pytest.mark.gpu,
    pytest.mark.skipif(not (R.available() and os.path.exists(EXT)), reason="reference build / extension not present"),
]

GRAPH = "'person', 'p_personid', 'knows', 'k_person1id', 'k_person2id'"


@pytest.fixture(scope="module")
def db():
    vid, src, dst = datagen.ldbc_knows(1500, 40_000, 0xD0C)
    # a few dangling edge rows and a NULL-free but non-person id
    src = np.concatenate([src, np.array([-5, vid[3]], np.int64)])
    dst = np.concatenate([dst, np.array([vid[2], -6], np.int64)])
    d = R.RefDuckDB(threads=4)
    d.load_ldbc(vid, src, dst)
    d.execute(f"LOAD '{EXT}'")
    yield d, vid
    d.close()


def test_khop_count_matches_reference_joins(db):
    d, _ = db
    got = d.execute(f"SELECT hops, rows FROM gg_khop_count({GRAPH}, 1, 3) ORDER BY hops")
    for hops, rows in got.tolist():
        assert rows == int(d.execute(R.sql_khop(hops))[0, 0])


def test_khop_rows_match_reference_joins(db):
    d, _ = db
    got = d.execute(f"SELECT hops, v0, v1, v2 FROM gg_khop({GRAPH}, 1, 2)")
    one = got[got[:, 0] == 1][:, 1:3]
    two = got[got[:, 0] == 2][:, 1:4]
    assert np.array_equal(sort_rows(one), sort_rows(d.execute(R.sql_khop_rows(1))))
    assert np.array_equal(sort_rows(two), sort_rows(d.execute(R.sql_khop_rows(2))))
    # the function composes with ordinary SQL (aggregation on top of the GPU source)
    agg = d.execute(f"SELECT v0, count(*) FROM gg_khop({GRAPH}, 2, 2) GROUP BY v0 ORDER BY v0")
    two_ref = d.execute(R.sql_khop_rows(2))
    u, c = np.unique(two_ref[:, 0], return_counts=True)
    assert np.array_equal(agg, np.stack([u, c], axis=1))


def test_shortest_path_matches_recursive_cte(db):
    d, vid = db
    sources = datagen.pick_sources(vid, 70, 3)  # > 64: two bit-lane batches
    src_sql = "SELECT p_personid FROM person WHERE p_personid IN (" + ", ".join(str(int(s)) for s in sources) + ")"
    for max_hops in (1, 3):
        got = d.execute(f"SELECT * FROM gg_shortest_path({GRAPH}, '{src_sql}', {max_hops})")
        ref = d.execute(R.sql_shortest(sources, max_hops))
        assert np.array_equal(sort_rows(got), sort_rows(ref))


def test_null_keys_and_int32_columns_follow_join_semantics(db):
    """Nullable INTEGER key columns: NULL never joins (JoinHashTable::PrepareKeys filters NULL keys,
    join_hashtable.cpp:126-148); the sinks skip such rows, so GPU and CPU relations stay equal."""
    d, _ = db
    d.execute("CREATE TABLE v32 (id INTEGER)")
    d.execute("CREATE TABLE e32 (a INTEGER, b INTEGER)")
    d.execute("INSERT INTO v32 VALUES (1), (2), (3), (4), (NULL), (7)")
    d.execute("INSERT INTO e32 VALUES (1,2), (2,3), (NULL,3), (3,NULL), (3,4), (4,1), (2,2), (9,1), (1,9), (NULL,NULL), (2,3)")
    got = d.execute("SELECT v0, v1, v2 FROM gg_khop('v32','id','e32','a','b', 2, 2)")
    ref = d.execute("SELECT p0.id, p1.id, p2.id FROM v32 p0, e32 k1, v32 p1, e32 k2, v32 p2 "
                    "WHERE p0.id = k1.a AND k1.b = p1.id AND p1.id = k2.a AND k2.b = p2.id")
    assert ref.shape[0] > 0 and np.array_equal(sort_rows(got), sort_rows(ref))
    sp = d.execute("SELECT * FROM gg_shortest_path('v32','id','e32','a','b', 'SELECT id FROM v32', 4)")
    cte = d.execute("""WITH RECURSIVE f(s, h, x) AS (SELECT id, 0, id FROM v32 WHERE id IS NOT NULL
        UNION SELECT f.s, f.h+1, k.b FROM f, e32 k, v32 p WHERE f.x = k.a AND k.b = p.id AND f.h < 4)
        SELECT s, x, min(h) FROM f GROUP BY s, x""")
    assert np.array_equal(sort_rows(sp), sort_rows(cte))


def test_connectedsegments_gpu_table_function_vs_reference_query():
    """BASELINE.json configs[4] inside the reference: its own ConnectedSegments query (11 hash joins,
    benchmark/trainbenchmark/queries/connectedsegments.sql) and the GPU operator run on the same tables;
    both must give the golden rows."""
    from tests import trainbenchmark as tb

    d = R.RefDuckDB(threads=4)
    t = tb.tables()
    d.load_table("Segment", {"id": t["Segment"][:, 0], "length": t["Segment"][:, 1]})
    d.load_table("TrackElement", {"id": tb.load("TrackElement")[:, 0]})
    d.load_table("Sensor", {"id": tb.load("Sensor")[:, 0]})
    d.load_table("connectsTo", {"TrackElement1_id": t["connectsTo"][:, 0], "TrackElement2_id": t["connectsTo"][:, 1]})
    d.load_table("monitoredBy", {"TrackElement_id": t["monitoredBy"][:, 0], "Sensor_id": t["monitoredBy"][:, 1]})
    d.execute(f"LOAD '{EXT}'")
    # the reference's query (benchmark/trainbenchmark/queries/connectedsegments.sql:1-25), built by the
    # same generator the planner-rule tests use: 5 connectsTo joins, 6 monitoredBy joins, sensors equal
    from tests.test_plan_rule import connectedsegments_sql

    cpu = d.execute(connectedsegments_sql(5))
    gpu = d.execute("""SELECT * FROM gg_same_neighbour_paths(
        'SELECT id FROM TrackElement UNION ALL SELECT id FROM Sensor', 'SELECT id FROM Segment',
        'connectsTo', 'TrackElement1_id', 'TrackElement2_id', 'monitoredBy', 'TrackElement_id', 'Sensor_id', 5)""")
    d.close()
    assert np.array_equal(sort_rows(cpu), sort_rows(tb.CONNECTEDSEGMENTS_GOLDEN))
    assert np.array_equal(sort_rows(gpu), sort_rows(tb.CONNECTEDSEGMENTS_GOLDEN))


# ---- plan-level substitution (gg_plan_rule.cpp): the reference's OWN SQL, planned onto the GPU ---------
def _both_plans(d, sql):
    """Run sql with the planner rules off (the reference's hash joins) and on (GPU expansion)."""
    d.execute("PRAGMA disable_gpu_graph")
    assert "GG_" not in d.explain(sql)
    cpu = d.execute(sql)
    d.execute("PRAGMA enable_gpu_graph")
    assert "GG_PATH" in d.explain(sql) or "GG_WALK_ENDPOINTS" in d.explain(sql), d.explain(sql)
    gpu = d.execute(sql)
    d.execute("PRAGMA disable_gpu_graph")
    return cpu, gpu


def _chain(h, select):
    frm = ", ".join(f"knows k{i}" for i in range(1, h + 1))
    cond = " AND ".join(f"k{i}.k_person2id = k{i+1}.k_person1id" for i in range(1, h))
    return f"SELECT {select} FROM {frm} WHERE {cond}"


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_plan_rule_join_chains_give_the_reference_result(db):
    d, vid = db
    # a keyed copy of the vertex table: the vertex-validated rule needs a declared-unique key
    d.execute("CREATE TABLE person_pk (p_personid BIGINT PRIMARY KEY)")
    d.execute("INSERT INTO person_pk SELECT p_personid FROM person")
    keyed = lambda sql: sql.replace("person ", "person_pk ")
    if R.rules_route() == "shim":  # the shim also tells whether the extension registered its rules with it
        lib = d.hook
        assert lib is not None and lib.gg_plan_hook_registered(0) == 1 and lib.gg_plan_hook_registered(1) == 1
    else:  # a reference with the call-outs of oracle/callout.patch: registered there, no shim in the process
        import ctypes
        ext = ctypes.CDLL(EXT)
        assert d.hook is None and ext.gg_plan_rules_by_callout() == 1

    # edge-only chains: count(*) (aggregate rule) and materialised endpoints (join rule)
    for h in (2, 3):
        cpu, gpu = _both_plans(d, _chain(h, "count(*)"))
        assert cpu[0, 0] > 0 and np.array_equal(cpu, gpu)
    cpu, gpu = _both_plans(d, _chain(2, "k1.k_person1id, k1.k_person2id, k2.k_person2id"))
    assert np.array_equal(sort_rows(cpu), sort_rows(gpu))
    cpu, gpu = _both_plans(d, _chain(2, "k2.k_person2id, k1.k_person1id, k2.k_person1id, k2.k_person2id"))
    assert np.array_equal(sort_rows(cpu), sort_rows(gpu))
    # dangling endpoints (-5, -6 in the fixture) are vertices of an edge-only chain: they must show up
    assert (cpu == -5).any()

    # vertex-validated chains: the oracle's own SQL formulation (person joins on every position)
    for h in (1, 2):
        cpu, gpu = _both_plans(d, keyed(R.sql_khop(h)))
        assert cpu[0, 0] > 0 and np.array_equal(cpu, gpu)
    cpu, gpu = _both_plans(d, keyed(R.sql_khop_rows(2)))
    assert np.array_equal(sort_rows(cpu), sort_rows(gpu))
    assert not (cpu < 0).any()

    # single source pinned by a constant (interactive-complex-3.sql:9-11), ordinary SQL on top
    s = int(vid[7])
    sql = _chain(2, "DISTINCT k2.k_person2id") + f" AND k1.k_person1id = {s} ORDER BY 1"
    cpu, gpu = _both_plans(d, sql)
    assert cpu.shape[0] > 0 and np.array_equal(cpu, gpu)
    absent = int(np.sort(vid)[5]) + 1  # inside the column's min/max statistics, but nobody's id
    assert absent not in set(vid.tolist())
    cpu, gpu = _both_plans(d, _chain(2, "count(*)") + f" AND k1.k_person1id = {absent}")
    assert cpu[0, 0] == 0 and np.array_equal(cpu, gpu)

    # grouped aggregate above a substituted join
    sql = _chain(2, "k1.k_person1id, count(*)") + " GROUP BY k1.k_person1id ORDER BY 1"
    cpu, gpu = _both_plans(d, sql)
    assert np.array_equal(cpu, gpu)


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_plan_rule_sees_table_changes_between_executions(db):
    """The scan opens at execution time: a second run of the same SQL reflects rows inserted meanwhile."""
    d, _ = db
    d.execute("CREATE TABLE e (a BIGINT NOT NULL, b BIGINT NOT NULL)")
    d.execute("INSERT INTO e VALUES (1, 2), (2, 3)")
    sql = "SELECT count(*) FROM e k1, e k2 WHERE k1.b = k2.a"
    d.execute("PRAGMA enable_gpu_graph")
    assert "GG_PATH_COUNT" in d.explain(sql)
    assert int(d.execute(sql)[0, 0]) == 1
    d.execute("INSERT INTO e VALUES (3, 4), (3, 5)")
    assert int(d.execute(sql)[0, 0]) == 3
    d.execute("PRAGMA disable_gpu_graph")
    assert int(d.execute(sql)[0, 0]) == 3


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_plan_rule_friends_cte_gives_the_reference_result(db):
    """bi-10's friends / friends_shortest (recursive CTE + min) with the rules off (PhysicalRecursiveCTE +
    hash aggregate) and on (64-lane BFS): same relation."""
    d, vid = db
    d.execute("CREATE TABLE IF NOT EXISTS person_pk (p_personid BIGINT PRIMARY KEY)")
    if int(d.execute("SELECT count(*) FROM person_pk")[0, 0]) == 0:
        d.execute("INSERT INTO person_pk SELECT p_personid FROM person")

    def both(sql):
        d.execute("PRAGMA disable_gpu_graph")
        assert "REC_CTE" in d.explain(sql)
        cpu = d.execute(sql)
        d.execute("PRAGMA enable_gpu_graph")
        assert "GG_SHORTEST_PATH_BFS" in d.explain(sql), d.explain(sql)
        assert "GG_EDGE_SINK" in d.explain(sql)  # tables through pipeline sinks; GG_NO_PIPELINE_SINKS: from the scan's init
        gpu = d.execute(sql)
        os.environ["GG_NO_PIPELINE_SINKS"] = "1"
        try:
            assert "GG_EDGE_SINK" not in d.explain(sql) and "GG_SHORTEST_PATH_BFS" in d.explain(sql)
            assert np.array_equal(sort_rows(d.execute(sql)), sort_rows(gpu))
        finally:
            del os.environ["GG_NO_PIPELINE_SINKS"]
        d.execute("PRAGMA disable_gpu_graph")
        return sort_rows(cpu), sort_rows(gpu)

    s = int(vid[11])
    # the literal text of benchmark/ldbc/queries/bi-10-shortestpath.sql:8-31 (edge-only step, CASE expression)
    bi10 = f"""WITH RECURSIVE friends(startPerson, hopCount, friend) AS (
        SELECT p_personid, 0, p_personid FROM person WHERE 1=1 AND p_personid = {s}
      UNION
        SELECT f.startPerson, f.hopCount+1,
               CASE WHEN f.friend = k.k_person1id then k.k_person2id ELSE k.k_person1id END
          FROM friends f, knows k WHERE 1=1 AND f.friend = k.k_person1id AND f.hopCount < 3)
      , friends_shortest AS (
        SELECT startPerson, min(hopCount) AS hopCount, friend FROM friends GROUP BY startPerson, friend)
    SELECT startPerson, hopCount, friend FROM friends_shortest"""
    cpu, gpu = both(bi10)
    assert cpu.shape[0] > 100 and np.array_equal(cpu, gpu)
    # dangling destination ids (-6 in the fixture) are reachable in the edge-only form; a seed without any
    # edge still yields its own (s, 0, s) row
    d.execute("CREATE TABLE loner (p_personid BIGINT NOT NULL)")
    d.execute("INSERT INTO loner VALUES (987654321)")
    cpu, gpu = both(bi10.replace("FROM person WHERE 1=1 AND p_personid = " + str(s), "FROM loner"))
    assert cpu.tolist() == [[987654321, 0, 987654321]] and np.array_equal(cpu, gpu)

    # the oracle's validated formulation, > 64 seeds (two bit-lane batches), then every person as a seed
    sources = datagen.pick_sources(vid, 70, 5)
    keyed = lambda q: re.sub(r"\bperson\b", "person_pk", q)
    sql = keyed(R.sql_shortest(sources, 3))
    cpu, gpu = both(sql)
    assert np.array_equal(cpu, gpu) and not (cpu[:, 1] < 0).any()
    every = keyed(R.sql_shortest([0], 2)).replace("WHERE p_personid IN (0)", "")
    cpu, gpu = both(every)
    assert cpu.shape[0] > vid.size and np.array_equal(cpu, gpu)


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_plan_rule_connectedsegments_query_text():
    """BASELINE.json configs[4]: the reference's own ConnectedSegments text (11 hash joins), planned by the
    reference and by the same-neighbour rule (one GPU operator): both give the golden rows at SF1 and the
    same relation on a 16-fold replica; shorter walks too."""
    from tests import trainbenchmark as tb
    from tests.test_plan_rule import connectedsegments_sql

    def both(d, sql):
        d.execute("PRAGMA disable_gpu_graph")
        cpu = d.execute(sql)
        d.execute("PRAGMA enable_gpu_graph")
        assert "GG_SAME_NEIGHBOUR_WALKS" in d.explain(sql) and "HASH_JOIN" not in d.explain(sql)
        # the plan's three table passes (endpoints of the path table, the filter table, the path table's CSR) are
        # pipeline sinks over the reference's own table scans, like the other rules' build sides
        assert "GG_EDGE_SINK" in d.explain(sql)
        gpu = d.execute(sql)
        again = d.execute(sql)
        assert np.array_equal(sort_rows(gpu), sort_rows(again))
        d.execute("PRAGMA disable_gpu_graph")
        return sort_rows(cpu), sort_rows(gpu)

    for copies in (1, 16):
        t = tb.tables()
        if copies > 1:
            rep = datagen.replicate_tables({"Segment": t["Segment"][:, :1], "connectsTo": t["connectsTo"],
                                            "monitoredBy": t["monitoredBy"]}, copies)
            t = {"Segment": np.hstack([rep["Segment"].reshape(-1, 1), np.ones((rep["Segment"].size, 1), np.int64)]),
                 "connectsTo": rep["connectsTo"], "monitoredBy": rep["monitoredBy"]}
        d = R.RefDuckDB(threads=4)
        d.execute("CREATE TABLE Segment (id int NOT NULL, length int NOT NULL DEFAULT 1, PRIMARY KEY (id))")
        d.execute("CREATE TABLE connectsTo (TrackElement1_id int NOT NULL, TrackElement2_id int NOT NULL)")
        d.execute("CREATE TABLE monitoredBy (TrackElement_id int NOT NULL, Sensor_id int NOT NULL)")
        for name, cols in (("Segment", "id, length"), ("connectsTo", "TrackElement1_id, TrackElement2_id"),
                           ("monitoredBy", "TrackElement_id, Sensor_id")):
            d.load_table("stage_" + name, {c.strip(): t[name][:, i] for i, c in enumerate(cols.split(","))})
            d.execute(f"INSERT INTO {name} SELECT {cols} FROM stage_{name}")  # BIGINT staging -> INT columns
        d.execute(f"LOAD '{EXT}'")
        cpu, gpu = both(d, connectedsegments_sql())
        assert np.array_equal(cpu, gpu) and cpu.shape[0] == 4 * copies
        if copies == 1:
            assert np.array_equal(gpu, sort_rows(tb.CONNECTEDSEGMENTS_GOLDEN))
        for hops in (1, 3):
            cpu, gpu = both(d, connectedsegments_sql(hops))
            assert cpu.shape[0] > 0 and np.array_equal(cpu, gpu)
        d.close()


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_plan_rule_prepared_statements_and_concurrent_connections(db):
    import threading

    d, _ = db
    d.execute("CREATE TABLE pe (a BIGINT NOT NULL, b BIGINT NOT NULL)")
    d.execute("INSERT INTO pe SELECT k_person1id, k_person2id FROM knows LIMIT 20000")
    sql = "SELECT count(*) FROM pe k1, pe k2 WHERE k1.b = k2.a"
    expect = int(d.execute(sql)[0, 0])  # rules off: the reference's hash join
    d.execute("PRAGMA enable_gpu_graph")
    # planned once (with the rule), executed many times: every execution re-reads the table
    d.execute("PREPARE walks AS " + sql)
    d.execute("PRAGMA disable_gpu_graph")  # the prepared plan keeps its GPU operator
    assert int(d.execute("EXECUTE walks")[0, 0]) == expect
    d.execute("INSERT INTO pe SELECT k_person1id, k_person2id FROM knows LIMIT 5000 OFFSET 20000")
    more = int(d.execute(sql)[0, 0])
    assert more > expect and int(d.execute("EXECUTE walks")[0, 0]) == more

    # statements on several connections at once: each takes its own device context from the pool
    d.execute("PRAGMA enable_gpu_graph")
    results, errs = [], []

    def work():
        try:
            c = d.connect()
            c.execute("PRAGMA enable_gpu_graph")  # per connection, like the reference's own pragmas
            assert "GG_" in c.explain(sql)
            for _ in range(4):
                results.append(int(c.execute(sql)[0, 0]))
            c.close()
        except Exception as ex:  # pragma: no cover
            errs.append(ex)

    th = [threading.Thread(target=work) for _ in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    d.execute("PRAGMA disable_gpu_graph")
    assert not errs and results == [more] * 12

    # the prepared plan depends on the table: dropping it must not leave a dangling plan that runs
    try:
        d.execute("DROP TABLE pe")
        dropped = True
    except RuntimeError:
        dropped = False  # the reference refuses the drop while a prepared statement depends on it
    if dropped:
        with pytest.raises(RuntimeError):
            d.execute("EXECUTE walks")


def test_oversized_results_are_produced_part_by_part(db, monkeypatch):
    """The reference streams a join result of any size; the GPU operator materialises walks in HBM, so
    beyond a device-memory budget it expands the sources part by part.  With a 1 MiB budget the 2-hop
    result of the fixture splits into dozens of parts — same relation, all sources and a source list."""
    d, vid = db
    d.execute("PRAGMA disable_gpu_graph")
    whole = d.execute(f"SELECT v0, v1, v2 FROM gg_khop({GRAPH}, 2, 2)")
    assert whole.shape[0] * 24 > 20 * (1 << 20)
    monkeypatch.setenv("GG_RESULT_BUDGET_MB", "1")
    parts = d.execute(f"SELECT v0, v1, v2 FROM gg_khop({GRAPH}, 2, 2)")
    mixed = d.execute(f"SELECT hops, v0, v1, v2 FROM gg_khop({GRAPH}, 1, 2)")
    monkeypatch.delenv("GG_RESULT_BUDGET_MB")
    assert np.array_equal(sort_rows(parts), sort_rows(whole))
    assert np.array_equal(sort_rows(mixed[mixed[:, 0] == 2][:, 1:]), sort_rows(whole))
    assert np.array_equal(sort_rows(mixed[mixed[:, 0] == 1][:, 1:3]), sort_rows(d.execute(R.sql_khop_rows(1))))
    if bool(R.rules_route()):  # the substituted join takes the same route
        sql = "SELECT k1.k_person1id, k2.k_person1id, k2.k_person2id FROM knows k1, knows k2 " \
              "WHERE k1.k_person2id = k2.k_person1id"
        cpu = d.execute(sql)
        d.execute("PRAGMA enable_gpu_graph")
        monkeypatch.setenv("GG_RESULT_BUDGET_MB", "1")
        gpu = d.execute(sql)
        monkeypatch.delenv("GG_RESULT_BUDGET_MB")
        d.execute("PRAGMA disable_gpu_graph")
        assert np.array_equal(sort_rows(cpu), sort_rows(gpu))


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_plan_rule_predicates_on_walk_positions(db):
    """Predicates on key columns other than the pinned source stay as a filter above the GPU scan — the
    friends-of-friends branch of interactive-complex-3.sql:9-11 (`k1.src = C ... AND k2.dst <> X`), ranges,
    ORs, a pinned far end."""
    d, vid = db
    s, x = int(vid[7]), int(vid[100])
    two = int(d.execute(f"SELECT k2.k_person2id FROM knows k1, knows k2 WHERE k1.k_person1id = {s} "
                        "AND k1.k_person2id = k2.k_person1id LIMIT 1")[0, 0])
    lo, hi = int(np.sort(vid)[200]), int(np.sort(vid)[900])
    cases = [
        _chain(2, "k2.k_person2id") + f" AND k1.k_person1id = {s} AND k2.k_person2id <> {two}",
        _chain(2, "k1.k_person1id, k2.k_person2id") + f" AND k2.k_person1id > {lo} AND k2.k_person1id <= {hi}",
        _chain(2, "count(*)") + f" AND (k1.k_person2id = {x} OR k1.k_person2id = {s})",
        _chain(3, "k1.k_person1id, k2.k_person1id, k3.k_person1id, k3.k_person2id") +
        f" AND k1.k_person1id = {s} AND k3.k_person2id = {s}",
        _chain(2, "count(*)") + f" AND k1.k_person1id <> {s} AND k2.k_person2id < {hi}",
    ]
    for sql in cases:
        cpu, gpu = _both_plans(d, sql)
        assert cpu.shape[0] > 0 and np.array_equal(sort_rows(cpu), sort_rows(gpu)), sql
    # the friends + friends-of-friends derived table of interactive-complex-3.sql:3-12, verbatim shape: both
    # branches and the dedupe of the UNION are one device operator (GG_WALK_ENDPOINTS)
    ic3 = (f"select k_person2id from knows where k_person1id = {s} union "
           f"select k2.k_person2id from knows k1, knows k2 where k1.k_person1id = {s} "
           f"and k1.k_person2id = k2.k_person1id and k2.k_person2id <> {s}")
    cpu, gpu = _both_plans(d, ic3)
    assert cpu.shape[0] > 10 and np.array_equal(sort_rows(cpu), sort_rows(gpu))


def test_table_functions_accept_views_and_reject_unknown_names(db):
    """Base tables are read straight from storage; anything else named in a gg function (here a view) goes
    through a statement on a side connection; unknown names surface as the reference's own binder error."""
    d, vid = db
    d.execute("PRAGMA disable_gpu_graph")
    cut = int(np.sort(vid)[700])
    d.execute(f"CREATE VIEW knows_low AS SELECT k_person1id AS a, k_person2id AS b FROM knows WHERE k_person1id < {cut}")
    d.execute(f"CREATE VIEW person_all AS SELECT p_personid AS id FROM person")
    got = d.execute("SELECT hops, rows FROM gg_khop_count('person_all', 'id', 'knows_low', 'a', 'b', 1, 2) ORDER BY hops")
    one = int(d.execute("SELECT count(*) FROM person_all p0, knows_low k, person_all p1 WHERE p0.id = k.a AND k.b = p1.id")[0, 0])
    two = int(d.execute("SELECT count(*) FROM person_all p0, knows_low k1, person_all p1, knows_low k2, person_all p2 "
                        "WHERE p0.id = k1.a AND k1.b = p1.id AND p1.id = k2.a AND k2.b = p2.id")[0, 0])
    assert got.tolist() == [[1, one], [2, two]] and one > 0
    with pytest.raises(RuntimeError, match="(?i)knows_missing|does not exist|not found"):
        d.execute("SELECT * FROM gg_khop_count('person', 'p_personid', 'knows_missing', 'a', 'b', 1, 1)")
    with pytest.raises(RuntimeError):
        d.execute("SELECT * FROM gg_khop_count('person', 'no_such_column', 'knows', 'k_person1id', 'k_person2id', 1, 1)")
    # the session is still usable after the errors
    assert int(d.execute("SELECT rows FROM gg_khop_count('person', 'p_personid', 'knows', 'k_person1id', 'k_person2id', 1, 1)")[0, 0]) > 0


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_pinned_graphs_are_reused_and_dropped_when_rows_are_appended(db):
    """gg_graph_pin builds a graph once and keeps it on the device; statements that need exactly that graph
    skip ingest and build.  It is a snapshot: appending rows makes the row count differ, the pin is dropped
    and the statement reads the table again."""
    import time

    d, vid = db
    d.execute("CREATE TABLE pk (a BIGINT NOT NULL, b BIGINT NOT NULL)")
    d.execute("INSERT INTO pk SELECT k_person1id, k_person2id FROM knows")
    sql = "SELECT count(*) FROM pk k1, pk k2 WHERE k1.b = k2.a"
    expect = int(d.execute(sql)[0, 0])
    one = "SELECT k2.b FROM pk k1, pk k2 WHERE k1.b = k2.a AND k1.a = %d" % int(vid[5])
    expect_one = sort_rows(d.execute(one))
    d.execute("PRAGMA enable_gpu_graph")
    d.execute("PRAGMA gg_use_pinned_graphs")  # pinned graphs are opt-in, per connection
    assert int(d.execute(sql)[0, 0]) == expect
    pinned = d.execute("SELECT vertices, edges FROM gg_graph_pin('', '', 'pk', 'a', 'b')")
    assert pinned[0, 1] == int(d.execute("SELECT count(*) FROM pk")[0, 0])
    t = time.perf_counter()
    for _ in range(5):
        assert int(d.execute(sql)[0, 0]) == expect
        assert np.array_equal(sort_rows(d.execute(one)), expect_one)
    with_pin = (time.perf_counter() - t) / 5
    # concurrent statements on one pinned graph
    import threading
    out, errs = [], []

    def work():
        try:
            c = d.connect()
            c.execute("PRAGMA enable_gpu_graph")  # the switches belong to a connection
            c.execute("PRAGMA gg_use_pinned_graphs")
            for _ in range(3):
                out.append(int(c.execute(sql)[0, 0]))
                out.append(sort_rows(c.execute(one)).shape[0])
            c.close()
        except Exception as ex:  # pragma: no cover
            errs.append(ex)

    th = [threading.Thread(target=work) for _ in range(3)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs and sorted(set(out)) == sorted({expect, expect_one.shape[0]})
    # rows appended: the pin no longer matches the table and is dropped
    d.execute("INSERT INTO pk VALUES (%d, %d)" % (int(vid[5]), int(vid[6])))
    d.execute("PRAGMA disable_gpu_graph")
    more = int(d.execute(sql)[0, 0])
    d.execute("PRAGMA enable_gpu_graph")
    assert more > expect and int(d.execute(sql)[0, 0]) == more
    # the vertex-validated form is a different graph: pin it too, through the table function signature
    d.execute(f"SELECT * FROM gg_graph_pin({GRAPH})")
    got = d.execute(f"SELECT hops, rows FROM gg_khop_count({GRAPH}, 1, 2) ORDER BY hops")
    d.execute("PRAGMA disable_gpu_graph")
    assert got[:, 1].tolist() == [int(d.execute(R.sql_khop(h))[0, 0]) for h in (1, 2)]
    assert int(d.execute("SELECT * FROM gg_graph_unpin()")[0, 0]) >= 1
    d.execute("PRAGMA gg_ignore_pinned_graphs")
    assert with_pin < 1.0


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_pinned_graphs_are_opt_in_and_never_outlive_a_write(db):
    """A pinned graph is a snapshot.  Nothing uses one unless the connection said PRAGMA gg_use_pinned_graphs; an
    INSERT, UPDATE or DELETE planned on a pinned table drops the pin (also when the transaction rolls back); a
    transaction with changes of its own neither pins nor uses pins.  Whatever happens, a substituted plan
    returns what the reference's own plan returns on the same tables."""
    d, vid = db
    d.execute("CREATE TABLE pw (a BIGINT NOT NULL, b BIGINT NOT NULL)")
    d.execute("INSERT INTO pw SELECT k_person1id, k_person2id FROM knows WHERE k_person1id >= 0 AND k_person2id >= 0")
    sql = "SELECT count(*), sum(k2.b) FROM pw k1, pw k2 WHERE k1.b = k2.a"

    def cpu():
        d.execute("PRAGMA disable_gpu_graph")
        out = d.execute(sql)
        d.execute("PRAGMA enable_gpu_graph")
        return out

    def pins():
        return int(d.execute("SELECT * FROM gg_graph_pins()")[0, 0])

    pin = "SELECT * FROM gg_graph_pin('', '', 'pw', 'a', 'b')"
    d.execute("PRAGMA enable_gpu_graph")
    d.execute("SELECT * FROM gg_graph_unpin()")
    try:
        # 1. not opted in: the pin exists but the statement reads the table — an UPDATE shows at once
        d.execute(pin)
        assert pins() == 1
        before = cpu()
        assert np.array_equal(d.execute(sql), before)
        d.execute("UPDATE pw SET b = %d WHERE a = %d" % (int(vid[7]), int(vid[9])))
        assert pins() == 0  # ... and the write dropped the pin anyway
        after = cpu()
        assert not np.array_equal(after, before) and np.array_equal(d.execute(sql), after)
        # 2. opted in: UPDATE, DELETE and INSERT each drop the pin, the next statement sees the change
        d.execute("PRAGMA gg_use_pinned_graphs")
        for change in ("UPDATE pw SET b = %d WHERE a = %d" % (int(vid[11]), int(vid[12])),
                       "DELETE FROM pw WHERE a = %d" % int(vid[13]),
                       "INSERT INTO pw VALUES (%d, %d)" % (int(vid[14]), int(vid[15]))):
            d.execute(pin)
            assert pins() == 1 and np.array_equal(d.execute(sql), cpu()) and pins() == 1
            d.execute(change)
            assert pins() == 0
            assert np.array_equal(d.execute(sql), cpu())
        # 3. a write that is rolled back still cost the pin; the result is the table's again
        d.execute(pin)
        keep = cpu()
        d.execute("BEGIN TRANSACTION")
        d.execute("DELETE FROM pw WHERE a = %d" % int(vid[20]))
        assert pins() == 0
        assert np.array_equal(d.execute(sql), cpu())  # inside the transaction: its own deletes are seen
        d.execute("ROLLBACK")
        assert np.array_equal(d.execute(sql), keep) and np.array_equal(cpu(), keep)
        # 4. a transaction with changes of its own does not pin ...
        d.execute("CREATE TABLE other (x BIGINT)")
        d.execute("BEGIN TRANSACTION")
        d.execute("INSERT INTO other VALUES (1)")
        with pytest.raises(Exception, match="uncommitted changes"):
            d.execute(pin)
        d.execute("ROLLBACK")
        # ... and does not use a pin made before it started
        d.execute(pin)
        d.execute("BEGIN TRANSACTION")
        d.execute("INSERT INTO other VALUES (2)")
        assert pins() == 1 and np.array_equal(d.execute(sql), keep)
        d.execute("ROLLBACK")
        # 4b. statements PREPARED before the pin and executed after it never pass the planner again: the pins go when
        # the executor builds the pipelines of a plan with an INSERT / DELETE / UPDATE operator (gg_pipeline.cpp)
        d.execute("PREPARE del_pw AS DELETE FROM pw WHERE a = %d" % int(vid[21]))
        d.execute("PREPARE upd_pw AS UPDATE pw SET b = %d WHERE a = %d" % (int(vid[22]), int(vid[23])))
        d.execute("PREPARE ins_pw AS INSERT INTO pw VALUES (%d, %d)" % (int(vid[24]), int(vid[25])))
        for stmt in ("del_pw", "upd_pw", "ins_pw"):
            d.execute(pin)
            assert pins() == 1 and np.array_equal(d.execute(sql), cpu()) and pins() == 1
            d.execute("EXECUTE " + stmt)
            assert pins() == 0, stmt
            assert np.array_equal(d.execute(sql), cpu()), stmt
        for stmt in ("del_pw", "upd_pw", "ins_pw"):
            d.execute("DEALLOCATE " + stmt)
        # 5. another connection has its own switches: no pragma, no substituted plan
        c = d.connect()
        assert "GG_" not in c.explain(sql)
        c.execute("PRAGMA enable_gpu_graph")
        assert "GG_" in c.explain(sql) and "GG_" in d.explain(sql)
        c.execute("PRAGMA disable_gpu_graph")
        assert "GG_" not in c.explain(sql) and "GG_" in d.explain(sql)
        c.close()
    finally:
        d.execute("SELECT * FROM gg_graph_unpin()")
        d.execute("PRAGMA gg_ignore_pinned_graphs")
        d.execute("PRAGMA disable_gpu_graph")


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_statements_shaped_like_the_ldbc_friends_queries_give_the_reference_result():
    """Five statements with the shape of the reference's interactive-complex-3/5/6/9/11 (tests/ldbc_shapes.py:
    friends UNION friends-of-friends of one person, joined with person / place / message / forum / organisation /
    tag columns, aggregated, ordered, limited) over a populated database: with the planner rules on, the
    friends UNION friends-of-friends table — both branches and the dedupe — runs as GG_WALK_ENDPOINTS on the GPU,
    everything above it — the joins that fetch the projected person columns, aggregates, ORDER BY, LIMIT — stays
    with the reference's operators, and every statement returns exactly the rows the reference's own plan returns,
    in order."""
    from tests import ldbc_shapes
    d = R.RefDuckDB(threads=4)
    ldbc_shapes.populate(d)
    d.execute(f"LOAD '{EXT}'")
    try:
        for name, sql in ldbc_shapes.statements().items():
            d.execute("PRAGMA disable_gpu_graph")
            assert "GG_" not in d.explain(sql)
            cpu = d.execute_text(sql)
            d.execute("PRAGMA enable_gpu_graph")
            assert "GG_WALK_ENDPOINTS" in d.explain(sql), name
            gpu = d.execute_text(sql)
            assert len(cpu) > 0 and gpu == cpu, name
        # inside a transaction that changed the edge table the statement still sees its own rows
        d.execute("PRAGMA enable_gpu_graph")
        d.execute("BEGIN")
        d.execute(f"INSERT INTO knows VALUES ('2012-01-01 00:00:00', {ldbc_shapes.PERSON_B}, 424242), "
                  f"('2012-01-01 00:00:00', 424242, {ldbc_shapes.PERSON_A})")
        sql = f"select count(*) from {ldbc_shapes.friends(ldbc_shapes.PERSON_B, ldbc_shapes.PERSON_B)}"
        with_rows = d.execute(sql)[0, 0]
        d.execute("ROLLBACK")
        without = d.execute(sql)[0, 0]
        d.execute("PRAGMA disable_gpu_graph")
        assert without == d.execute(sql)[0, 0] and with_rows >= without + 1
    finally:
        d.execute("PRAGMA disable_gpu_graph")
        d.close()


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_sinks_run_as_pipelines_of_the_references_executor(db, monkeypatch):
    """The tables of a substituted plan flow into the device graph through pipeline sinks that the reference's
    executor schedules (Executor::BuildPipelines case in gg_pipeline.cpp): its profiler reports the sink and the
    table scan below it as operators of their own, prepared statements rebuild the graph on every execution, and
    the scan-function route (GG_NO_PIPELINE_SINKS) returns the same rows."""
    d, vid = db
    s = int(vid[11])
    sql = _chain(2, "count(*)")
    rows = _chain(2, "k2.k_person2id") + f" AND k1.k_person1id = {s}"
    d.execute("PRAGMA disable_gpu_graph")
    want_count, want_rows = d.execute(sql), d.execute(rows)
    d.execute("PRAGMA enable_gpu_graph")
    try:
        assert "GG_EDGE_SINK" in d.explain(sql)
        assert np.array_equal(d.execute(sql), want_count)
        assert np.array_equal(sort_rows(d.execute(rows)), sort_rows(want_rows))
        profile = "\n".join(" ".join(str(c) for c in r) for r in d.execute_text("EXPLAIN ANALYZE " + sql))
        assert "GG_EDGE_SINK" in profile and "SEQ_SCAN" in profile and "GG_PATH_COUNT" in profile
        d.execute("PREPARE two_hop AS " + sql)
        for _ in range(3):
            assert np.array_equal(d.execute("EXECUTE two_hop"), want_count)
        d.execute("DEALLOCATE two_hop")
        monkeypatch.setenv("GG_NO_PIPELINE_SINKS", "1")
        assert "GG_EDGE_SINK" not in d.explain(sql) and "GG_PATH_COUNT" in d.explain(sql)
        assert np.array_equal(d.execute(sql), want_count)
        assert np.array_equal(sort_rows(d.execute(rows)), sort_rows(want_rows))
    finally:
        d.execute("PRAGMA disable_gpu_graph")


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_pipeline_sinks_inside_a_plan_with_a_recursive_cte(db):
    """A join chain taken over by the join rule inside a statement that also runs a recursive CTE: the chain's
    sinks get their pipelines next to the CTE's own (which the executor resets and re-runs per iteration), and the
    statement returns the reference's rows — whether the chain feeds the CTE's seed or is joined with its result."""
    d, vid = db
    s = int(vid[3])
    seed_chain = ("WITH RECURSIVE reach(x, hop) AS ("
                  f"SELECT k2.k_person2id, 2 FROM knows k1, knows k2 WHERE k1.k_person1id = {s} AND k1.k_person2id = k2.k_person1id "
                  "UNION SELECT k.k_person2id, r.hop + 1 FROM reach r, knows k WHERE r.x = k.k_person1id AND r.hop < 3) "
                  "SELECT count(*), sum(x % 1000), max(hop) FROM reach")
    beside = ("WITH RECURSIVE up(n) AS (SELECT 1 UNION SELECT n + 1 FROM up WHERE n < 4) "
              "SELECT up.n, c.walks FROM up, (SELECT count(*) AS walks FROM knows k1, knows k2 "
              "WHERE k1.k_person2id = k2.k_person1id) c ORDER BY up.n")
    for sql in (seed_chain, beside):
        d.execute("PRAGMA disable_gpu_graph")
        cpu = d.execute(sql)
        d.execute("PRAGMA enable_gpu_graph")
        try:
            plan = d.explain(sql)
            gpu = d.execute(sql)
            again = d.execute(sql)
        finally:
            d.execute("PRAGMA disable_gpu_graph")
        assert "GG_PATH" in plan and "GG_EDGE_SINK" in plan, plan
        assert cpu.shape[0] > 0 and np.array_equal(cpu, gpu) and np.array_equal(cpu, again), sql


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_a_substituted_scan_in_the_recursive_arm_of_a_cte(db):
    """A count / a join chain taken over INSIDE the recursive arm (a scalar subquery, an IN subquery): the pipelines
    of that arm are re-run per iteration by Executor::ReschedulePipelines, outside the main schedule, so the graph's
    sinks must run once before the pipeline that pulls from the CTE — not as dependencies of the inner reader
    (src/parallel/executor.cpp:140-170 dereferences such a dependency unchecked)."""
    d, vid = db
    s = int(vid[3])
    scalar = ("WITH RECURSIVE up(n) AS (SELECT 1 UNION SELECT n + 1 FROM up WHERE n < 6 AND n < "
              "(SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id)) "
              "SELECT count(*), max(n) FROM up")
    semi = ("WITH RECURSIVE reach(x, hop) AS ("
            f"SELECT {s}::BIGINT, 0 UNION SELECT k.k_person2id, r.hop + 1 FROM reach r, knows k "
            "WHERE r.x = k.k_person1id AND r.hop < 3 AND k.k_person2id IN "
            f"(SELECT k2.k_person2id FROM knows k1, knows k2 WHERE k1.k_person1id = {s} AND k1.k_person2id = k2.k_person1id)) "
            "SELECT count(*), sum(x % 1000), max(hop) FROM reach")
    for sql in (scalar, semi):
        d.execute("PRAGMA disable_gpu_graph")
        cpu = d.execute(sql)
        d.execute("PRAGMA enable_gpu_graph")
        try:
            plan = d.explain(sql)
            gpu = d.execute(sql)
            again = d.execute(sql)
        finally:
            d.execute("PRAGMA disable_gpu_graph")
        assert "GG_PATH" in plan and "GG_EDGE_SINK" in plan and "REC_CTE" in plan, plan
        assert cpu.shape[0] > 0 and np.array_equal(cpu, gpu) and np.array_equal(cpu, again), sql
    # (no prepared form here: the reference itself crashes on the second EXECUTE of a prepared recursive CTE with a
    # sink in its recursive arm — PhysicalRecursiveCTE::pipelines keeps the pipelines of the first execution)


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_join_chains_with_payload_columns_of_their_edges(db):
    """Columns of the edge instances other than the two keys (the reference gathers them from the hash join's build
    side per match, join_hashtable.cpp:466-473): the walks come back from the device with the rowid of every edge,
    the columns are fetched by rowid in the statement's transaction (GG_PATH_EDGES).  INTEGER and VARCHAR columns
    with NULLs, parallel edges (every combination of them is a row of its own), one and all sources, three hops,
    under an aggregate; a transaction with changes of its own keeps the reference's joins."""
    d, vid = db
    d.execute("CREATE TABLE kw (a BIGINT NOT NULL, b BIGINT NOT NULL, w INTEGER, tag VARCHAR)")
    d.execute("INSERT INTO kw SELECT k_person1id, k_person2id, CAST(k_person1id % 97 AS INTEGER), "
              "CASE WHEN k_person2id % 5 = 0 THEN NULL ELSE 't' || CAST(k_person2id % 11 AS VARCHAR) END "
              "FROM knows WHERE k_person1id >= 0 AND k_person2id >= 0 AND k_person1id % 4 = 0")
    d.execute("INSERT INTO kw SELECT a, b, w + 1000, 'dup' FROM kw WHERE a % 8 = 0")  # parallel edges, other payload
    s = int(d.execute("SELECT min(a) FROM kw")[0, 0])
    cases = [
        "SELECT k1.a, k1.w, k2.b, k2.w FROM kw k1, kw k2 WHERE k1.b = k2.a",
        f"SELECT k1.w, k2.tag, k2.b FROM kw k1, kw k2 WHERE k1.b = k2.a AND k1.a = {s}",
        "SELECT count(*), sum(k1.w), sum(k3.w), count(k2.tag), min(k3.tag) FROM kw k1, kw k2, kw k3 "
        "WHERE k1.b = k2.a AND k2.b = k3.a",
        # PREDICATES on payload columns of the edge instances (round 4): evaluated on the fetched columns, per walk
        "SELECT k1.a, k2.b FROM kw k1, kw k2 WHERE k1.b = k2.a AND k2.w > 50",
        "SELECT k1.a, k1.w, k2.b FROM kw k1, kw k2 WHERE k1.b = k2.a AND k1.w < 20 AND k2.tag = 't3'",
        f"SELECT k2.b, k2.tag FROM kw k1, kw k2 WHERE k1.b = k2.a AND k1.a = {s} AND k2.tag >= 't5'",
        "SELECT count(*) FROM kw k1, kw k2, kw k3 WHERE k1.b = k2.a AND k2.b = k3.a AND k2.w >= 1000 AND k3.tag = 'dup'",
    ]
    for sql in cases:
        d.execute("PRAGMA disable_gpu_graph")
        cpu = d.execute_text(sql)
        d.execute("PRAGMA enable_gpu_graph")
        try:
            plan = d.explain(sql)
            gpu = d.execute_text(sql)
        finally:
            d.execute("PRAGMA disable_gpu_graph")
        assert "GG_PATH_EDGES" in plan and "HASH_JOIN" not in plan, plan
        key = lambda row: tuple("\0NULL" if c is None else c for c in row)  # noqa: E731
        assert len(cpu) > 0 and sorted(cpu, key=key) == sorted(gpu, key=key), sql
    # rows this transaction has appended have no rowid the base table could be asked for: the joins stay
    d.execute("PRAGMA enable_gpu_graph")
    try:
        d.execute("BEGIN TRANSACTION")
        d.execute(f"INSERT INTO kw VALUES ({s}, {s}, 5, 'mine')")
        assert "GG_PATH_EDGES" not in d.explain(cases[0])
        inside = d.execute_text(cases[1])
        d.execute("ROLLBACK")
        # PREPARED in a clean transaction, EXECUTED inside one with rows of its own on the edge table: the rowid of such
        # a row cannot be fetched — the statement fails loudly, it never returns a wrong row
        d.execute("PREPARE payload_walk AS " + cases[1])
        clean = d.execute_text("EXECUTE payload_walk")
        d.execute("BEGIN TRANSACTION")
        d.execute(f"INSERT INTO kw VALUES ({s}, {s}, 5, 'mine')")
        with pytest.raises(RuntimeError, match="not committed yet"):
            d.execute_text("EXECUTE payload_walk")
        d.execute("ROLLBACK")
        assert sorted(map(str, d.execute_text("EXECUTE payload_walk"))) == sorted(map(str, clean))
        d.execute("DEALLOCATE payload_walk")
    finally:
        d.execute("PRAGMA disable_gpu_graph")
    assert any("mine" in " ".join(str(c) for c in row) for row in inside)


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_several_substituted_scans_in_one_plan(db):
    """Two (three) GPU scans in one statement, each with its own sink pipelines: under UNION ALL (the second scan
    is the source of a union pipeline), on both sides of a join, and under a UNION that is deduped."""
    d, vid = db
    a, b = int(vid[5]), int(vid[9])
    two = "SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id"
    three = ("SELECT count(*) FROM knows k1, knows k2, knows k3 WHERE k1.k_person2id = k2.k_person1id "
             "AND k2.k_person2id = k3.k_person1id")
    ends = "SELECT k2.k_person2id AS v FROM knows k1, knows k2 WHERE k1.k_person1id = {} AND k1.k_person2id = k2.k_person1id"
    cases = [
        f"{two} UNION ALL {three} UNION ALL {two}",
        f"SELECT count(*) FROM ({ends.format(a)}) x, ({ends.format(b)}) y WHERE x.v = y.v",
        f"SELECT count(*) FROM ({ends.format(a)} UNION {ends.format(b)}) u",
    ]
    for sql in cases:
        d.execute("PRAGMA disable_gpu_graph")
        cpu = d.execute(sql)
        d.execute("PRAGMA enable_gpu_graph")
        try:
            plan = d.explain(sql)
            gpu = d.execute(sql)
        finally:
            d.execute("PRAGMA disable_gpu_graph")
        assert plan.count("GG_EDGE_SINK") >= 2, plan
        assert np.array_equal(sort_rows(cpu), sort_rows(gpu)), sql


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_distinct_end_vertices_of_a_pinned_walk_are_a_set_image(db):
    """`SELECT DISTINCT <end vertex>` of a walk of h >= 2 edges from one source: the reference dedupes the projection of
    its joins with a hash aggregate; the rule plans GG_WALK_ENDPOINTS (set images of 1..h hops with a flag per length,
    on the device) and keeps the vertices whose flag h is set.  2, 3 and 4 hops, predicates on the end vertex, several
    sources (high and low degree, one that is nobody's friend)."""
    d, vid = db
    chain = lambda h, extra="": ("SELECT DISTINCT k{h}.k_person2id FROM " + ", ".join(f"knows k{i}" for i in range(1, h + 1)) +  # noqa: E731
                                 " WHERE k1.k_person1id = {s}" + "".join(f" AND k{i}.k_person2id = k{i + 1}.k_person1id"
                                                                          for i in range(1, h)) + extra).replace("{h}", str(h))
    stmts = []
    for s in (int(vid[7]), int(vid[500]), -12345):
        stmts += [chain(2).replace("{s}", str(s)), chain(3).replace("{s}", str(s)),
                  chain(2, f" AND k2.k_person2id <> {s}").replace("{s}", str(s)),
                  chain(3, " AND k3.k_person2id > 5000000000000").replace("{s}", str(s))]
    stmts.append(chain(4).replace("{s}", str(int(vid[7]))))
    d.execute("PRAGMA disable_gpu_graph")
    want = [sort_rows(d.execute(q)) for q in stmts]
    assert sum(w.shape[0] for w in want) > 1000
    d.execute("PRAGMA enable_gpu_graph")
    try:
        for q, w in zip(stmts, want):
            if "-12345" not in q:  # (a constant outside the column's range: the optimiser plans that one its own way)
                assert "GG_WALK_ENDPOINTS" in d.explain(q), d.explain(q)
            assert np.array_equal(sort_rows(d.execute(q)), w), q
    finally:
        d.execute("PRAGMA disable_gpu_graph")


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_generic_key_joins_stream_through_the_device_index(db):
    """PRAGMA enable_gpu_joins: ANY inner join on one integer key whose build side is a table scan runs as GG_KEY_JOIN —
    the build table sunk into a device index (Sink / Finalize), the probe side (any plan) streamed through Execute
    (gg_join_probe per chunk, build columns fetched by rowid).  Duplicates on both sides, NULL keys, more matches per
    probe chunk than a DataChunk holds (HAVE_MORE_OUTPUT), INTEGER and BIGINT keys, a chain of such joins, a join under
    an aggregate, and a join with a base table INSIDE the arm of a recursive CTE (rebuilt per iteration, like the
    reference's own hash join there).  Oracle: the same statement on the reference's hash joins."""
    d, vid = db
    d.execute("CREATE TABLE IF NOT EXISTS kl (a INTEGER, b BIGINT, s VARCHAR)")
    d.execute("CREATE TABLE IF NOT EXISTS kr (b BIGINT, c INTEGER, t VARCHAR)")
    d.execute("CREATE TABLE IF NOT EXISTS ra (i BIGINT, nxt BIGINT)")
    if int(d.execute("SELECT count(*) FROM kl")[0, 0]) == 0:
        d.execute("INSERT INTO kl SELECT i::INTEGER, (i % 7)::BIGINT, 'l' || i FROM range(3000) t(i)")
        d.execute("INSERT INTO kl VALUES (NULL, NULL, 'nulls'), (-1, 99, 'no match')")
        d.execute("INSERT INTO kr SELECT (i % 5)::BIGINT, i::INTEGER, 'r' || i FROM range(4000) t(i)")
        d.execute("INSERT INTO kr VALUES (NULL, 1, 'null key'), (6, NULL, 'null payload')")
        d.execute("INSERT INTO ra SELECT i, (i * 7 + 3) % 50 FROM range(50) t(i)")
    stmts = ["SELECT kl.a, kl.s, kr.c, kr.t FROM kl, kr WHERE kl.b = kr.b",
             "SELECT count(*), sum(kl.a), sum(kr.c) FROM kl JOIN kr ON kl.b = kr.b",
             "SELECT kl.a, kr.c FROM kl, kr WHERE kl.a = kr.c",
             "SELECT x.a, y.t, z.i FROM kl x JOIN kr y ON x.b = y.b JOIN ra z ON x.b = z.i WHERE x.a < 40",
             "SELECT k.k_person1id, p.p_personid FROM knows k JOIN person p ON k.k_person2id = p.p_personid WHERE k.k_person1id < 0",
             "WITH RECURSIVE t(x) AS (SELECT 1::BIGINT UNION SELECT ra.nxt FROM t, ra WHERE t.x = ra.i) SELECT x FROM t"]
    d.execute("PRAGMA disable_gpu_graph")
    want = [sort_rows(d.execute(q)) for q in stmts]
    text_want = sorted(d.query_text(stmts[0]))
    d.execute("PRAGMA enable_gpu_graph")
    assert "GG_" not in d.explain(stmts[0])  # (not a walk: the join rule alone leaves it)
    d.execute("PRAGMA enable_gpu_joins")
    try:
        plans = [d.explain(q) for q in stmts]
        assert all("GG_KEY_JOIN" in p for p in plans[:5]), plans
        assert "HASH_JOIN" not in plans[0]  # (in a chain, the joins whose build side is a scan are taken: plans[3])
        for q, w in zip(stmts, want):
            assert np.array_equal(sort_rows(d.execute(q)), w), q
        assert sorted(d.query_text(stmts[0])) == text_want  # (the VARCHAR payload columns of both sides)
        if "GG_KEY_JOIN" in plans[5]:  # (which side the reference's optimiser builds on is its choice)
            assert "REC_CTE" in plans[5]
        # repeated execution and a prepared statement: a fresh index per execution
        d.execute("PREPARE kj AS " + stmts[1])
        for _ in range(2):
            assert np.array_equal(sort_rows(d.execute("EXECUTE kj")), want[1])
        d.execute("DEALLOCATE kj")
        # a transaction with changes of its own declines (its rows have no fetchable rowid yet)
        d.execute("BEGIN TRANSACTION")
        d.execute("INSERT INTO kr VALUES (3, 123456, 'local')")
        assert "GG_KEY_JOIN" not in d.explain(stmts[0])
        d.execute("ROLLBACK")
        d.execute("PRAGMA disable_gpu_joins")
        assert "GG_KEY_JOIN" not in d.explain(stmts[0])
    finally:
        d.execute("PRAGMA disable_gpu_joins")
        d.execute("PRAGMA disable_gpu_graph")


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_count_of_a_single_key_join_is_a_sum_of_degrees(db):
    """count(*) over ONE inner equi-join on an integer key — duplicates on both sides, NULL keys, INTEGER and BIGINT
    columns, an empty side, the same table on both sides — planned as GG_JOIN_COUNT (build side = adjacency index keyed
    on its column, probe side = source list, rows = 1-hop walks counted from degrees); the reference's own vector for
    it is test/sql/join/inner/test_join_duplicates.test:14-24 (tests/test_reference_vectors.py replays it)."""
    d, vid = db
    d.execute("CREATE TABLE IF NOT EXISTS jl (a INTEGER, b BIGINT)")
    d.execute("CREATE TABLE IF NOT EXISTS jr (b BIGINT, c INTEGER)")
    d.execute("CREATE TABLE IF NOT EXISTS jempty (b BIGINT)")
    if int(d.execute("SELECT count(*) FROM jl")[0, 0]) == 0:
        d.execute("INSERT INTO jl VALUES (11, 1), (12, 2), (13, 3), (14, 1), (15, NULL), (16, 7), (NULL, 2)")
        d.execute("INSERT INTO jr SELECT (i % 5)::BIGINT, i::INTEGER FROM range(10240) t(i)")
        d.execute("INSERT INTO jr VALUES (NULL, 1), (NULL, 2), (2, NULL)")
    stmts = ["SELECT count(*) FROM jl INNER JOIN jr ON jl.b = jr.b",
             "SELECT count(*) FROM jr, jl WHERE jl.b = jr.b",
             "SELECT count(*) FROM jl, jr WHERE jl.a = jr.c",
             "SELECT count(*) FROM jl x, jl y WHERE x.b = y.b",
             "SELECT count(*) FROM jl JOIN jempty USING (b)",
             "SELECT count(*) FROM jempty JOIN jr USING (b)",
             "SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person2id",
             "SELECT count(*) FROM person p JOIN knows k ON p.p_personid = k.k_person1id"]
    d.execute("PRAGMA disable_gpu_graph")
    want = [d.execute(q) for q in stmts]
    d.execute("PRAGMA enable_gpu_graph")
    try:
        for q, w in zip(stmts, want):
            if "jempty" not in q:  # (the reference's optimiser folds a join with an empty table into EMPTY_RESULT)
                assert "GG_JOIN_COUNT" in d.explain(q), d.explain(q)
            assert np.array_equal(d.execute(q), w), (q, d.execute(q), w)
        # rows of such a join, other aggregates and joins with a further predicate stay with the reference
        assert "GG_" not in d.explain("SELECT jl.a, jr.c FROM jl, jr WHERE jl.b = jr.b")
        assert "GG_" not in d.explain("SELECT count(*) FROM jl, jr WHERE jl.b = jr.b AND jr.c > 5")
        assert "GG_" not in d.explain("SELECT sum(jr.c) FROM jl, jr WHERE jl.b = jr.b")
    finally:
        d.execute("PRAGMA disable_gpu_graph")


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_the_count_of_two_hop_walks_over_an_ownership_sharded_graph(db, monkeypatch):
    """GG_DEVICES=N: the plan shape whose result adds (count(*)) or concatenates (rows) over shards — all 2-hop walks —
    has its tables appended to N device contexts (device p mod the devices present: all on the one GPU here), N CSR shards
    built by gg_csr_build_shard and counted side by side; the counts add.  bench.py's N ranks inside one process,
    reached from the reference's executor.  Every other plan keeps its single graph."""
    d, vid = db
    d.execute("CREATE TABLE IF NOT EXISTS person_pk2 (p_personid BIGINT PRIMARY KEY)")
    if int(d.execute("SELECT count(*) FROM person_pk2")[0, 0]) == 0:
        d.execute("INSERT INTO person_pk2 SELECT p_personid FROM person")
    keyed = lambda sql: sql.replace("person ", "person_pk2 ")
    counts = [_chain(2, "count(*)"), "SELECT count(*) FROM knows", keyed(R.sql_khop(1)), keyed(R.sql_khop(2))]
    d.execute("PRAGMA disable_gpu_graph")
    want = [d.execute(q) for q in counts]
    rows_sql = _chain(2, "k1.k_person1id, k2.k_person2id")
    want_rows = d.execute(rows_sql)
    want3 = d.execute(_chain(3, "count(*)"))
    d.execute("PRAGMA enable_gpu_graph")
    try:
        for parts in ("3", "8"):
            monkeypatch.setenv("GG_DEVICES", parts)
            for q, w in zip(counts, want):
                if "GG_PATH_COUNT" not in d.explain(q):
                    continue  # (a bare count(*) of the edge table is not a walk pattern)
                if "2 hops" in d.explain(q):  # (a shard counts walks by their middle vertex: k_max = 2, gg.h)
                    assert f"shards: {parts}" in d.explain(q), d.explain(q)
                else:
                    assert "shards" not in d.explain(q)
                assert np.array_equal(d.execute(q), w), q
            # the ROWS of the 2-hop walks shard the same way (every part materialises the walks whose middle vertex it
            # owns, the pipeline's threads drain the parts side by side) — also part by part under a tiny budget
            assert f"shards: {parts}" in d.explain(rows_sql) and "shards" not in d.explain(_chain(3, "count(*)"))
            assert np.array_equal(sort_rows(d.execute(rows_sql)), sort_rows(want_rows))
            monkeypatch.setenv("GG_RESULT_BUDGET_MB", "1")
            assert np.array_equal(sort_rows(d.execute(rows_sql)), sort_rows(want_rows))
            monkeypatch.delenv("GG_RESULT_BUDGET_MB")
            assert np.array_equal(d.execute(_chain(3, "count(*)")), want3)
        # the scan-function route builds the same shards from its own sink pipelines
        monkeypatch.setenv("GG_NO_PIPELINE_SINKS", "1")
        assert "GG_EDGE_SINK" not in d.explain(counts[0]) and "shards: 8" in d.explain(counts[0])
        assert np.array_equal(d.execute(counts[0]), want[0])
        assert np.array_equal(d.execute(counts[3]), want[3])
        monkeypatch.delenv("GG_NO_PIPELINE_SINKS")
        # prepared: the graph is rebuilt, sharded, on every execution
        d.execute("PREPARE sharded AS " + counts[0])
        for _ in range(2):
            assert np.array_equal(d.execute("EXECUTE sharded"), want[0])
        d.execute("DEALLOCATE sharded")
    finally:
        d.execute("PRAGMA disable_gpu_graph")


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_repeated_statements_and_pin_cycles_on_both_ingest_routes(db, monkeypatch):
    """The same statements over and over: through pipeline sinks, through the scan-function route (ingest tasks on the
    reference's scheduler) and on a graph pinned and unpinned each round.  (An ingest task whose first morsel was the
    transaction-local one used to walk a row group at an unset pointer: one fault in ~40 pinned builds at SF100,
    scripts/stress_sql.py; this loop keeps the paths it took exercised.)"""
    d, vid = db
    s = int(vid[11])
    stmts = [_chain(2, "count(*)"),
             _chain(2, "k1.k_person1id, k2.k_person2id") + f" AND k1.k_person1id = {s}",
             R.sql_shortest([int(v) for v in vid[:7]], 3)]
    d.execute("PRAGMA disable_gpu_graph")
    want = [sort_rows(d.execute(q)) for q in stmts]
    d.execute("PRAGMA enable_gpu_graph")
    try:
        for _ in range(6):
            for route in ("sinks", "scan"):
                if route == "scan":
                    monkeypatch.setenv("GG_NO_PIPELINE_SINKS", "1")
                else:
                    monkeypatch.delenv("GG_NO_PIPELINE_SINKS", raising=False)
                for q, w in zip(stmts, want):
                    assert np.array_equal(sort_rows(d.execute(q)), w), (route, q)
            monkeypatch.delenv("GG_NO_PIPELINE_SINKS", raising=False)
            d.execute("PRAGMA gg_use_pinned_graphs")
            d.execute(f"SELECT * FROM gg_graph_pin({GRAPH})")
            d.execute("SELECT * FROM gg_graph_pin('', '', 'knows', 'k_person1id', 'k_person2id')")
            for q, w in zip(stmts, want):
                assert np.array_equal(sort_rows(d.execute(q)), w), ("pinned", q)
            d.execute("SELECT * FROM gg_graph_unpin()")
            d.execute("PRAGMA gg_ignore_pinned_graphs")
    finally:
        monkeypatch.delenv("GG_NO_PIPELINE_SINKS", raising=False)
        d.execute("PRAGMA gg_ignore_pinned_graphs")
        d.execute("PRAGMA disable_gpu_graph")


@pytest.mark.skipif(not bool(R.rules_route()), reason="plan hook shim not built")
def test_randomised_statements_with_the_rules_off_and_on():
    """scripts/fuzz_sql.py for a fixed number of statements (its own database): random join chains, friends unions,
    shortest-path CTEs and key joins, each with the planner rules off and on — the same rows.  Longer runs:
    profiles/r04_fuzz.txt."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_sql.py"), "--seed", "11", "--iterations", "300"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fuzz_sql ok: 300 statements" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
