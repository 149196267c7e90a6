"""Planner rules (duckdb_pgq_amd/host/gg_plan_rule.cpp), CPU part: which logical plans of the compiled
reference are taken over and which are left alone.  Only EXPLAIN is used here — planning touches no GPU
(the scans open at execution time); result parity of the substituted plans is in
tests/test_duckdb_extension.py (-m gpu).  Needs oracle/_ref (the compiled reference), the extension and
the interposition shim, all built by __graft_entry__.build() where /root/reference exists."""
import os

import pytest

from oracle import ref_duckdb as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = os.path.join(ROOT, "duckdb_pgq_amd", "gg_duckdb.duckdb_extension")

pytestmark = pytest.mark.skipif(
    not (R.available() and os.path.exists(EXT) and os.path.exists(R.PLAN_HOOK)),
    reason="reference build / extension / plan hook not present")


@pytest.fixture(scope="module")
def db():
    d = R.RefDuckDB(threads=2)
    d.execute("CREATE TABLE person (p_personid BIGINT PRIMARY KEY)")
    d.execute("CREATE TABLE person_nokey (p_personid BIGINT NOT NULL)")
    d.execute("CREATE TABLE knows (k_person1id BIGINT NOT NULL, k_person2id BIGINT NOT NULL, k_weight INTEGER)")
    d.execute("CREATE TABLE knows_nullable (a BIGINT, b BIGINT)")
    d.execute("CREATE TABLE e32 (a INTEGER NOT NULL, b INTEGER NOT NULL)")
    d.execute("INSERT INTO person VALUES (1), (2), (3)")
    d.execute("INSERT INTO knows VALUES (1, 2, 0), (2, 3, 0), (3, 1, 0)")
    # (statistics propagation turns joins over empty tables into EMPTY_RESULT before any rule sees them)
    d.execute("INSERT INTO person_nokey VALUES (1), (2), (3)")
    d.execute("INSERT INTO knows_nullable VALUES (1, 2), (2, 3), (3, NULL)")
    d.execute("INSERT INTO e32 VALUES (1, 2), (2, 3), (3, 1)")
    d.execute(f"LOAD '{EXT}'")
    yield d
    d.execute("PRAGMA disable_gpu_graph")
    d.close()


def chain(h, table="knows", a="k_person1id", b="k_person2id", select="count(*)"):
    frm = ", ".join(f"{table} k{i}" for i in range(1, h + 1))
    cond = " AND ".join(f"k{i}.{b} = k{i+1}.{a}" for i in range(1, h))
    return f"SELECT {select} FROM {frm} WHERE {cond}"


TAKEN = [
    # edge-only chains (interactive-complex-3.sql:9-11 idiom)
    (chain(2), "GG_PATH_COUNT", "2 hops"),
    (chain(4), "GG_PATH_COUNT", "4 hops"),
    (chain(3, select="k1.k_person1id, k3.k_person2id"), "GG_PATH_EXPAND", "3 hops"),
    (chain(2, table="e32", a="a", b="b", select="k1.a, k2.b"), "GG_PATH_EXPAND", "2 hops"),
    # the far end written first: the walk is read in the other direction
    ("SELECT count(*) FROM knows k1, knows k2 WHERE k2.k_person1id = k1.k_person2id", "GG_PATH_COUNT", "2 hops"),
    # vertex-validated chains (the oracle's SQL formulation, oracle/ref_duckdb.py sql_khop)
    (R.sql_khop(1), "GG_PATH_COUNT", "vertices: person.p_personid"),
    (R.sql_khop(2), "GG_PATH_COUNT", "vertices: person.p_personid"),
    (R.sql_khop_rows(2), "GG_PATH_EXPAND", "2 hops"),
    # single source pinned by a constant
    (chain(2, select="k2.k_person2id") + " AND k1.k_person1id = 2", "GG_PATH_EXPAND", "from 2"),
    (chain(2) + " AND k2.k_person2id = 3", "GG_PATH_COUNT", "from 3"),
]

LEFT_ALONE = [
    # not a walk: common neighbour, same-source fan, self-loop filter, cycle
    "SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person2id",
    "SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person1id = k2.k_person1id",
    "SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person2id = k2.k_person1id AND k2.k_person2id = k1.k_person1id",
    # only part of the walk is validated against the vertex table
    "SELECT count(*) FROM knows k1, knows k2, person p WHERE k1.k_person2id = k2.k_person1id AND p.p_personid = k2.k_person2id",
    # vertex key without a uniqueness constraint: a duplicate id would multiply rows on the CPU side
    "SELECT count(*) FROM person_nokey p0, knows k1, person_nokey p1 "
    "WHERE p0.p_personid = k1.k_person1id AND k1.k_person2id = p1.p_personid",
    # nullable edge columns in an edge-only chain: the outer ends are not join keys
    chain(2, table="knows_nullable", a="a", b="b"),
    # payload columns, non-equality predicates, outer joins, other filters
    chain(2, select="k1.k_weight"),
    "SELECT count(*) FROM knows k1, knows k2 WHERE k1.k_person2id < k2.k_person1id",
    "SELECT count(*) FROM knows k1 LEFT JOIN knows k2 ON k1.k_person2id = k2.k_person1id",
    chain(2) + " AND k1.k_person1id > 2",
    chain(2) + " AND k2.k_person1id = 2",
    # aggregates other than an ungrouped count(*) keep their aggregate; the join under them is still a walk
]


def test_rules_are_inert_until_enabled(db):
    db.execute("PRAGMA disable_gpu_graph")
    plan = db.explain(chain(2))
    assert "HASH_JOIN" in plan and "GG_" not in plan


@pytest.mark.parametrize("sql,operator,detail", TAKEN)
def test_walk_patterns_are_taken_over(db, sql, operator, detail):
    db.execute("PRAGMA enable_gpu_graph")
    plan = db.explain(sql)
    assert operator in plan and "HASH_JOIN" not in plan, plan
    assert detail in " ".join(plan.replace("│", " ").split()), plan


@pytest.mark.parametrize("sql", LEFT_ALONE)
def test_everything_else_is_left_to_the_reference(db, sql):
    db.execute("PRAGMA enable_gpu_graph")
    plan = db.explain(sql)
    assert "GG_" not in plan, plan


def test_sub_chain_of_a_larger_join_and_grouped_aggregate(db):
    db.execute("PRAGMA enable_gpu_graph")
    # grouped aggregate: the aggregate stays, the join below it is substituted
    plan = db.explain(chain(2, select="k1.k_person1id, count(*)") + " GROUP BY k1.k_person1id")
    assert "GG_PATH_EXPAND" in plan and "HASH_GROUP_BY" in plan and "HASH_JOIN" not in plan
    # a walk joined with something else: the walk part is substituted where the reference's join-order
    # optimiser keeps it together as a subtree (a derived table does)
    plan = db.explain("SELECT count(*) FROM (SELECT k2.k_person2id AS f FROM knows k1, knows k2 "
                      "WHERE k1.k_person2id = k2.k_person1id) w, knows_nullable x WHERE x.a = w.f")
    assert "GG_PATH_EXPAND" in plan and plan.count("HASH_JOIN") == 1


# ---- recursive CTE + min(hop)  ->  64-lane BFS --------------------------------------------------------
def friends(seed_where="WHERE p_personid = 2", union="UNION", step_from="friends f, knows k",
            step_where="f.friend = k.k_person1id AND f.hopCount < 5", nxt="k.k_person2id", inc="f.hopCount+1",
            agg="min(hopCount)", vertex="person"):
    return f"""WITH RECURSIVE friends(startPerson, hopCount, friend) AS (
        SELECT p_personid, 0, p_personid FROM {vertex} {seed_where}
      {union}
        SELECT f.startPerson, {inc}, {nxt} FROM {step_from} WHERE {step_where})
    SELECT startPerson, {agg} AS hopCount, friend FROM friends GROUP BY startPerson, friend"""


BI10_CASE = "CASE WHEN f.friend = k.k_person1id then k.k_person2id ELSE k.k_person1id END"

BFS_TAKEN = [
    (friends(), "from 1 id"),
    # the literal friends / friends_shortest text of benchmark/ldbc/queries/bi-10-shortestpath.sql:8-31
    (friends(seed_where="WHERE 1=1 AND p_personid = 2", nxt=BI10_CASE,
             step_where="1=1 AND f.friend = k.k_person1id AND f.hopCount < 5"), "vertices: endpoint ids"),
    (friends(seed_where="WHERE p_personid IN (1, 3)"), "from 2 ids"),
    (friends(seed_where="WHERE p_personid IN (" + ", ".join(str(i * 3 + 1) for i in range(40)) + ")"), "from 40 ids"),
    (friends(seed_where="WHERE p_personid IN (1, 2)"), "from person where"),   # consecutive values: becomes a pushed-down range
    (friends(seed_where=""), "from every person"),
    (friends(step_where="f.friend = k.k_person1id AND f.hopCount <= 2"), "min hops <= 3"),
    # the oracle's formulation: the new vertex validated against the vertex table
    (R.sql_shortest([1, 2], 4), "vertices: person.p_personid"),
]

BFS_LEFT_ALONE = [
    friends(union="UNION ALL"),                                   # bag semantics: not a fixpoint of sets
    friends(step_where="f.friend = k.k_person1id"),               # no hop bound
    friends(inc="f.hopCount+2"),
    friends(agg="max(hopCount)"),
    friends(nxt="k.k_person1id"),                                 # does not advance along the edge
    friends(step_where="f.friend = k.k_person1id AND f.hopCount < 5 AND k.k_weight > 0"),
    friends(step_from="friends f, knows_nullable k", step_where="f.friend = k.a AND f.hopCount < 5", nxt="k.b"),
    friends(step_from="friends f, knows k, person_nokey p",
            step_where="f.friend = k.k_person1id AND k.k_person2id = p.p_personid AND f.hopCount < 5"),
]


@pytest.mark.parametrize("sql,detail", BFS_TAKEN)
def test_friends_cte_with_min_hop_becomes_bfs(db, sql, detail):
    db.execute("PRAGMA enable_gpu_graph")
    plan = db.explain(sql)
    assert "GG_SHORTEST_PATH_BFS" in plan and "REC_CTE" not in plan and "HASH_GROUP_BY" not in plan, plan
    assert detail in " ".join(plan.replace("│", " ").split()), plan


@pytest.mark.parametrize("sql", BFS_LEFT_ALONE)
def test_other_recursive_ctes_are_left_alone(db, sql):
    db.execute("PRAGMA enable_gpu_graph")
    plan = db.explain(sql)
    assert "GG_" not in plan and "REC_CTE" in plan, plan
