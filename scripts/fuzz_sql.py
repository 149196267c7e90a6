#!/usr/bin/env python3
"""Randomised differential test of the planner rules: random statements of the shapes the rules know (join chains over
an edge table with and without vertex tables, pinned sources, DISTINCT, aggregates, payload columns and predicates, the
friends UNION, the recursive shortest-path CTE, single-key joins) and of shapes near them, each executed inside the
compiled reference with the rules off (its own hash joins) and on; the two results must be the same multiset of rows,
whatever the rules decided.  Reports how many statements ended up on GG operators.

    python scripts/fuzz_sql.py [--seconds 300] [--seed 1] [--iterations 0]

Test infrastructure (the compiled reference is the checker)."""
import argparse
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GG_CRASH_TRACE", "1")
import numpy as np  # noqa: E402

from duckdb_pgq_amd import datagen  # noqa: E402
from oracle import ref_duckdb as R  # noqa: E402


def make_db(seed, threads):
    rng = np.random.default_rng(seed)
    V = int(rng.integers(20, 400))
    E = int(rng.integers(V, 12 * V))
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=int(rng.integers(0, 6)), dup_edges=int(rng.integers(0, 30)))
    d = R.RefDuckDB(threads=threads)
    d.load_table("person", {"p_personid": vid})
    d.execute("CREATE TABLE person_pk (p_personid BIGINT PRIMARY KEY)")
    d.execute("INSERT INTO person_pk SELECT p_personid FROM person")
    w = (np.arange(src.size) * 7919 + seed) % 23
    d.load_table("knows", {"k_person1id": src, "k_person2id": dst, "w": w})
    # the same edges with NULLs in the keys and a text payload
    d.execute("CREATE TABLE knows_n AS SELECT CASE WHEN w = 3 THEN NULL ELSE k_person1id END AS k_person1id, "
              "CASE WHEN w = 5 THEN NULL ELSE k_person2id END AS k_person2id, w, 't' || CAST(w AS VARCHAR) AS tag FROM knows")
    n1, n2 = int(rng.integers(5, 300)), int(rng.integers(5, 300))
    d.load_table("t1", {"k": rng.integers(0, 40, n1), "x": rng.integers(0, 1000, n1)})
    d.load_table("t2", {"k": rng.integers(0, 40, n2), "y": rng.integers(0, 1000, n2)})
    d.execute(f"LOAD '{R.EXTENSION}'")
    return d, vid, src, dst


def pick_ids(rnd, vid, n):
    ids = [int(vid[rnd.randrange(len(vid))]) for _ in range(n)]
    if rnd.random() < 0.3:
        ids.append(rnd.choice([-12345, 7, 1 << 40]))
    return ids


def chain_statement(rnd, vid):
    h = rnd.choice([1, 2, 2, 2, 3, 3, 4])
    edge = rnd.choice(["knows", "knows", "knows", "knows_n"])
    vertex = rnd.choice([None, None, "person", "person_pk"])
    ks = [f"k{i}" for i in range(1, h + 1)]
    frm = [f"{edge} {k}" for k in ks]
    cond = []
    for i in range(1, h):
        a, b = f"k{i}.k_person2id", f"k{i + 1}.k_person1id"
        cond.append(f"{a} = {b}" if rnd.random() < 0.7 else f"{b} = {a}")
    ps = []
    if vertex:
        positions = list(range(h + 1)) if rnd.random() < 0.7 else sorted(rnd.sample(range(h + 1), rnd.randint(1, h + 1)))
        for p in positions:
            ps.append(p)
            frm.append(f"{vertex} p{p}")
            col = f"k{p}.k_person2id" if p > 0 else "k1.k_person1id"
            if 0 < p < h and rnd.random() < 0.5:
                col = f"k{p + 1}.k_person1id"
            cond.append(f"p{p}.p_personid = {col}" if rnd.random() < 0.5 else f"{col} = p{p}.p_personid")
    rnd.shuffle(frm)
    rnd.shuffle(cond)
    pos_col = lambda p: ("k1.k_person1id" if p == 0 else f"k{p}.k_person2id")  # noqa: E731
    filters = []
    r = rnd.random()
    if r < 0.35:
        filters.append(f"k1.k_person1id = {pick_ids(rnd, vid, 1)[0]}")
    elif r < 0.5:
        filters.append(f"k1.k_person1id IN ({', '.join(map(str, pick_ids(rnd, vid, rnd.randint(1, 5))))})")
    elif r < 0.55:
        lo = int(vid[rnd.randrange(len(vid))])
        filters.append(f"k1.k_person1id BETWEEN {lo} AND {lo + rnd.choice([0, 1000, 10 ** 12])}")
    if rnd.random() < 0.2:
        a, b = rnd.sample(range(h + 1), 2) if h >= 1 else (0, 1)
        filters.append(f"{pos_col(a)} {rnd.choice(['<>', '<', '>=', '='])} {pos_col(b)}")
    if rnd.random() < 0.15:
        filters.append(f"{pos_col(h)} <> {pick_ids(rnd, vid, 1)[0]}")
    if rnd.random() < 0.2:
        k = rnd.choice(ks)
        filters.append(rnd.choice([f"{k}.w > {rnd.randint(0, 22)}", f"{k}.w = {rnd.randint(0, 22)}", f"{k}.w % 2 = 0",
                                   f"{k}.w BETWEEN 4 AND 15"] + ([f"{k}.tag >= 't5'", f"{k}.tag IS NOT NULL"] if edge == "knows_n" else [])))
    heavy = h >= 3 and not filters
    sel_kind = rnd.random()
    cols = [pos_col(p) for p in range(h + 1)]
    if heavy or sel_kind < 0.3:
        select = "count(*)"
    elif sel_kind < 0.45:
        select = "DISTINCT " + pos_col(h)
    elif sel_kind < 0.55:
        select = f"DISTINCT {pos_col(0)}, {pos_col(h)}"
    elif sel_kind < 0.65:
        select = f"{pos_col(0)}, count(*)"
    elif sel_kind < 0.75:
        k = rnd.choice(ks)
        select = ", ".join(rnd.sample(cols, rnd.randint(1, len(cols))) + [f"{k}.w"] + ([f"{k}.tag"] if edge == "knows_n" and rnd.random() < 0.5 else []))
    elif sel_kind < 0.8:
        select = f"count(*), sum({rnd.choice(ks)}.w), min({pos_col(h)})"
    else:
        select = ", ".join(rnd.sample(cols, rnd.randint(1, len(cols))))
    where = " AND ".join(cond + filters)
    sql = f"SELECT {select} FROM {', '.join(frm)}" + (f" WHERE {where}" if where else "")
    if select.startswith(pos_col(0) + ", count"):
        sql += f" GROUP BY {pos_col(0)}"
    return sql


def friends_statement(rnd, vid):
    s = pick_ids(rnd, vid, 1)[0]
    edge = rnd.choice(["knows", "knows_n"])
    outer = rnd.choice(["select count(*) from ({}) f", "select * from ({}) f", "select f.k_person2id, count(*) from ({}) f group by 1"])
    excl = rnd.choice([f" and k2.k_person2id <> {s}", ""])
    inner = (f"select k_person2id from {edge} where k_person1id = {s} union select k2.k_person2id from {edge} k1, {edge} k2 "
             f"where k1.k_person1id = {s} and k1.k_person2id = k2.k_person1id{excl}")
    return outer.format(inner)


def shortest_statement(rnd, vid):
    seeds = pick_ids(rnd, vid, rnd.choice([1, 2, 5, 64, 70]))
    sql = R.sql_shortest(seeds, rnd.choice([0, 1, 2, 3, 5, 9]))
    if rnd.random() < 0.5:
        sql = sql.replace(", person p", "").replace("AND k.k_person2id = p.p_personid ", "")
    if rnd.random() < 0.4:
        sql = sql.replace("SELECT startPerson, friend, min(hopCount) AS hopCount FROM friends GROUP BY startPerson, friend",
                          "SELECT count(*), sum(hopCount) FROM (SELECT startPerson, friend, min(hopCount) AS hopCount FROM friends "
                          "GROUP BY startPerson, friend) t")
    return sql


def key_join_statement(rnd, vid):
    sel = rnd.choice(["count(*)", "a.x, b.y", "a.k, a.x, b.y", "b.y", "sum(a.x + b.y)", "a.k, count(*)"])
    form = rnd.choice(["FROM t1 a, t2 b WHERE a.k = b.k", "FROM t1 a JOIN t2 b ON a.k = b.k", "FROM t2 b JOIN t1 a ON b.k = a.k"])
    extra = rnd.choice(["", " AND a.x > 500" if "WHERE" in form else " WHERE a.x > 500", ""])
    sql = f"SELECT {sel} {form}{extra}"
    if sel == "a.k, count(*)":
        sql += " GROUP BY a.k"
    return sql


def rows_of(d, sql):
    return sorted(d.execute_text(sql), key=lambda r: tuple("" if x is None else x for x in r))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--iterations", type=int, default=0)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--pinned", action="store_true", help="the graphs of knows / knows_n pinned on the device, the connection opted in")
    ap.add_argument("--writes", action="store_true", help="INSERT / DELETE / UPDATE on the edge tables every 40 statements")
    ap.add_argument("--cpu-only", action="store_true", help="the reference's plan only (checks the generator's SQL where there is no GPU)")
    a = ap.parse_args()
    if not R.rules_route():
        print("no rule route (plan hook shim / call-outs) in this build: nothing to test")
        return
    rnd = random.Random(a.seed)
    t0 = last = time.time()
    i = taken = writes = 0
    d = None
    kinds = {}
    while (a.iterations and i < a.iterations) or (not a.iterations and time.time() - t0 < a.seconds):
        if i % 150 == 0:  # a fresh database (new graph) every so often
            if d:
                d.close()
            d, vid, src, dst = make_db(a.seed * 1000 + i, a.threads)
            if a.pinned and not a.cpu_only:
                d.execute("PRAGMA gg_use_pinned_graphs")
                for edge in ("knows", "knows_n"):
                    d.execute(f"SELECT * FROM gg_graph_pin('', '', '{edge}', 'k_person1id', 'k_person2id')")
                    for vertex in ("person", "person_pk"):
                        d.execute(f"SELECT * FROM gg_graph_pin('{vertex}', 'p_personid', '{edge}', 'k_person1id', 'k_person2id')")
        if a.writes and i % 40 == 39:  # the tables change under the statements (and under the pins, which must go)
            w = rnd.choice(["insert", "insert", "delete", "update", "repin"])
            table = rnd.choice(["knows", "knows_n"])
            if w == "insert":
                vals = ", ".join(f"({pick_ids(rnd, vid, 1)[0]}, {pick_ids(rnd, vid, 1)[0]}, {rnd.randint(0, 22)}"
                                 + (", 'tx')" if table == "knows_n" else ")") for _ in range(rnd.randint(1, 6)))
                d.execute(f"INSERT INTO {table} VALUES {vals}")
            elif w == "delete":
                d.execute(f"DELETE FROM {table} WHERE w = {rnd.randint(0, 22)} AND k_person1id % 3 = {rnd.randint(0, 2)}")
            elif w == "update":
                d.execute(f"UPDATE {table} SET w = (w + 1) % 23 WHERE w = {rnd.randint(0, 22)}")
            elif a.pinned:
                d.execute(f"SELECT * FROM gg_graph_pin('', '', '{table}', 'k_person1id', 'k_person2id')")
            writes += 1
        kind = rnd.choice(["chain"] * 6 + ["friends", "shortest", "keyjoin"])
        sql = {"chain": chain_statement, "friends": friends_statement, "shortest": shortest_statement,
               "keyjoin": key_join_statement}[kind](rnd, vid)
        joins = rnd.random() < 0.5
        try:
            d.execute("PRAGMA disable_gpu_graph")
            d.execute("PRAGMA disable_gpu_joins")
            cpu = rows_of(d, sql)
            if a.cpu_only:
                i += 1
                continue
            d.execute("PRAGMA enable_gpu_graph")
            if joins:
                d.execute("PRAGMA enable_gpu_joins")
            on_gg = "GG_" in d.explain(sql)
            gpu = rows_of(d, sql)
            if cpu != gpu:
                raise AssertionError(f"{len(cpu)} rows from the reference's plan, {len(gpu)} with the rules on; first difference: "
                                     f"{next(((x, y) for x, y in zip(cpu, gpu) if x != y), (cpu[-1:] , gpu[-1:]))}")
        except Exception:
            print(f"FAILED at iteration {i} (rerun: --seed {a.seed} --iterations {i + 1}), gpu joins {joins}:\n{sql}", flush=True)
            traceback.print_exc()
            sys.exit(1)
        taken += on_gg
        k = kinds.setdefault(kind, [0, 0])
        k[0] += 1
        k[1] += on_gg
        i += 1
        if time.time() - last > 20:
            last = time.time()
            print(f"[{last - t0:5.0f} s] {i} statements, {taken} on GG operators; by kind (all, taken): {kinds}", flush=True)
    print(f"fuzz_sql ok: {i} statements in {time.time() - t0:.0f} s, {taken} on GG operators (seed {a.seed}), {writes} writes in between; by kind: {kinds}", flush=True)
    if d:
        d.close()


if __name__ == "__main__":
    main()
