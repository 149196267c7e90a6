#!/usr/bin/env python3
"""Randomised differential test of the C-ABI (libgg.so on cuda:0) against the CPU oracle — the same bar as
tests/test_gpu_parity.py (bit-exact), on shapes nobody wrote down: every iteration draws a graph (size, id
distribution, degree model, dangling and duplicate rows, vertex table or endpoints only), a build mode and a few
operations with random arguments and testing knobs, and compares everything the library returns with the oracle.
Stops at the first difference and prints the seed that reproduces it.

    python scripts/fuzz_gg.py [--seconds 420] [--seed 1] [--iterations 0] [--max-rows 3000000] [--max-vertices 300000]
                                [--min-frac 0]

Test infrastructure (it calls oracle/): never part of the product path."""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import duckdb_pgq_amd as pkg  # noqa: E402
from duckdb_pgq_amd import datagen  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.oracle_lib import sort_rows  # noqa: E402

I64 = np.iinfo(np.int64)


def log_uniform(rng, lo, hi):
    return int(round(np.exp(rng.uniform(np.log(max(lo, 1)), np.log(max(hi, 1))))))


def draw_ids(rng, V):
    """V distinct int64 ids in one of the shapes the dictionaries distinguish."""
    kind = rng.choice(["ldbc", "dense", "dense_offset", "clustered", "wide", "negative"])
    if kind == "ldbc":
        return kind, datagen.person_ids(V, int(rng.integers(1, 1 << 30)))
    if kind == "dense":
        return kind, rng.permutation(V).astype(np.int64)
    if kind == "dense_offset":
        return kind, rng.permutation(V).astype(np.int64) + int(rng.integers(-(1 << 40), 1 << 40))
    if kind == "clustered":  # runs of consecutive ids far apart
        runs = max(1, V // int(rng.integers(1, 500)))
        base = np.sort(rng.choice(1 << 44, size=runs, replace=False)).astype(np.int64) << 16
        ids = (np.repeat(base, -(-V // runs))[:V] + np.arange(V) % (-(-V // runs))).astype(np.int64)
        return kind, rng.permutation(np.unique(ids))
    if kind == "wide":  # anywhere in int64, the extremes included
        ids = np.unique(rng.integers(I64.min, I64.max, size=V + 8, dtype=np.int64, endpoint=True))
        ids = np.unique(np.concatenate([ids, np.array([I64.min, I64.max, 0, -1], np.int64)]))
        return kind, rng.permutation(ids)[:V] if ids.size >= V else rng.permutation(ids)
    ids = -np.unique(rng.integers(1, 1 << 50, size=V + 8, dtype=np.int64))
    return kind, rng.permutation(ids)[:V]


def draw_graph(rng, max_rows, max_vertices=300_000, min_frac=0.0):
    V = log_uniform(rng, max(1, max_vertices * min_frac), max_vertices)
    kind, vid = draw_ids(rng, V)
    V = vid.size
    E = 0 if rng.random() < 0.03 else min(log_uniform(rng, max(1, max_rows * min_frac), max_rows), max(1, V) * 4000)
    model = rng.choice(["uniform", "powerlaw", "hubs", "sorted_src", "chain"])
    if E == 0:
        s = d = np.zeros(0, np.int64)
    elif model == "uniform":
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    elif model == "powerlaw":
        w = rng.pareto(1.3, V) + 1e-3
        p = w / w.sum()
        s, d = rng.choice(V, E, p=p), rng.choice(V, E, p=p)
    elif model == "hubs":  # a few vertices with most of the rows, out-rows past every store form's boundary
        hubs = rng.integers(0, V, max(1, min(V, 4)))
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        m = rng.random(E) < 0.6
        s[m] = hubs[rng.integers(0, hubs.size, int(m.sum()))]
        m = rng.random(E) < 0.3
        d[m] = hubs[rng.integers(0, hubs.size, int(m.sum()))]
    elif model == "sorted_src":
        s, d = np.sort(rng.integers(0, V, E)), rng.integers(0, V, E)
    else:  # long paths: many BFS levels
        s = np.arange(E) % V
        d = (s + 1 + (rng.random(E) < 0.01) * rng.integers(0, V, E)) % V
    src, dst = vid[s], vid[d]
    if E and rng.random() < 0.3:  # duplicate rows
        k = int(rng.integers(1, max(2, E // 10)))
        src, dst = np.concatenate([src, src[:k]]), np.concatenate([dst, dst[:k]])
    return {"ids": kind, "model": str(model), "vid": vid, "src": src.astype(np.int64), "dst": dst.astype(np.int64)}


def dense_of(o_vid, ids):
    """Dense index of each id that is a vertex (order and repeats kept) — numpy only."""
    if o_vid.size == 0:
        return np.zeros(0, np.uint32)
    order = np.argsort(o_vid, kind="stable")
    sv = o_vid[order]
    pos = np.minimum(np.searchsorted(sv, ids), sv.size - 1)
    ok = sv[pos] == ids
    return order[pos[ok]].astype(np.uint32)


def check(cond, what):
    if not cond:
        raise AssertionError(what)


def one_iteration(gg, orc, rng, a, note):
    max_rows = a.max_rows
    g = draw_graph(rng, a.max_rows, a.max_vertices, a.min_frac)
    vid, src, dst = g["vid"], g["src"], g["dst"]
    edge_only = rng.random() < 0.4
    keep_rowid = rng.random() < 0.5
    legacy = rng.random() < 0.1
    chunk = int(rng.choice([0, 1024, 4096, 122_880]))
    rowid = None
    if keep_rowid and rng.random() < 0.5 and src.size:
        rowid = rng.permutation(src.size).astype(np.int64) * 3 + 1
    if not edge_only and src.size and rng.random() < 0.3:  # rows whose endpoint is not in the vertex table
        k = int(rng.integers(1, 20))
        stranger = np.setdiff1d(np.array([7, -9, 1 << 61, -(1 << 61)], np.int64), vid)
        if stranger.size:
            src = np.concatenate([src, np.full(k, stranger[0], np.int64)])
            dst = np.concatenate([dst, vid[rng.integers(0, vid.size, k)]])
            if rowid is not None:
                rowid = np.concatenate([rowid, np.arange(k, dtype=np.int64) + 10 * src.size])
    note.update(ids=g["ids"], model=g["model"], V=int(vid.size), E=int(src.size), edge_only=edge_only,
                keep_rowid=keep_rowid, legacy=legacy, chunk=chunk, explicit_rowid=rowid is not None)
    gg.debug_reset()
    gg.staging_clear()
    gg.set_edge_rowid(keep_rowid)
    gg.force_legacy_build(legacy)
    gg.chunk_rows = chunk
    if edge_only:
        gg.append_edges(src, dst, rowid)
        n = gg.vertices_from_edges()
        vtab = np.unique(np.concatenate([src, dst]))
        check(n == vtab.size, f"vertices_from_edges: {n} != {vtab.size}")
    else:
        vtab = vid
        gg.append_vertices(vid)
        gg.append_edges(src, dst, rowid)
    gg.chunk_rows = 0
    csr = gg.build_csr()
    rc, og = orc.csr_build(vtab, src, dst, rowid)
    check(rc == 0, f"oracle build rc {rc}")
    try:
        o_off, o_nbr, o_eid, o_vid = og.arrays()
        check((csr.V, csr.E, csr.dropped) == (og.V, og.E, og.dropped), "V/E/dropped")
        off, nbr, eid, v = csr.export()
        check(np.array_equal(v, o_vid), "vertex ids")
        check(np.array_equal(off, o_off), "row offsets")
        check(np.array_equal(nbr, o_nbr), "neighbours (order inside rows included)")
        if keep_rowid:
            check(np.array_equal(eid, o_eid), "edge rowids")
        V, E = og.V, og.E
        deg = np.diff(o_off).astype(np.float64)
        indeg = np.bincount(o_nbr, minlength=V).astype(np.float64) if V else np.zeros(0)

        def walks(k, w0):
            """number of h-hop walks for h = 1..k from source weights w0 (float estimate, to bound the oracle's work)"""
            out, w = [], w0
            for _ in range(k):
                out.append(float((w * deg).sum()))
                w = np.bincount(o_nbr, weights=np.repeat(w, np.diff(o_off)), minlength=V) if E else np.zeros(V)
            return out

        ops = rng.permutation(["khop_all", "khop_list", "materialise", "bfs", "join", "mid", "edges", "endpoints"])
        for op in ops[: int(rng.integers(2, 6))]:
            note["op"] = str(op)
            force = int(rng.choice([0, 0, 1, 2, 3]))
            gg.force_frontier(force)
            note["force"] = force
            if op in ("khop_all", "khop_list", "materialise", "edges"):
                if op == "khop_all" or (op != "khop_list" and rng.random() < 0.5):
                    sources, dense, w0 = None, None, np.ones(V)
                else:
                    n_src = log_uniform(rng, 1, max(1, min(V, 20_000)))
                    sources = vtab[rng.integers(0, max(V, 1), n_src)] if V else np.zeros(0, np.int64)
                    if rng.random() < 0.5:
                        sources = np.concatenate([sources, np.array([12345678901, -77], np.int64), sources[:3]])
                    dense = dense_of(o_vid, sources)
                    w0 = np.bincount(dense, minlength=V).astype(np.float64) if V else np.zeros(0)
                k_max = int(rng.integers(1, 5))
                est = walks(k_max, w0) if V else [0.0] * k_max
                while k_max > 1 and sum(est[:k_max]) > 4e7:
                    k_max -= 1
                if sum(est[:k_max]) > 4e7:
                    continue
                k_min = int(rng.integers(1, k_max + 1))
                note.update(k_min=k_min, k_max=k_max, n_sources=None if sources is None else int(sources.size))
                ref = og.khop(k_min, k_max, sources_dense=dense)
                cnt = gg.khop_count(csr, k_min, k_max, sources=sources)
                check(cnt[k_min:k_max + 1] == ref["rows"][k_min:k_max + 1], f"khop_count {cnt} vs {ref['rows']}")
                if op in ("khop_all", "khop_list"):
                    got = gg.expand_khop(csr, k_min, k_max, sources=sources)
                    check(got == ref, f"expand_khop {got} vs {ref}")
                elif op == "materialise":
                    total = sum(ref["rows"][k_min:k_max + 1])
                    if total <= 1_500_000:
                        got = gg.expand_khop(csr, k_min, k_max, sources=sources, materialise=True)
                        tables = got.pop("tables")
                        check(got == ref, f"expand_khop(materialise) stats {got} vs {ref}")
                        rows = og.khop_rows(k_min, k_max, sources_dense=dense)
                        for h in range(k_min, k_max + 1):
                            check(np.array_equal(sort_rows(tables[h]), sort_rows(rows[h])), f"materialised rows, h={h}")
                    else:
                        res = gg.expand_khop_result(csr, k_max, sources=sources)
                        n, dg = res.digest(csr, k_max)
                        res.close()
                        one = og.khop(k_max, k_max, sources_dense=dense)
                        check((n, dg) == (one["rows"][k_max], one["digest"][k_max]), f"result digest h={k_max}")
                elif op == "edges" and keep_rowid:
                    k = k_max
                    if ref["rows"][k] <= 300_000 or k_min == k:
                        one = og.khop(k, k, sources_dense=dense)
                        if one["rows"][k] <= 300_000:
                            res = gg.expand_khop_edges(csr, k, sources=sources)
                            n = res.rows(k)
                            check(n == one["rows"][k], "walks with edges: row count")
                            walk = np.concatenate([res.fetch(k, o) for o in range(0, n, 1024)]) if n else np.zeros((0, k + 1), np.int64)
                            edges = np.concatenate([res.fetch_edges(k, o) for o in range(0, n, 1024)]) if n else np.zeros((0, k), np.int64)
                            res.close()
                            # every edge rowid names a row of the edge table that joins the two vertices beside it
                            rid = np.arange(src.size, dtype=np.int64) if rowid is None else rowid
                            pos = {int(r): i for i, r in enumerate(rid.tolist())} if n and src.size <= 400_000 else None
                            if pos is not None:
                                for j in range(k):
                                    at = np.array([pos[int(r)] for r in edges[:, j].tolist()], np.int64)
                                    check(np.array_equal(src[at], walk[:, j]) and np.array_equal(dst[at], walk[:, j + 1]),
                                          f"edge rowid column {j}")
                            rows = og.khop_rows(k, k, sources_dense=dense)[k]
                            check(np.array_equal(sort_rows(walk), sort_rows(rows)), "walks with edges: vertex columns")
            elif op == "bfs" and V:
                n_src = int(rng.integers(1, 65))
                sources = vtab[rng.integers(0, V, n_src)]
                if rng.random() < 0.3 and not np.isin(424242424242, vtab):
                    sources[-1] = 424242424242  # a source that is no vertex
                max_hops = int(rng.choice([-1, 0, 1, 2, 3, 5, 8]))
                note.update(n_sources=n_src, max_hops=max_hops)
                if V * n_src <= 8_000_000:
                    dist, st = gg.bfs64(csr, sources, max_hops)
                    o_dist, o_st = og.bfs64(dense_lookup(o_vid, sources), max_hops)
                    check(np.array_equal(dist, o_dist), "bfs64 distances")
                    check(st == o_st, f"bfs64 stats {st} vs {o_st}")
            elif op == "join" and keep_rowid:
                m = log_uniform(rng, 1, 50_000)
                keys = vtab[rng.integers(0, max(V, 1), m)] if V else np.zeros(0, np.int64)
                keys = np.concatenate([keys, np.array([-31337], np.int64)])
                est = float(deg[dense_of(o_vid, keys)].sum()) if V else 0.0
                if est <= 3e5:
                    got = gg.join_probe(csr, keys)
                    rid = np.arange(src.size, dtype=np.int64) if rowid is None else rowid
                    order = np.argsort(src, kind="stable")
                    ss = src[order]
                    lo, hi = np.searchsorted(ss, keys, "left"), np.searchsorted(ss, keys, "right")
                    if not edge_only:  # rows whose destination is no vertex were dropped by the build
                        okrow = np.isin(dst, vtab) & np.isin(src, vtab)
                    else:
                        okrow = np.ones(src.size, bool)
                    exp = [(i, int(rid[order[j]])) for i in range(keys.size) for j in range(lo[i], hi[i]) if okrow[order[j]]]
                    exp = np.array(exp, np.int64).reshape(-1, 2)
                    check(np.array_equal(sort_rows(got), sort_rows(exp)), "join_probe pairs")
            elif op == "mid" and V:
                k_min = int(rng.integers(1, 3))
                est = float((deg * indeg).sum())
                if est <= 4e7:
                    counted = gg.expand_khop(csr, k_min, 2)
                    cuts = np.unique(np.concatenate([[0, V], rng.integers(0, V + 1, int(rng.integers(0, 5)))]))
                    tot = {1: [0, 0], 2: [0, 0]}
                    for lo, hi in zip(cuts[:-1], cuts[1:]):
                        res = gg.expand_khop_mid_result(csr, int(lo), int(hi), k_min=k_min, with_stats=bool(rng.random() < 0.5))
                        for h in range(k_min, 3):
                            n, dg = res.digest(csr, h)
                            check(n == res.rows(h), "mid result rows")
                            tot[h][0] += n
                            tot[h][1] = (tot[h][1] + dg) & 0xFFFFFFFF
                        res.close()
                    for h in range(k_min, 3):
                        check(tot[h] == [counted["rows"][h], counted["digest"][h]], f"middle-vertex parts, h={h}")
                    ref = og.khop(k_min, 2)
                    check(counted == ref, "expand_khop vs oracle (mid)")
            elif op == "endpoints" and V:
                n_src = log_uniform(rng, 1, max(1, min(V, 2000)))
                sources = vtab[rng.integers(0, V, n_src)]
                k = int(rng.integers(1, 4))
                dense = dense_of(o_vid, sources)
                w0 = np.bincount(dense, minlength=V).astype(np.float64)
                if sum(walks(k, w0)) <= 2e7:
                    ids, masks = gg.walk_endpoints(csr, sources, k)
                    want = np.zeros(V, np.int64)
                    front = np.zeros(V, bool)
                    front[dense] = True
                    row_of = np.repeat(np.arange(V), np.diff(o_off))
                    for h in range(1, k + 1):
                        nxt = np.zeros(V, bool)
                        nxt[o_nbr[front[row_of]]] = True
                        want[nxt] |= 1 << h
                        front = nxt
                    keep = np.flatnonzero(want)
                    check(np.array_equal(ids, o_vid[keep]), "walk_endpoints ids (vertex order)")
                    check(np.array_equal(masks, want[keep]), "walk_endpoints masks")
    finally:
        csr.close()
        og.close()
        gg.debug_reset()
        gg.staging_clear()


def dense_lookup(o_vid, ids):
    """dense index per id, -1 for strangers (what OracleCsr.lookup returns, without a Python call per id)"""
    if o_vid.size == 0:
        return np.full(len(ids), -1, np.int64)
    order = np.argsort(o_vid, kind="stable")
    sv = o_vid[order]
    pos = np.minimum(np.searchsorted(sv, ids), sv.size - 1)
    return np.where(sv[pos] == ids, order[pos], -1).astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=420.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--iterations", type=int, default=0)
    ap.add_argument("--max-rows", type=int, default=3_000_000)
    ap.add_argument("--max-vertices", type=int, default=300_000)
    ap.add_argument("--min-frac", type=float, default=0.0, help="sizes are drawn log-uniformly from [min_frac x max, max]")
    a = ap.parse_args()
    orc = oracle_lib.load()
    gg = pkg.GG(0)
    t0 = last = time.time()
    i = 0
    while (a.iterations and i < a.iterations) or (not a.iterations and time.time() - t0 < a.seconds):
        seed = a.seed * 1_000_003 + i
        note = {"seed": a.seed, "iteration": i}
        try:
            one_iteration(gg, orc, np.random.default_rng(seed), a, note)
        except Exception:
            print(f"FAILED at iteration {i} (rerun: --seed {a.seed} --iterations {i + 1}): {note}", flush=True)
            traceback.print_exc()
            sys.exit(1)
        i += 1
        if time.time() - last > 20:
            last = time.time()
            print(f"[{last - t0:6.0f} s] {i} iterations, last: {note}", flush=True)
    print(f"fuzz ok: {i} iterations in {time.time() - t0:.0f} s (seed {a.seed})", flush=True)
    gg.close()


if __name__ == "__main__":
    main()
