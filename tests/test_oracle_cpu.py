"""CPU tests of the oracle itself: the join/CTE restatement of the reference operators against the
reference's own golden rows, and against the direct CSR formulation used for full-size inputs."""
import numpy as np
import pytest

from duckdb_pgq_amd import datagen
from tests import trainbenchmark as tb
from tests.oracle_lib import sort_rows


def test_jht_chain_prepend_order(orc):
    # duplicates of one key chain newest-first (join_hashtable.cpp:251-259); within a probe chunk all
    # first matches are emitted before second matches (ScanStructure::NextInnerJoin)
    build = np.array([5, 7, 5, 5, 9], np.int64)
    probe = np.array([5, 9, 4, 5], np.int64)
    m = orc.hash_join(build, probe)
    assert m.tolist() == [[0, 3], [1, 4], [3, 3], [0, 2], [3, 2], [0, 0], [3, 0]]


def test_jht_long_chain(orc):
    # test/sql/join/inner/test_join_duplicates.test:14-24 — a 10 240-long duplicate chain
    build = np.full(10240, 1, np.int64)
    probe = np.array([1, 2, 1], np.int64)
    m = orc.hash_join(build, probe)
    assert m.shape[0] == 2 * 10240
    assert set(m[:, 0].tolist()) == {0, 2}
    assert sorted(m[m[:, 0] == 0, 1].tolist()) == list(range(10240))


def test_connectedsegments_golden(orc):
    # the only golden vector on graph data in the reference:
    # benchmark/trainbenchmark/connectedsegments.benchmark:34-38
    rows = tb.connectedsegments_via_joins(orc, tb.tables())
    assert sort_rows(rows).tolist() == sort_rows(tb.CONNECTEDSEGMENTS_GOLDEN).tolist()


@pytest.mark.parametrize("V,E,seed,dangling,dup", [(1, 0, 1, 0, 0), (10, 40, 2, 0, 0), (50, 400, 3, 6, 20), (200, 1500, 4, 10, 0), (7, 60, 5, 0, 30)])
@pytest.mark.parametrize("k", [(1, 1), (1, 2), (2, 2), (1, 3), (3, 3)])
def test_khop_join_equals_csr(orc, V, E, seed, dangling, dup, k):
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=dangling, dup_edges=dup)
    rc, g = orc.csr_build(vid, src, dst)
    assert rc == 0
    kmin, kmax = k
    j = orc.khop_join(vid, src, dst, kmin, kmax)
    c = g.khop_rows(kmin, kmax)
    st = g.khop(kmin, kmax, threads=2)
    for h in range(kmin, kmax + 1):
        jr = vid[j[h]] if j[h].size else j[h]
        assert sort_rows(jr).tolist() == sort_rows(c[h]).tolist()
        assert st["rows"][h] == j[h].shape[0]
        assert st["digest"][h] == orc.digest_rows(j[h])
    g.close()


def test_khop_sources_subset_and_missing(orc):
    vid, src, dst = datagen.small_graph(60, 500, 11, dangling=4)
    rc, g = orc.csr_build(vid, src, dst)
    assert rc == 0
    sources = np.concatenate([vid[[3, 3, 17, 59]], np.array([123456789, -5], np.int64)])
    j = orc.khop_join(vid, src, dst, 1, 2, sources=sources)
    dense = g.lookup(sources)
    dense = dense[dense >= 0].astype(np.uint32)
    st = g.khop(1, 2, sources_dense=dense)
    for h in (1, 2):
        assert st["rows"][h] == j[h].shape[0]
        assert st["digest"][h] == orc.digest_rows(j[h])
    g.close()


def test_duplicate_vertex_rejected(orc):
    rc, g = orc.csr_build(np.array([1, 2, 1], np.int64), np.array([1], np.int64), np.array([2], np.int64))
    assert rc == -4
    g.close()


@pytest.mark.parametrize("V,E,seed,max_hops", [(30, 60, 1, 5), (100, 300, 2, 3), (100, 300, 2, 0), (64, 2000, 3, 2), (40, 30, 4, 6)])
def test_cte_shortest_equals_bfs(orc, V, E, seed, max_hops):
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=3)
    rc, g = orc.csr_build(vid, src, dst)
    assert rc == 0
    sources = np.concatenate([datagen.pick_sources(vid, 10, seed), np.array([999], np.int64), vid[:1]])
    ref = orc.cte_shortest(vid, src, dst, sources, max_hops)
    dense = g.lookup(sources)
    dist, st = g.bfs64(dense, max_hops)
    got = set()
    for i, s in enumerate(sources):
        for v in np.nonzero(dist[i] >= 0)[0]:
            got.add((int(s), int(vid[v]), int(dist[i, v])))
    assert got == {tuple(r) for r in ref.tolist()}
    g.close()


def test_datagen_deterministic():
    a = datagen.ldbc_knows(500, 6000, 42)
    b = datagen.ldbc_knows(500, 6000, 42)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    vid, src, dst = a
    assert np.unique(vid).size == vid.size
    half = src.size // 2
    assert np.array_equal(src[:half], dst[half:]) and np.array_equal(dst[:half], src[half:])
    assert not np.any(src == dst)


def test_recursive_cte_restatement_on_the_references_own_vectors(orc):
    """test/sql/cte/test_recursive_cte_union.test written as graphs (x -> x+1 is an edge of a path graph):
    :8-14   `select 1 union select x+1 from t where x < 3`      -> 1, 2, 3
    :40-44  `select 1 union select x from t` (UNION dedupe stops the self-reference) -> 1
    :54-60  two references, `m.x + f.x ... where m.x < 3`       -> not a walk pattern (not restated)
    The friends CTE carries (start, hop, vertex); its min(hop) per vertex is what orc_cte_shortest returns."""
    vid = np.array([1, 2, 3, 4, 5], np.int64)
    # path 1 -> 2 -> 3 -> 4 -> 5, recursion allowed while x < 3 i.e. two steps from the seed 1
    got = orc.cte_shortest(vid, vid[:-1], vid[1:], np.array([1], np.int64), 2)
    assert sort_rows(got).tolist() == [[1, 1, 0], [1, 2, 1], [1, 3, 2]]
    # self-reference x -> x: the vertex is reached once, at hop 0, however long the recursion may run
    loop = orc.cte_shortest(np.array([1], np.int64), np.array([1], np.int64), np.array([1], np.int64),
                            np.array([1], np.int64), 50)
    assert loop.tolist() == [[1, 1, 0]]
    # a cycle 1 -> 2 -> 3 -> 1 terminates by dedupe too and keeps the shortest hop of every vertex
    cyc = orc.cte_shortest(vid[:3], vid[:3], np.roll(vid[:3], -1), np.array([2], np.int64), 50)
    assert sort_rows(cyc).tolist() == [[2, 1, 2], [2, 2, 0], [2, 3, 1]]


def test_threaded_csr_build_of_the_oracle_is_the_stable_counting_sort(orc):
    """orc_csr_build splits the edge rows over threads above 100 000 rows; the CSR must still be the stable
    counting sort by source (ascending edge position inside a row), here re-derived with numpy."""
    import numpy as np

    from duckdb_pgq_amd import datagen

    vid, src, dst = datagen.ldbc_knows(5000, 400_000, 17)
    src[::97] = -5  # dangling rows are dropped
    rc, g = orc.csr_build(vid, src, dst)
    assert rc == 0
    off, nbr, eid, v2 = g.arrays()
    order = np.argsort(vid, kind="stable")
    pos = np.searchsorted(vid[order], src)
    ok = (pos < vid.size) & (vid[order][np.minimum(pos, vid.size - 1)] == src)
    u = order[np.minimum(pos, vid.size - 1)][ok]
    v = order[np.searchsorted(vid[order], dst[ok])]
    k = np.argsort(u, kind="stable")
    assert np.array_equal(nbr, v[k]) and np.array_equal(eid, np.flatnonzero(ok)[k])
    assert np.array_equal(off, np.concatenate([[0], np.cumsum(np.bincount(u, minlength=vid.size))]))
    assert g.dropped == int((~ok).sum())
    g.close()


def test_khop_join_equals_csr_on_the_fuzzers_graph_shapes(orc):
    """The two restatements inside the oracle — the reference's join hash table run as k chained joins, and the CSR
    formulation every GPU result is compared with — on the graph shapes scripts/fuzz_gg.py draws (ids anywhere in int64,
    negative, clustered, dense; power-law, hub, chain and sorted edge tables; duplicate rows): same rows, counts and
    digests."""
    import importlib.util
    import os

    import numpy as np

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_gg.py")
    spec = importlib.util.spec_from_file_location("fuzz_gg_generator", path)
    fuzz = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(fuzz)  # (imports the package: the library itself is only loaded by GG(), not here)
    except Exception as e:  # pragma: no cover
        pytest.skip(f"fuzz_gg not importable here: {e}")
    seen = set()
    for i in range(40):
        rng = np.random.default_rng(1000 + i)
        g = fuzz.draw_graph(rng, max_rows=600, max_vertices=120)
        vid, src, dst = g["vid"], g["src"], g["dst"]
        seen.add((g["ids"], g["model"]))
        rc, c = orc.csr_build(vid, src, dst)
        assert rc == 0
        deg = max(1.0, src.size / max(1, vid.size))
        kmax = 3 if src.size * deg * deg <= 300_000 else 2
        j = orc.khop_join(vid, src, dst, 1, kmax)
        rows = c.khop_rows(1, kmax)
        st = c.khop(1, kmax)
        for h in range(1, kmax + 1):
            jr = vid[j[h]] if j[h].size else j[h]
            assert sort_rows(jr).tolist() == sort_rows(rows[h]).tolist(), (i, g["ids"], g["model"], h)
            assert st["rows"][h] == j[h].shape[0] and st["digest"][h] == orc.digest_rows(j[h])
        c.close()
    assert len(seen) >= 12  # (the draw reached most (id shape, degree model) combinations)


def test_cte_shortest_equals_bfs_on_the_fuzzers_graph_shapes(orc):
    """The recursive-CTE restatement (UNION dedupe + min over hop counts, as the reference executes bi-10's `friends`)
    against the oracle's 64-lane bitset BFS — the form the GPU BFS is compared with — on the fuzzer's graph shapes."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_gg.py")
    spec = importlib.util.spec_from_file_location("fuzz_gg_generator", path)
    fuzz = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(fuzz)
    except Exception as e:  # pragma: no cover
        pytest.skip(f"fuzz_gg not importable here: {e}")
    for i in range(25):
        rng = np.random.default_rng(2000 + i)
        g = fuzz.draw_graph(rng, max_rows=400, max_vertices=80)
        vid, src, dst = g["vid"], g["src"], g["dst"]
        rc, c = orc.csr_build(vid, src, dst)
        assert rc == 0
        sources = vid[rng.integers(0, vid.size, int(rng.integers(1, 12)))]
        sources = np.concatenate([sources, sources[:1]])  # (a duplicate seed: UNION keeps one)
        max_hops = int(rng.choice([0, 1, 2, 4, 7]))
        ref = orc.cte_shortest(vid, src, dst, sources, max_hops)
        uniq = np.array(list(dict.fromkeys(sources.tolist())), np.int64)
        dist, _ = c.bfs64(c.lookup(uniq), max_hops)
        got = {(int(s), int(vid[v]), int(dist[k, v])) for k, s in enumerate(uniq) for v in np.nonzero(dist[k] >= 0)[0]}
        assert got == {tuple(r) for r in ref.tolist()}, (i, g["ids"], g["model"], max_hops)
        c.close()
