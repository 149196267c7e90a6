"""GPU parity tests: the HIP path (through the C-ABI, libgg.so) against the CPU oracle on the same
seeded inputs.  Bar: bit-exact (all integer work).  Run on the GPU box with `-m gpu`."""
import numpy as np
import pytest

from duckdb_pgq_amd import datagen
from tests.oracle_lib import sort_rows

pytestmark = pytest.mark.gpu


def dsum(a, b):
    """digest sums are 32-bit (Row digest)"""
    return ((a & 0xFFFFFFFF) + (b & 0xFFFFFFFF)) & 0xFFFFFFFF


def build_both(gg, orc, vid, src, dst, rowid=None, chunk_rows=0):
    gg.staging_clear()
    old = gg.chunk_rows
    gg.chunk_rows = chunk_rows
    gg.append_vertices(vid)
    gg.append_edges(src, dst, rowid)
    gg.chunk_rows = old
    csr = gg.build_csr()
    rc, g = orc.csr_build(vid, src, dst, rowid)
    assert rc == 0
    return csr, g


def assert_csr_equal(csr, g):
    off, nbr, eid, vid = csr.export()
    o_off, o_nbr, o_eid, o_vid = g.arrays()
    assert csr.V == g.V and csr.E == g.E and csr.dropped == g.dropped
    assert np.array_equal(off, o_off)
    assert np.array_equal(vid, o_vid)
    assert np.array_equal(nbr, o_nbr)  # bit-exact INCLUDING order inside rows (stable build)
    assert np.array_equal(eid, o_eid)


CASES = [
    # V, E, seed, dangling, dup
    (1, 0, 1, 0, 0),
    (1, 5, 2, 0, 0),        # self loops only
    (2, 1, 3, 0, 0),
    (10, 40, 4, 0, 0),
    (50, 400, 5, 6, 20),
    (300, 5000, 6, 10, 100),
    (2049, 30000, 7, 0, 0),  # > 2048 vertices: two radix passes
    (5000, 200000, 8, 50, 0),
]


@pytest.mark.parametrize("V,E,seed,dangling,dup", CASES)
def test_csr_build_bit_exact(gg, orc, V, E, seed, dangling, dup):
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=dangling, dup_edges=dup)
    csr, g = build_both(gg, orc, vid, src, dst)
    assert_csr_equal(csr, g)
    csr.close()
    g.close()


def test_csr_empty_tables(gg, orc):
    z = np.zeros(0, np.int64)
    csr, g = build_both(gg, orc, z, z, z)
    assert csr.V == 0 and csr.E == 0
    st = gg.expand_khop(csr, 1, 2)
    assert st["rows"][1] == 0 and st["rows"][2] == 0 and st["traversed_edges"] == 0
    csr.close()
    g.close()
    # vertices but no edges
    vid = datagen.person_ids(100, 3)
    csr, g = build_both(gg, orc, vid, z, z)
    assert_csr_equal(csr, g)
    st = gg.expand_khop(csr, 1, 2)
    assert st["rows"][1] == 0 and st["traversed_edges"] == 0 and st["frontier_entries"] == 100
    dist, bst = gg.bfs64(csr, vid[:3], 5)
    assert (dist >= 0).sum() == 3
    csr.close()
    g.close()


def test_csr_rowid_and_chunked_append(gg, orc):
    """Sink-style 1024-row appends with explicit (non-monotone) rowids; ragged last chunk."""
    vid, src, dst = datagen.small_graph(700, 10_000 + 37, 21, dangling=5)
    rowid = (np.arange(src.size, dtype=np.int64) * 7919) % 1_000_003
    csr, g = build_both(gg, orc, vid, src, dst, rowid=rowid, chunk_rows=1024)
    assert_csr_equal(csr, g)
    csr.close()
    g.close()


def test_csr_sentinel_and_extreme_ids(gg, orc):
    vid = np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0, -1, 42], np.int64)
    src = np.array([vid[0], vid[1], vid[0], 0, 42, 42, 7], np.int64)
    dst = np.array([vid[1], vid[0], vid[0], -1, vid[0], 42, 0], np.int64)
    csr, g = build_both(gg, orc, vid, src, dst)
    assert_csr_equal(csr, g)
    assert csr.dropped == 1
    csr.close()
    g.close()


def test_duplicate_vertex_is_an_error(gg):
    from duckdb_pgq_amd import GGError

    gg.staging_clear()
    gg.append_vertices(np.array([1, 2, 3, 2], np.int64))
    gg.append_edges(np.array([1], np.int64), np.array([2], np.int64))
    with pytest.raises(GGError) as e:
        gg.build_csr()
    assert e.value.code == -4


def test_csr_build_repeatable(gg, orc):
    """The staged columns stay resident: building twice gives the same bits (no atomics decide a slot)."""
    vid, src, dst = datagen.ldbc_knows(3000, 120_000, 99)
    csr, g = build_both(gg, orc, vid, src, dst)
    a = csr.export()
    csr2 = gg.build_csr()
    b = csr2.export()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert_csr_equal(csr2, g)
    csr.close()
    csr2.close()
    g.close()


@pytest.mark.parametrize("V,E,seed,dangling,dup", CASES[3:])
@pytest.mark.parametrize("k", [(1, 1), (1, 2), (2, 2), (1, 3), (2, 4)])
def test_khop_count_digest(gg, orc, V, E, seed, dangling, dup, k):
    if k[1] >= 3 and E > 30000:
        pytest.skip("oracle enumeration too large")
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=dangling, dup_edges=dup)
    csr, g = build_both(gg, orc, vid, src, dst)
    got = gg.expand_khop(csr, k[0], k[1])
    ref = g.khop(k[0], k[1])
    assert got == ref
    csr.close()
    g.close()


def test_khop_materialised_rows_equal_join_formulation(gg, orc):
    """Materialised rows (fetched in <=1024-row slices) equal the rows the reference's hash-join chain
    produces (oracle restatement of JoinHashTable), as a sorted multiset."""
    vid, src, dst = datagen.small_graph(120, 900, 31, dangling=6, dup_edges=15)
    csr, g = build_both(gg, orc, vid, src, dst)
    got = gg.expand_khop(csr, 1, 3, materialise=True)
    ref = orc.khop_join(vid, src, dst, 1, 3)
    for h in (1, 2, 3):
        r = vid[ref[h]] if ref[h].size else ref[h]
        assert got["tables"][h].shape == r.shape
        assert np.array_equal(sort_rows(got["tables"][h]), sort_rows(r))
        assert got["digest"][h] == orc.digest_rows(ref[h])
    csr.close()
    g.close()


def test_digest_of_materialised_rows_equals_the_count_mode_digest(gg, orc):
    """gg_result_digest reads the id columns a materialising expansion left in HBM, maps every id back to its dense
    index and sums the row hashes: rows and digest must be the count-only expansion's and the oracle's for the same
    walks (1..4 hops, all sources and a source list; dangling rows and duplicate edges in the table)."""
    vid, src, dst = datagen.small_graph(300, 2500, 41, dangling=9, dup_edges=30)
    csr, g = build_both(gg, orc, vid, src, dst)
    sources = np.concatenate([datagen.pick_sources(vid, 40, 2), vid[:3], np.array([-9], np.int64)])
    dense = g.lookup(sources)
    dense = dense[dense >= 0].astype(np.uint32)
    for k in (1, 2, 3, 4):
        for srcs, sd in ((None, None), (sources, dense)):
            res = gg.expand_khop_result(csr, k, sources=srcs)
            want = g.khop(k, k, sources_dense=sd)
            counted = gg.expand_khop(csr, k, k, sources=srcs)
            n, dig = res.digest(csr, k)
            assert n == res.rows(k) == want["rows"][k] == counted["rows"][k], (k, srcs is None)
            assert dig == want["digest"][k] == counted["digest"][k], (k, srcs is None)
            res.close()
    csr.close()
    g.close()


def test_two_hop_rows_by_middle_vertex_ranges_partition_the_materialised_result(gg, orc):
    """The product form of the materialising expansion (k_mat_mid2: rows grouped by middle vertex, the ids of out(x)
    gathered once per run): the whole result equals the join formulation's rows as a sorted multiset; middle-vertex
    ranges partition it (rows and digests add up, the 1-hop tables are the edges into each range); odd and even row
    blocks, out-rows on both sides of the flat form's 1024 leaves, vertices without out- or in-edges."""
    rng = np.random.default_rng(5)
    V, E = 700, 16000
    vid = np.arange(V, dtype=np.int64) * 7 + 11
    s, d = rng.integers(0, V - 50, E), rng.integers(0, V - 50, E)  # the last 50 vertices stay isolated
    s[:600] = 3   # a hub with an out-row of ~600 leaves
    d[600:1100] = 5  # and one with ~500 in-edges
    # out-rows right at the boundaries of the store forms: one row, one lane pair, a 128-leaf block of the long form,
    # the longest row the flat form takes (1024 leaves, staged in LDS) and the first ones of the long form
    special = ((40, 127, 1200), (41, 128, 1400), (42, 129, 1600), (43, 1, 1800), (44, 2, 1810), (45, 1023, 2000),
               (46, 1024, 3100), (47, 1025, 4200), (48, 2500, 5300))
    keep = ~np.isin(s, [v for v, _, _ in special])
    for vtx, deg, at in special:
        s[at:at + deg] = vtx
        keep[at:at + deg] = True
    s, d = s[keep], d[keep]
    d[:9] = [v for v, _, _ in special]  # every one of them has an in-edge, so their products are not empty
    d[9:13] = (47, 47, 48, 48)  # (several entries per run for the long form, an odd and an even number)
    d[13] = 48
    src, dst = vid[s], vid[d]
    csr, g = build_both(gg, orc, vid, src, dst)
    ref = orc.khop_join(vid, src, dst, 1, 2)
    whole = gg.expand_khop(csr, 1, 2, materialise=True)
    for h in (1, 2):
        assert np.array_equal(sort_rows(whole["tables"][h]), sort_rows(vid[ref[h]]))
    counted = gg.expand_khop(csr, 1, 2)
    bounds = [0, 4, 5, 6, 300, 301, V]
    rows = {1: [], 2: []}
    dig = {1: 0, 2: 0}
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        res = gg.expand_khop_mid_result(csr, lo, hi, k_min=1)
        for h in (1, 2):
            n, dgst = res.digest(csr, h)
            assert n == res.rows(h) == res.stats["rows"][h] and dgst == res.stats["digest"][h]
            dig[h] = (dig[h] + dgst) & 0xFFFFFFFF
            got = [res.fetch(h, o) for o in range(0, n, 1024)]
            if got:
                rows[h].append(np.concatenate(got))
        # the same range without the counting expansion in front (stats == NULL): the same rows
        bare = gg.expand_khop_mid_result(csr, lo, hi, k_min=1, with_stats=False)
        assert bare.stats is None
        for h in (1, 2):
            assert bare.digest(csr, h) == (res.rows(h), res.stats["digest"][h]) and bare.rows(h) == res.rows(h)
        bare.close()
        res.close()
    for h in (1, 2):
        allrows = np.concatenate(rows[h])
        assert np.array_equal(sort_rows(allrows), sort_rows(vid[ref[h]]))
        assert dig[h] == counted["digest"][h]
    csr.close()
    g.close()


def test_walks_with_their_edge_rowids(gg, orc):
    """gg_expand_khop_edges: every walk comes with the rowid of each edge it takes (what a late join with the edge
    table's payload columns needs).  Checked on a multigraph with explicit, non-contiguous rowids: the vertex columns
    are the join formulation's rows; edge j of every row really is a table row (v_{j-1}, v_j); no edge sequence
    repeats; with parallel edges the number of rows is the product of the multiplicities; all sources and a source
    list; and append positions stand in when the Sink passed no rowids."""
    vid, src, dst = datagen.small_graph(200, 1500, 77, dangling=5, dup_edges=40)
    rowid = (np.arange(src.size, dtype=np.int64) * 7 + 1000)[::-1].copy()  # explicit, not the append order
    gg.staging_clear()
    gg.set_edge_rowid(True)
    gg.append_vertices(vid)
    gg.append_edges(src, dst, rowid=rowid)
    csr = gg.build_csr()
    by_rowid = {int(r): (int(a), int(b)) for r, a, b in zip(rowid, src, dst)}
    sources = np.concatenate([datagen.pick_sources(vid, 25, 4), np.array([-3], np.int64)])
    for k in (1, 2, 3):
        ref = vid[orc.khop_join(vid, src, dst, k, k)[k]]
        for srcs in (None, sources):
            res = gg.expand_khop_edges(csr, k, sources=srcs)
            n = res.rows(k)
            v = np.concatenate([res.fetch(k, o) for o in range(0, n, 1024)]) if n else np.zeros((0, k + 1), np.int64)
            e = np.concatenate([res.fetch_edges(k, o) for o in range(0, n, 1024)]) if n else np.zeros((0, k), np.int64)
            res.close()
            want = ref if srcs is None else ref[np.isin(ref[:, 0], srcs)]
            assert np.array_equal(sort_rows(v), sort_rows(want)), (k, srcs is None)
            for j in range(k):
                ends = np.array([by_rowid[int(r)] for r in e[:, j]], np.int64).reshape(-1, 2)
                assert np.array_equal(ends, v[:, j:j + 2]), (k, j)
            assert np.unique(e, axis=0).shape[0] == e.shape[0]
    csr.close()
    # no explicit rowids: the append position is the edge's id
    gg.staging_clear()
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    csr = gg.build_csr()
    res = gg.expand_khop_edges(csr, 2)
    n = res.rows(2)
    v = np.concatenate([res.fetch(2, o) for o in range(0, n, 1024)])
    e = np.concatenate([res.fetch_edges(2, o) for o in range(0, n, 1024)])
    res.close()
    for j in range(2):
        assert np.array_equal(np.stack([src[e[:, j]], dst[e[:, j]]], axis=1), v[:, j:j + 2])
    csr.close()
    gg.set_edge_rowid(False)
    gg.staging_clear()
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    csr = gg.build_csr()
    with pytest.raises(Exception):
        gg.expand_khop_edges(csr, 1)
    csr.close()


def test_khop_source_list(gg, orc):
    vid, src, dst = datagen.ldbc_knows(2000, 60_000, 5)
    csr, g = build_both(gg, orc, vid, src, dst)
    sources = np.concatenate([datagen.pick_sources(vid, 300, 8), vid[:5], vid[:5], np.array([-1, 77], np.int64)])
    got = gg.expand_khop(csr, 1, 2, sources=sources, materialise=True)
    dense = g.lookup(sources)
    dense = dense[dense >= 0].astype(np.uint32)
    ref = g.khop(1, 2, sources_dense=dense)
    tables = got.pop("tables")
    assert got == ref
    rows = g.khop_rows(1, 2, sources_dense=dense)
    for h in (1, 2):
        assert np.array_equal(sort_rows(tables[h]), sort_rows(rows[h]))
    # empty / all-missing source lists
    assert gg.expand_khop(csr, 1, 2, sources=np.array([-3, -4], np.int64))["traversed_edges"] == 0
    assert gg.expand_khop(csr, 1, 2, sources=np.zeros(0, np.int64))["traversed_edges"] == 0
    csr.close()
    g.close()


@pytest.mark.parametrize("V,E,seed,dangling,dup", CASES[1:] + [(64, 6000, 41, 0, 200)])
def test_walk_counts_from_degrees_equal_the_counting_expansion(gg, orc, V, E, seed, dangling, dup):
    """gg_khop_count (what the planner's count(*) asks for: sums of degree products, no walk formed) against the rows
    of the counting expansion and of the oracle — every vertex as source, source lists with repeats and strangers,
    k up to 4, and ownership shards (the walks whose middle vertex is owned)."""
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=dangling, dup_edges=dup)
    csr, g = build_both(gg, orc, vid, src, dst)
    small = lambda k: E * (max(E, 1) / max(V, 1)) ** (k - 1) <= 2e8  # noqa: E731  (the oracle enumerates the walks)
    for k_min, k_max in [(1, 1), (1, 2), (2, 2), (1, 3), (3, 3), (2, 4)]:
        if not small(k_max):
            continue
        ref = g.khop(k_min, k_max)
        got = gg.khop_count(csr, k_min, k_max)
        assert got[k_min:k_max + 1] == ref["rows"][k_min:k_max + 1], (k_min, k_max)
        assert got[:k_min] == [0] * k_min and got[k_max + 1:] == [0] * (8 - k_max)
        if k_max <= 3:
            assert got == gg.expand_khop(csr, k_min, k_max)["rows"]
    sources = np.concatenate([vid[: max(1, V // 3)], vid[:2], np.array([-5, 123456789], np.int64)])
    dense = g.lookup(sources)
    dense = dense[dense >= 0].astype(np.uint32)
    for k_min, k_max in [(1, 1), (1, 2), (2, 3), (4, 4)]:
        if not small(k_max):
            continue
        ref = g.khop(k_min, k_max, sources_dense=dense)
        assert gg.khop_count(csr, k_min, k_max, sources=sources)[k_min:k_max + 1] == ref["rows"][k_min:k_max + 1]
    assert gg.khop_count(csr, 1, 3, sources=np.zeros(0, np.int64)) == [0] * 9
    assert gg.khop_count(csr, 1, 3, sources=np.array([-3], np.int64)) == [0] * 9
    csr.close()
    whole = g.khop(1, 2)["rows"]
    for parts in (2, 5):
        acc = [0] * 9
        for part in range(parts):
            sh = gg.build_csr_shard(part, parts)
            got = gg.khop_count(sh, 1, 2)
            assert got == gg.expand_khop(sh, 1, 2)["rows"]
            acc = [a + b for a, b in zip(acc, got)]
            sh.close()
        assert acc == whole
    g.close()


def test_join_probe_returns_every_matching_build_row_per_probe_key(gg, orc):
    """gg_join_probe: the device side of a generic single-key hash join — for a batch of probe keys the rowids of the
    build rows under each key, in probe order (JoinHashTable::Probe + NextInnerJoin for one chunk).  Duplicate probe
    keys, keys that match nothing, a 10 240-long duplicate chain (test_join_duplicates.test), an empty batch."""
    rng = np.random.default_rng(11)
    build_key = np.concatenate([rng.integers(-50, 50, 5000), np.full(10240, 7), np.array([1 << 40, -(1 << 40)])]).astype(np.int64)
    rowid = (np.arange(build_key.size, dtype=np.int64) * 3 + 1000)
    gg.staging_clear()
    gg.set_edge_rowid(True)
    gg.append_edges(build_key, build_key, rowid)
    gg.vertices_from_edges()
    csr = gg.build_csr()
    keys = np.concatenate([rng.integers(-60, 60, 3000), np.array([7, 7, 1 << 40, 12345, -(1 << 40)])]).astype(np.int64)
    got = gg.join_probe(csr, keys)
    order = np.argsort(build_key, kind="stable")
    sk, sr = build_key[order], rowid[order]
    lo, hi = np.searchsorted(sk, keys, "left"), np.searchsorted(sk, keys, "right")
    want = np.concatenate([np.stack([np.full(h - l, i, np.int64), sr[l:h]], axis=1) for i, (l, h) in enumerate(zip(lo, hi))])
    assert got.shape == want.shape and np.array_equal(got, want)  # (row order included: ascending position, rowid order)
    assert gg.join_probe(csr, np.zeros(0, np.int64)).shape == (0, 2)
    assert gg.join_probe(csr, np.array([99999], np.int64)).shape == (0, 2)
    csr.close()


def test_khop_ranges_partition_the_result(gg, orc):
    """Sharding entry point: per-range counts/digests add up to the whole (what the multi-GPU path sums)."""
    vid, src, dst = datagen.ldbc_knows(5000, 200_000, 17)
    csr, g = build_both(gg, orc, vid, src, dst)
    whole = gg.expand_khop(csr, 1, 2)
    assert whole == g.khop(1, 2)
    for parts in (2, 3, 8):
        b = gg.khop_partition(csr, parts)
        assert b[0] == 0 and b[-1] == csr.V and all(x <= y for x, y in zip(b, b[1:]))
        rows = [0, 0, 0]
        dig = [0, 0, 0]
        te = 0
        for lo, hi in zip(b, b[1:]):
            st = gg.expand_khop_range(csr, lo, hi, 1, 2)
            assert st == g.khop(1, 2, lo=lo, hi=hi)
            for h in (1, 2):
                rows[h] += st["rows"][h]
                dig[h] = dsum(dig[h], st["digest"][h])
            te += st["traversed_edges"]
        assert rows[1:] == whole["rows"][1:3] and dig[1:] == whole["digest"][1:3] and te == whole["traversed_edges"]
    csr.close()
    g.close()


def test_khop_high_degree_and_zero_degree_runs(gg, orc):
    """A hub with 5000 neighbours, followed by >256 isolated vertices (exercises the tile window
    fallback), then a chain."""
    V = 6000
    vid = datagen.person_ids(V, 77)
    hub = np.full(5000, vid[0])
    spokes = vid[1:5001]
    chain_s = vid[5500:5999]
    chain_d = vid[5501:6000]
    src = np.concatenate([hub, spokes[:100], chain_s])
    dst = np.concatenate([spokes, hub[:100], chain_d])
    csr, g = build_both(gg, orc, vid, src, dst)
    assert_csr_equal(csr, g)
    for k in [(1, 1), (1, 2), (2, 3)]:
        assert gg.expand_khop(csr, *k) == g.khop(*k)
    csr.close()
    g.close()


@pytest.mark.parametrize("V,E,seed,max_hops", [(30, 60, 1, 5), (100, 300, 2, 3), (100, 300, 2, 0), (64, 2000, 3, 2), (3000, 40000, 4, -1), (40, 30, 4, 6)])
def test_bfs64_equals_recursive_cte(gg, orc, V, E, seed, max_hops):
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=3)
    csr, g = build_both(gg, orc, vid, src, dst)
    sources = np.concatenate([datagen.pick_sources(vid, 61, seed), np.array([999], np.int64), vid[:2]])
    assert sources.size <= 64
    dist, st = gg.bfs64(csr, sources, max_hops)
    o_dist, o_st = g.bfs64(g.lookup(sources), max_hops)
    assert np.array_equal(dist, o_dist)
    assert st == o_st
    if max_hops >= 0 and E <= 2000:
        ref = orc.cte_shortest(vid, src, dst, sources, max_hops)  # the reference's relation
        got = {(int(sources[i]), int(vid[v]), int(dist[i, v])) for i in range(sources.size) for v in np.nonzero(dist[i] >= 0)[0]}
        assert got == {tuple(r) for r in ref.tolist()}
    # explicit target list, including ids that are not vertices
    targets = np.concatenate([vid[::7], np.array([-12345], np.int64)])
    d2, _ = gg.bfs64(csr, sources, max_hops, targets=targets)
    assert np.array_equal(d2[:, :-1], dist[:, ::7]) and np.all(d2[:, -1] == -1)
    csr.close()
    g.close()


def test_medium_ldbc_shape_end_to_end(gg, orc):
    """LDBC SF1-sized synthetic graph: CSR bit-exact, 2-hop count/digest, 64-source BFS."""
    vid, src, dst = datagen.ldbc("sf1")
    csr, g = build_both(gg, orc, vid, src, dst, chunk_rows=122_880)  # one row group per append
    assert_csr_equal(csr, g)
    assert gg.expand_khop(csr, 1, 2) == g.khop(1, 2)
    sources = datagen.pick_sources(vid, 64, 1)
    dist, st = gg.bfs64(csr, sources, -1)
    o_dist, o_st = g.bfs64(g.lookup(sources), -1)
    assert np.array_equal(dist, o_dist) and st == o_st
    csr.close()
    g.close()


@pytest.mark.parametrize("V,E,seed,dangling,dup", CASES[3:] + [(64, 6000, 41, 0, 200), (1000, 300000, 42, 0, 0)])
def test_khop_product_kernel_equals_frontier_kernel(gg, orc, V, E, seed, dangling, dup):
    """All-sources 2-hop and 3-hop counts run through the product kernels (2-hop: around the middle vertex;
    3-hop: {2-hop rows ending in b} x out(b)); they must agree bit-for-bit with the frontier kernels and with
    the oracle, for every k_min."""
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=dangling, dup_edges=dup)
    csr, g = build_both(gg, orc, vid, src, dst)
    for kmax in (2, 3):
        if kmax == 3 and E > 20000:  # (the oracle walks every 3-hop path)
            continue
        for kmin in range(1, kmax + 1):
            ref = g.khop(kmin, kmax)
            assert gg.expand_khop(csr, kmin, kmax) == ref, (kmin, kmax)
            gg.force_frontier(True)
            try:
                assert gg.expand_khop(csr, kmin, kmax) == ref, (kmin, kmax)
            finally:
                gg.force_frontier(False)
    csr.close()
    g.close()


@pytest.mark.parametrize("V,E,seed,dangling,dup", CASES[3:] + [(64, 6000, 41, 0, 200), (1000, 300000, 42, 0, 0)])
def test_last_hop_of_an_explicit_frontier_as_a_product(gg, orc, V, E, seed, dangling, dup):
    """Source lists, source ranges and walks of four hops end in a product form: k_expand_pairs + sort + k_expand_front
    (the second-to-last frontier as pairs grouped by last vertex: from 65 536 entries on; knob 2 at any size) or the
    frontier itself sorted by last vertex and k_expand_mid3's tiles over the reverse entries for the last TWO hops (when
    its children are an eighth of the edge table or more; knob 3 at any size).  Counts and digests must equal the
    frontier kernels' (knob 1) and the oracle's."""
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=dangling, dup_edges=dup)
    csr, g = build_both(gg, orc, vid, src, dst)
    sources = np.concatenate([vid[: max(1, V // 2)], vid[:3], np.array([-9], np.int64)])
    dense = g.lookup(sources)
    dense = dense[dense >= 0].astype(np.uint32)
    small = lambda k: E * (max(E, 1) / max(V, 1)) ** (k - 1) <= 1e8  # noqa: E731
    try:
        for kmin, kmax in [(1, 2), (2, 2), (1, 3), (3, 3), (2, 4)]:
            if not small(kmax):
                continue
            ref_all, ref_list = g.khop(kmin, kmax), g.khop(kmin, kmax, sources_dense=dense)
            lo, hi = V // 4, V - V // 5
            ref_range = g.khop(kmin, kmax, lo=lo, hi=hi)
            for knob in (3, 2, 1):
                gg.force_frontier(knob)
                assert gg.expand_khop(csr, kmin, kmax) == ref_all, (knob, kmin, kmax)
                assert gg.expand_khop(csr, kmin, kmax, sources=sources) == ref_list, (knob, kmin, kmax)
                assert gg.expand_khop_range(csr, lo, hi, kmin, kmax) == ref_range, (knob, kmin, kmax)
        # the same walks MATERIALISED: the last level through k_mat_front (knobs 2, 3: level k - 1 sorted by last
        # vertex, P + 2 id columns) against k_mat_last (knob 1) and the oracle's rows
        for kmin, kmax in [(1, 2), (2, 3), (4, 4)]:
            if not small(kmax) or E * (max(E, 1) / max(V, 1)) ** (kmax - 1) > 3e6:
                continue
            want = g.khop_rows(kmin, kmax, sources_dense=dense)
            for knob in (2, 1):
                gg.force_frontier(knob)
                got = gg.expand_khop(csr, kmin, kmax, sources=sources, materialise=True)["tables"]
                for h in range(kmin, kmax + 1):
                    assert np.array_equal(sort_rows(got[h]), sort_rows(want[h])), (knob, h)
    finally:
        gg.force_frontier(0)
    csr.close()
    g.close()


def test_a_large_source_list_takes_the_product_form_by_itself(gg, orc):
    """65 536 frontier entries and more: no knob.  3 000 sources of a graph with mean degree 60 have ~180 k 1-hop rows."""
    vid, src, dst = datagen.ldbc_knows(5000, 300_000, 77)
    csr, g = build_both(gg, orc, vid, src, dst)
    sources = vid[:3000]
    dense = g.lookup(sources).astype(np.uint32)
    got = gg.expand_khop(csr, 1, 2, sources=sources)
    assert got == g.khop(1, 2, sources_dense=dense)
    gg.profile_reset()
    gg.profile_select(None)
    gg.profile(True)
    gg.expand_khop(csr, 1, 2, sources=sources)
    gg.profile(False)
    prof = gg.profile_get()
    assert ("expand_front" in prof or "expand_front3" in prof) and "expand_fused2" not in prof
    csr.close()
    g.close()


def test_three_hop_product_kernel_on_a_skewed_graph(gg):
    """3-hop product kernel against the frontier kernels on a graph with hubs, isolated vertices and runs of
    reverse entries without 2-hop rows (windows of the flattened index that hold no children)."""
    rng = np.random.default_rng(77)
    V, E = 20_000, 400_000
    vid = np.arange(V, dtype=np.int64) * 5 + 3
    src = vid[rng.integers(0, V, E)]
    dst = vid[rng.integers(0, V, E)]
    src[:60_000] = vid[rng.integers(0, 4, 60_000)]        # four hubs with huge out-rows
    dst[60_000:120_000] = vid[rng.integers(4, 8, 60_000)]  # four with huge in-rows
    # 3000 vertices that are only ever destinations of edges leaving vertices nobody points to
    lonely_src = vid[V - 6000:V - 3000]
    keep = ~np.isin(dst, lonely_src)
    src, dst = src[keep], dst[keep]
    src = np.concatenate([src, np.repeat(lonely_src, 3)])
    dst = np.concatenate([dst, vid[rng.integers(V - 3000, V, 9000)]])
    gg.staging_clear()
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    csr = gg.build_csr()
    for kmin in (1, 2, 3):
        got = gg.expand_khop(csr, kmin, 3)
        gg.force_frontier(True)
        try:
            want = gg.expand_khop(csr, kmin, 3)
        finally:
            gg.force_frontier(False)
        assert got == want, kmin
    assert got["rows"][3] > 10**8
    csr.close()


def test_khop_mid_ranges_partition_the_result(gg, orc):
    """Middle-vertex sharding (multi-GPU entry point of the product kernel): shards add up."""
    vid, src, dst = datagen.ldbc_knows(6000, 250_000, 23)
    # make it properly directed: drop a third of the rows so in- and out-lists differ
    keep = (np.arange(src.size) % 3) != 0
    src, dst = src[keep], dst[keep]
    csr, g = build_both(gg, orc, vid, src, dst)
    whole = g.khop(1, 2)
    assert gg.expand_khop(csr, 1, 2) == whole
    for parts in (1, 2, 5, 8):
        b = gg.khop_partition_mid(csr, parts)
        assert b[0] == 0 and b[-1] == csr.V and all(x <= y for x, y in zip(b, b[1:]))
        rows = [0, 0, 0]
        dig = [0, 0, 0]
        te = fr = 0
        for lo, hi in zip(b, b[1:]):
            st = gg.expand_khop_mid(csr, lo, hi)
            for h in (1, 2):
                rows[h] += st["rows"][h]
                dig[h] = dsum(dig[h], st["digest"][h])
            te += st["traversed_edges"]
            fr += st["frontier_entries"]
        assert rows[1:] == whole["rows"][1:3] and dig[1:] == whole["digest"][1:3]
        assert te == whole["traversed_edges"] and fr == whole["frontier_entries"]
    csr.close()
    g.close()


def test_sharded_build_and_expand_add_up(gg, orc):
    """Multi-GPU path on one GPU: every 'rank' builds only the CSR rows of the vertices it owns
    (gg_csr_build_shard) and expands them; the shards' counts/digests add up to the whole query."""
    from duckdb_pgq_amd import GGError

    vid, src, dst = datagen.ldbc_knows(7000, 300_000, 29)
    keep = (np.arange(src.size) % 4) != 0  # directed: in- and out-lists differ
    src, dst = src[keep], dst[keep]
    src = np.concatenate([src, np.array([-99, vid[0]], np.int64)])  # two dangling rows
    dst = np.concatenate([dst, np.array([vid[1], -98], np.int64)])
    csr, g = build_both(gg, orc, vid, src, dst)
    whole = g.khop(1, 2)
    assert gg.expand_khop(csr, 1, 2) == whole
    csr.close()
    for parts in (2, 3, 8):
        rows = [0, 0, 0]
        dig = [0, 0, 0]
        te = fr = 0
        for part in range(parts):
            sh = gg.build_csr_shard(part, parts)
            st = gg.expand_khop(sh, 1, 2)
            for h in (1, 2):
                rows[h] += st["rows"][h]
                dig[h] = dsum(dig[h], st["digest"][h])
            te += st["traversed_edges"]
            fr += st["frontier_entries"]
            if part == 0:
                with pytest.raises(GGError):
                    gg.bfs64(sh, vid[:2], 3)
                with pytest.raises(GGError):
                    gg.expand_khop(sh, 1, 3)
                with pytest.raises(GGError):
                    gg.expand_khop(sh, 1, 2, materialise=True)
            sh.close()
        assert rows[1:] == whole["rows"][1:3] and dig[1:] == whole["digest"][1:3]
        assert te == whole["traversed_edges"] and fr == whole["frontier_entries"]
    g.close()


def test_each_shard_matches_oracle_sharded_the_same_way(gg, orc):
    """Per-shard results (not only their sum) equal the oracle restricted by the mirrored ownership
    function (duckdb_pgq_amd/sharding.py:owner_of == owns() in csrc/gg_csr.hip)."""
    from tests.test_multirank_cpu import shard_stats

    vid, src, dst = datagen.small_graph(120, 1500, 91, dangling=5, dup_edges=30)
    csr, g = build_both(gg, orc, vid, src, dst)
    csr.close()
    for parts in (2, 5):
        for part in range(parts):
            sh = gg.build_csr_shard(part, parts)
            got, ref = gg.expand_khop(sh, 1, 2), shard_stats(orc, g, vid, part, parts)
            assert got["rows"][1:3] == ref["rows"][1:3] and got["digest"][1:3] == ref["digest"][1:3]
            assert got["traversed_edges"] == ref["traversed_edges"]
            assert got["frontier_entries"] == ref["frontier_entries"]
            sh.close()
    g.close()


def test_bfs_deeper_than_254_levels(gg, orc):
    """A 700-vertex chain: distances exceed one byte, the BFS reruns with 16-bit cells."""
    vid = datagen.person_ids(700, 13)
    src, dst = vid[:-1], vid[1:]
    csr, g = build_both(gg, orc, vid, src, dst)
    sources = np.array([vid[0], vid[350], vid[699], vid[0]], np.int64)
    for max_hops in (-1, 300, 254, 255):
        dist, st = gg.bfs64(csr, sources, max_hops)
        o_dist, o_st = g.bfs64(g.lookup(sources), max_hops)
        assert np.array_equal(dist, o_dist) and st == o_st
    assert dist[0].max() == 255 and gg.bfs64(csr, sources, -1)[0][0].max() == 699
    csr.close()
    g.close()


def test_trainbenchmark_connectedsegments_path_part_on_gpu(gg, orc):
    """Train Benchmark SF1 (data the reference ships): the connectsTo part of ConnectedSegments — the
    6-vertex / 5-edge fixed-length path ct1..ct5 from every Segment — runs on the GPU (k = 5 materialised);
    the monitoredBy same-sensor joins are then applied with the oracle's hash join, and the result must be
    the reference's golden rows (benchmark/trainbenchmark/connectedsegments.benchmark:34-38)."""
    from tests import trainbenchmark as tb

    t = tb.tables()
    te = tb.load("TrackElement")[:, 0]
    ct, mb, seg = t["connectsTo"], t["monitoredBy"], t["Segment"][:, 0]
    gg.staging_clear()
    gg.append_vertices(te)
    gg.append_edges(ct[:, 0], ct[:, 1])
    csr = gg.build_csr()
    got = gg.expand_khop(csr, 5, 5, sources=seg, materialise=True)["tables"][5]
    # the same five joins through the oracle's JoinHashTable restatement
    rows = seg.reshape(-1, 1)
    for _ in range(5):
        m = orc.hash_join(ct[:, 0], rows[:, -1])
        rows = np.hstack([rows[m[:, 0]], ct[m[:, 1], 1:2]])
    assert np.array_equal(sort_rows(got), sort_rows(rows))
    # mb1..mb6 with equal sensors
    m = orc.hash_join(mb[:, 0], got[:, 0])
    out = np.hstack([mb[m[:, 1], 1:2], got[m[:, 0]]])
    for i in range(2, 7):
        m = orc.hash_join(mb[:, 0], out[:, i])
        keep = mb[m[:, 1], 1] == out[m[:, 0], 0]
        out = out[m[keep, 0]]
    assert np.array_equal(sort_rows(out), sort_rows(tb.CONNECTEDSEGMENTS_GOLDEN))
    csr.close()


def test_long_duplicate_chains(gg, orc):
    """The reference's duplicate-key stress (test/sql/join/inner/test_join_duplicates.test:14-24: a
    10 240-long chain for one key): 10 240 parallel edges a->b, a few b->a / b->c, plus a self loop."""
    vid = np.array([11, 22, 33, 44], np.int64)
    a, b, c, d = vid
    src = np.concatenate([np.full(10240, a), np.full(3, b), np.full(700, b), [d]])
    dst = np.concatenate([np.full(10240, b), np.full(3, a), np.full(700, c), [d]])
    csr, g = build_both(gg, orc, vid, src, dst)
    assert_csr_equal(csr, g)
    for k in [(1, 1), (1, 2), (2, 2), (1, 3)]:
        assert gg.expand_khop(csr, *k) == g.khop(*k)
    gg.force_frontier(True)
    try:
        assert gg.expand_khop(csr, 1, 2) == g.khop(1, 2)
    finally:
        gg.force_frontier(False)
    m = gg.expand_khop(csr, 2, 2, sources=np.array([a], np.int64), materialise=True)
    assert m["tables"][2].shape == (10240 * 703, 3)
    dist, st = gg.bfs64(csr, vid, -1)
    o_dist, o_st = g.bfs64(g.lookup(vid), -1)
    assert np.array_equal(dist, o_dist) and st == o_st
    csr.close()
    g.close()


def _connectedsegments_on_gpu(gg, te, sensors, seg, ct, mb):
    gg.staging_clear()
    gg.append_vertices(np.concatenate([te, sensors]))  # one id space: track elements and sensors
    gg.append_edges(ct[:, 0], ct[:, 1])
    path_csr = gg.build_csr()
    gg.staging_clear_edges()
    gg.append_edges(mb[:, 0], mb[:, 1])
    filter_csr = gg.build_csr()
    rows = gg.connected_paths_same_neighbour(path_csr, filter_csr, 5, sources=seg)
    path_csr.close()
    filter_csr.close()
    return rows


def test_trainbenchmark_connectedsegments_full_query_on_gpu(gg, orc):
    """BASELINE.json configs[4]: the whole ConnectedSegments query on the GPU (5-hop path over connectsTo
    from every Segment + same-sensor filter over monitoredBy).  SF1 = the reference's golden rows
    (benchmark/trainbenchmark/connectedsegments.benchmark:34-38); a 1024-fold id-shifted replication
    (SF1024-sized) = the shifted union of them, and equals the oracle's 11-join evaluation at SF16."""
    from tests import trainbenchmark as tb

    t = tb.tables()
    te = tb.load("TrackElement")[:, 0]
    sensors = tb.load("Sensor")[:, 0]
    seg, ct, mb = t["Segment"][:, 0], t["connectsTo"], t["monitoredBy"]
    got = _connectedsegments_on_gpu(gg, te, sensors, seg, ct, mb)
    assert np.array_equal(sort_rows(got), sort_rows(tb.CONNECTEDSEGMENTS_GOLDEN))

    base = {"te": te, "sensors": sensors, "seg": seg, "ct": ct, "mb": mb}
    for copies in (16, 1024):
        r = datagen.replicate_tables(base, copies)
        got = _connectedsegments_on_gpu(gg, r["te"], r["sensors"], r["seg"], r["ct"], r["mb"])
        shift = (np.arange(copies, dtype=np.int64) * r["_stride"])
        want = (tb.CONNECTEDSEGMENTS_GOLDEN[None, :, :] + shift[:, None, None]).reshape(-1, 7)
        assert np.array_equal(sort_rows(got), sort_rows(want))
        if copies == 16:
            ref = tb.connectedsegments_via_joins(orc, {"Segment": r["seg"].reshape(-1, 1), "connectsTo": r["ct"], "monitoredBy": r["mb"]})
            assert np.array_equal(sort_rows(got), sort_rows(ref))


def test_concurrent_sink_appends(gg, orc):
    """Sink is called concurrently from the pipeline's worker threads (physical_operator.hpp:137-139):
    8 threads append disjoint slices in 1024-row chunks with explicit rowids.  Row contents must not
    depend on the interleaving (compared per CSR row as sets of (rowid, neighbour))."""
    import threading

    vid, src, dst = datagen.ldbc_knows(4000, 150_000, 61)
    rowid = np.arange(src.size, dtype=np.int64)
    gg.staging_clear()
    gg.append_vertices(vid)
    nthreads = 8
    bounds = np.linspace(0, src.size, nthreads + 1).astype(int)
    errs = []

    def work(t):
        try:
            for o in range(bounds[t], bounds[t + 1], 1024):
                e = min(o + 1024, bounds[t + 1])
                gg.append_edges(src[o:e], dst[o:e], rowid[o:e])
        except Exception as ex:  # pragma: no cover
            errs.append(ex)

    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    assert gg.staging_counts() == (vid.size, src.size)
    csr = gg.build_csr()
    off, nbr, eid, _ = csr.export()
    rc, g = orc.csr_build(vid, src, dst, rowid)
    o_off, o_nbr, o_eid, _ = g.arrays()
    assert np.array_equal(off, o_off)
    # canonical order inside each row: by rowid
    row = np.repeat(np.arange(csr.V), np.diff(off))
    k = np.lexsort((eid, row))
    ko = np.lexsort((o_eid, row))
    assert np.array_equal(eid[k], o_eid[ko]) and np.array_equal(nbr[k], o_nbr[ko])
    assert gg.expand_khop(csr, 1, 2) == g.khop(1, 2)
    csr.close()
    g.close()


def test_build_without_edge_rowid(gg, orc):
    """gg_ctx_set_edge_rowid(0): same CSR (offsets, neighbours in rowid order), rowids reported as -1."""
    vid, src, dst = datagen.ldbc_knows(3000, 90_000, 71)
    gg.set_edge_rowid(False)
    try:
        csr, g = build_both(gg, orc, vid, src, dst)
        off, nbr, eid, v2 = csr.export()
        o_off, o_nbr, _, o_vid = g.arrays()
        assert np.array_equal(off, o_off) and np.array_equal(nbr, o_nbr) and np.array_equal(v2, o_vid)
        assert np.all(eid == -1)
        assert gg.expand_khop(csr, 1, 2) == g.khop(1, 2)
        dist, st = gg.bfs64(csr, vid[:64], 4)
        o_dist, o_st = g.bfs64(g.lookup(vid[:64]), 4)
        assert np.array_equal(dist, o_dist) and st == o_st
        csr.close()
        g.close()
    finally:
        gg.set_edge_rowid(True)


# (the last three: more ids than the first table holds — the second one takes them; more than both — the general path;
#  an LDBC shape, where the first table is sized to be just enough)
@pytest.mark.parametrize("V,E,seed", [(1, 3, 1), (7, 30, 2), (500, 9000, 3), (70000, 300000, 4), (200000, 150000, 5),
                                      (600000, 1000000, 6), (65645, 3877032, 7)])
def test_vertices_from_edges_is_the_sorted_distinct_endpoint_set(gg, orc, V, E, seed):
    """Join chains over an edge table alone: vertex table := distinct endpoints, ascending (gg.h)."""
    _, src, dst = datagen.small_graph(V, E, seed)
    gg.staging_clear()
    gg.append_edges(src, dst)
    n = gg.vertices_from_edges()
    expect = np.unique(np.concatenate([src, dst]))
    assert n == expect.size
    csr = gg.build_csr()
    rc, g = orc.csr_build(expect, src, dst, None)
    assert rc == 0
    assert_csr_equal(csr, g)  # vid column == np.unique, no edge dropped
    assert csr.dropped == 0 and csr.E == src.size
    k_max = 3 if E <= 30000 else 2
    assert gg.expand_khop(csr, 1, k_max) == g.khop(1, k_max)
    csr.close()
    g.close()


@pytest.mark.parametrize("legacy", [False, True])
def test_vertices_from_edges_with_clustered_ids_and_on_the_general_path(gg, orc, legacy):
    """Ids far from uniform — 100 000 consecutive ones and two outliers 2^62 apart — put nearly every id into one bucket
    of the interpolating bucket sort: the call must notice and take the general path (LSD rounds).  With the multi-pass
    build forced the general path is taken from the start.  Same vertex table either way."""
    rng = np.random.default_rng(5)
    ids = np.concatenate([np.arange(100_000, dtype=np.int64) * 3 + 17, np.array([-(1 << 62), 1 << 62], np.int64)])
    src = ids[rng.integers(0, ids.size, 400_000)]
    dst = ids[rng.integers(0, ids.size, 400_000)]
    src[:2], dst[:2] = ids[-2:], ids[-2:][::-1]
    gg.staging_clear()
    gg.force_legacy_build(legacy)
    gg.append_edges(src, dst)
    n = gg.vertices_from_edges()
    expect = np.unique(np.concatenate([src, dst]))
    assert n == expect.size
    csr = gg.build_csr()
    _, _, _, vid = csr.export()
    assert np.array_equal(vid, expect)
    rc, g = orc.csr_build(expect, src, dst, None)
    assert rc == 0 and gg.expand_khop(csr, 1, 2) == g.khop(1, 2)
    csr.close()
    g.close()


def test_vertices_from_edges_extreme_ids_and_empty(gg, orc):
    i64 = np.iinfo(np.int64)
    src = np.array([i64.min, -1, 0, i64.max, 5, i64.min, 1 << 32, -(1 << 32)], np.int64)
    dst = np.array([i64.max, 0, -1, i64.min, 5, 7, (1 << 32) + 1, -(1 << 32) - 1], np.int64)
    gg.staging_clear()
    gg.append_edges(src, dst)
    n = gg.vertices_from_edges()
    expect = np.unique(np.concatenate([src, dst]))
    assert n == expect.size
    csr = gg.build_csr()
    _, _, _, vid = csr.export()
    assert np.array_equal(vid, expect)
    csr.close()
    # only the sentinel id
    gg.staging_clear()
    gg.append_edges(np.array([i64.min], np.int64), np.array([i64.min], np.int64))
    assert gg.vertices_from_edges() == 1
    csr = gg.build_csr()
    assert csr.V == 1 and csr.E == 1
    csr.close()
    # no edges at all
    gg.staging_clear()
    assert gg.vertices_from_edges() == 0


@pytest.mark.parametrize("V,E,seed,n_src,max_hops", [(50, 200, 1, 7, 3), (3000, 40000, 4, 64, -1), (3000, 40000, 5, 64, 2),
                                                     (10, 0, 6, 3, 4)])
def test_bfs64_pairs_is_the_compacted_distance_matrix(gg, orc, V, E, seed, n_src, max_hops):
    vid, src, dst = datagen.small_graph(V, E, seed)
    csr, g = build_both(gg, orc, vid, src, dst)
    sources = datagen.pick_sources(vid, n_src, seed)
    sources = np.concatenate([sources, np.array([-77], np.int64)])[:64]  # one id that is not a vertex
    dist, st = gg.bfs64(csr, sources, max_hops)
    pairs, st2 = gg.bfs64_pairs(csr, sources, max_hops)
    assert st == st2 and pairs.shape[0] == st["reached_pairs"]
    lane, v = np.nonzero(dist >= 0)
    expect = np.stack([sources[lane], vid[v], dist[lane, v].astype(np.int64)], axis=1)
    assert np.array_equal(sort_rows(pairs), sort_rows(expect))
    csr.close()
    g.close()


@pytest.mark.parametrize("with_rowid", [False, True, "mixed"])
def test_concurrent_appends_across_many_staging_blocks(gg, orc, with_rowid):
    """16 threads push 3.3 M edge rows (more than three 1 M-row pinned blocks) in chunks of uneven size, so
    blocks fill, close, flush and reopen while other threads are still copying into them.  Every row must
    arrive exactly once (rows without rowid get their append position, which is then a permutation)."""
    import threading

    vid, src, dst = datagen.ldbc_knows(3000, 3_300_000, 77)
    rowid = np.arange(src.size, dtype=np.int64) + 10_000_000
    gg.staging_clear()
    gg.set_edge_rowid(True)
    gg.append_vertices(vid)
    nthreads = 16
    bounds = np.linspace(0, src.size, nthreads + 1).astype(int)
    errs = []

    def work(t):
        try:
            step = [1024, 777, 4096, 1][t % 4] if t % 4 != 3 else 50_000
            use_rowid = with_rowid is True or (with_rowid == "mixed" and t % 2 == 0)
            for o in range(bounds[t], bounds[t + 1], step):
                e = min(o + step, bounds[t + 1])
                gg.append_edges(src[o:e], dst[o:e], rowid[o:e] if use_rowid else None)
        except Exception as ex:  # pragma: no cover
            errs.append(ex)

    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    assert gg.staging_counts() == (vid.size, src.size)
    csr = gg.build_csr()
    off, nbr, eid, _ = csr.export()
    rc, g = orc.csr_build(vid, src, dst, rowid)
    o_off, o_nbr, o_eid, _ = g.arrays()
    assert csr.E == g.E and np.array_equal(off, o_off)
    row = np.repeat(np.arange(csr.V), np.diff(off))
    assert np.array_equal(np.sort(row * csr.V + nbr), np.sort(row * csr.V + o_nbr))  # same multiset of edges
    if with_rowid is True:
        k, ko = np.lexsort((eid, row)), np.lexsort((o_eid, row))
        assert np.array_equal(eid[k], o_eid[ko]) and np.array_equal(nbr[k], o_nbr[ko])
    elif with_rowid is False:
        assert np.array_equal(np.sort(eid), np.arange(csr.E))  # implicit rowids: a permutation of positions
    else:
        explicit = eid >= 10_000_000
        assert explicit.sum() == sum(bounds[t + 1] - bounds[t] for t in range(0, nthreads, 2))
        assert np.unique(eid).size == eid.size
    assert gg.expand_khop(csr, 1, 2) == g.khop(1, 2)
    csr.close()
    g.close()


def test_csr_lookup_and_pinned_host_buffers(gg, orc):
    vid, src, dst = datagen.small_graph(500, 4000, 9, dangling=6)
    csr, g = build_both(gg, orc, vid, src, dst)
    ids = np.concatenate([vid[::7], np.array([-3, 12345678901, np.iinfo(np.int64).min], np.int64), vid[:3]])
    got = gg.lookup(csr, ids)
    expect = np.asarray(g.lookup(ids), np.int64)
    assert np.array_equal(got.astype(np.int64), np.where(expect < 0, 0xFFFFFFFF, expect))
    assert gg.lookup(csr, np.zeros(0, np.int64)).size == 0
    # page-locked buffers: distinct while in use, reused after free
    a, b = gg.host_buffer(100_000), gg.host_buffer(100_000)
    a[:] = 1
    b[:] = 2
    assert a.sum() == 100_000 and b.sum() == 200_000
    pa = a.ctypes.data
    gg.lib.gg_host_free(gg.ctx, pa)
    c = gg.host_buffer(50_000)
    assert c.ctypes.data == pa  # the idle block is handed out again
    gg.lib.gg_host_free(gg.ctx, b.ctypes.data)
    gg.lib.gg_host_free(gg.ctx, c.ctypes.data)
    csr.close()
    g.close()


def test_vertices_from_edges_unions_with_staged_vertices(gg, orc):
    """Two edge tables over one id space (ConnectedSegments: connectsTo and monitoredBy): derive from the
    first, clear the edges, stage the second, derive again keeping what is there."""
    _, s1, d1 = datagen.small_graph(300, 2000, 21)
    _, s2, d2 = datagen.small_graph(500, 1500, 22)
    gg.staging_clear()
    gg.append_edges(s1, d1)
    n1 = gg.vertices_from_edges()
    assert n1 == np.unique(np.concatenate([s1, d1])).size
    gg.staging_clear_edges()
    gg.append_edges(s2, d2)
    n2 = gg.vertices_from_edges(keep_staged=True)
    expect = np.unique(np.concatenate([s1, d1, s2, d2]))
    assert n2 == expect.size
    csr = gg.build_csr()
    _, _, _, vid = csr.export()
    assert np.array_equal(vid, expect) and csr.E == s2.size
    csr.close()
    # keep with nothing staged before == plain derive; keep with no edges keeps the table as it is
    gg.staging_clear()
    gg.append_edges(s1, d1)
    assert gg.vertices_from_edges(keep_staged=True) == n1
    gg.staging_clear_edges()
    assert gg.vertices_from_edges(keep_staged=True) == n1


@pytest.mark.parametrize("name,ids", [
    ("dense_from_zero", np.arange(5000, dtype=np.int64)),
    ("dense_negative_offset", np.arange(5000, dtype=np.int64) * 3 - 7000),
    ("span_just_inside", np.concatenate([np.arange(4000, dtype=np.int64), [(1 << 20) - 1]])),
    ("span_just_outside", np.concatenate([np.arange(4000, dtype=np.int64), [1 << 20]])),
    ("around_int64_min", np.arange(3000, dtype=np.int64) + np.iinfo(np.int64).min),
    ("both_extremes", np.array([np.iinfo(np.int64).min, -1, 0, 1, np.iinfo(np.int64).max], np.int64)),
])
def test_dense_vertex_ids_take_the_direct_address_dictionary(gg, orc, name, ids):
    """Vertex ids spanning < 2^20 values are densified through an array indexed by id - min (the reference's
    perfect-hash-join case, plan_comparison_join.cpp:35-107); wider spans through the hash table.  Either
    way the CSR must be the oracle's, including edges whose endpoints fall outside [min, max]."""
    rng = np.random.default_rng(len(name))
    vid = ids[rng.permutation(ids.size)]
    E = 20 * vid.size
    src = vid[rng.integers(0, vid.size, E)]
    dst = vid[rng.integers(0, vid.size, E)]
    # dangling endpoints: below min, above max, and inside the span but not a vertex
    inside = np.setdiff1d(np.arange(int(vid.min()), int(vid.min()) + 50, dtype=np.int64), vid)[:5]
    extra = np.concatenate([inside, [np.iinfo(np.int64).max - 5, np.iinfo(np.int64).min + 5]]).astype(np.int64)
    extra = extra[~np.isin(extra, vid)]
    src = np.concatenate([src, extra, vid[: extra.size]])
    dst = np.concatenate([dst, vid[: extra.size], extra])
    csr, g = build_both(gg, orc, vid, src, dst)
    assert_csr_equal(csr, g)
    assert gg.expand_khop(csr, 1, 2) == g.khop(1, 2)
    csr.close()
    g.close()


@pytest.mark.parametrize("n_parts", [2, 3, 8])
def test_shard_built_from_its_local_edge_rows_equals_shard_built_from_the_whole_table(gg, orc, n_parts):
    """bench.py at N > 1 hash-partitions the edge table by endpoint owner (sharding.local_edge_rows): a rank
    stages only the rows with an endpoint it owns.  The shard CSR and its 2-hop result must not change."""
    from duckdb_pgq_amd import sharding

    vid, src, dst = datagen.ldbc_knows(3000, 120_000, 88)
    total = None
    for part in range(n_parts):
        per_input = []
        for s_, d_ in ((src, dst), sharding.local_edge_rows(src, dst, part, n_parts)):
            gg.staging_clear()
            gg.set_edge_rowid(False)
            gg.append_vertices(vid)
            gg.append_edges(s_, d_)
            c = gg.build_csr_shard(part, n_parts)
            per_input.append(gg.expand_khop(c, 1, 2))
            c.close()
        assert per_input[0] == per_input[1]
        vec = sharding.stats_to_vec(per_input[1])
        total = vec if total is None else [a + b for a, b in zip(total, vec)]
    gg.set_edge_rowid(True)
    local_sizes = [sharding.local_edge_rows(src, dst, p, n_parts)[0].size for p in range(n_parts)]
    assert max(local_sizes) < src.size * (2.0 / n_parts) * 1.1  # about 2/N - 1/N^2 of the table per rank
    rc, g = orc.csr_build(vid, src, dst)
    whole = g.khop(1, 2)
    assert total[0] == whole["rows"][1] and total[1] == whole["rows"][2] and total[4] == whole["traversed_edges"]
    g.close()


@pytest.mark.parametrize("keys", [[1, 2, 3], [-1, -2, 3]])
def test_perfect_hash_join_vectors_of_the_reference(gg, orc, keys):
    """test/sql/join/inner/test_join_perfect_hash.test:13-44 and :55-86: three build keys (positive; negative)
    probed by 15 rows, five per key -> 15 join rows.  As a graph: the build keys are the vertices (a dense
    id range, so the direct-address dictionary is taken), every probe row is an edge key -> key."""
    vid = np.array(keys, np.int64)
    probe = np.array(keys * 5, np.int64)
    csr, g = build_both(gg, orc, vid, probe, probe)
    assert_csr_equal(csr, g)
    st = gg.expand_khop(csr, 1, 1)
    assert st["rows"][1] == 15 and st == g.khop(1, 1)
    off, nbr, _, _ = csr.export()
    assert np.diff(off).tolist() == [5, 5, 5] and np.array_equal(vid[nbr], np.repeat(vid, 5))
    csr.close()
    g.close()


@pytest.mark.parametrize("n_parts", [1, 2, 3, 8])
@pytest.mark.parametrize("V,E,seed,max_hops", [(300, 2500, 31, -1), (3000, 40000, 32, 3), (40, 35, 33, 6)])
def test_graph_sharded_bfs_equals_the_whole_graph_bfs(gg, orc, n_parts, V, E, seed, max_hops):
    """the graph vertex-partitioned over N ranks, every rank pulling the next frontier
    words of its own vertices, words combined between levels (here on the host; over RCCL on N GPUs).  The
    union of the ranks' (source, vertex, distance) rows must be the unsharded BFS, i.e. the recursive CTE."""
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=3)
    sources = np.concatenate([datagen.pick_sources(vid, 40, seed), np.array([-9], np.int64)])
    gg.staging_clear()
    gg.set_edge_rowid(False)
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    whole = gg.build_csr()
    expect, st = gg.bfs64_pairs(whole, sources, max_hops)
    shards = [gg.build_csr_shard(p, n_parts) for p in range(n_parts)] if n_parts > 1 else [whole]
    got, levels = gg.bfs_sharded_emulated(shards, sources, max_hops)
    assert np.array_equal(sort_rows(got), sort_rows(expect))
    cte = orc.cte_shortest(vid, src, dst, sources[:-1], max_hops if max_hops >= 0 else V)
    assert np.array_equal(sort_rows(got[:, [0, 1, 2]]), sort_rows(cte))
    for c in shards:
        if c is not whole:
            c.close()
    whole.close()
    gg.set_edge_rowid(True)


@pytest.mark.parametrize("n_parts", [1, 3, 8])
@pytest.mark.parametrize("V,E,n_src,seed", [(3000, 40000, 3, 41), (20000, 400000, 64, 42), (500, 900, 2, 43)])
def test_graph_sharded_bfs_pushes_light_levels_and_pulls_heavy_ones(gg, orc, n_parts, V, E, n_src, seed):
    """Direction-optimising levels in the graph-sharded BFS: a rank pushes a light frontier along its edges grouped
    by source (into owned words only, so the exchange stays a SUM) and pulls a heavy one; ranks decide on their own.
    Rows equal the whole-graph BFS whichever mix of directions the ranks took."""
    vid, src, dst = datagen.small_graph(V, E, seed, dangling=2)
    sources = datagen.pick_sources(vid, n_src, seed)
    gg.staging_clear()
    gg.set_edge_rowid(False)
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    whole = gg.build_csr()
    expect, st = gg.bfs64_pairs(whole, sources, -1)
    shards = [gg.build_csr_shard(p, n_parts) for p in range(n_parts)] if n_parts > 1 else [whole]
    got, levels = gg.bfs_sharded_emulated(shards, sources, -1)
    assert np.array_equal(sort_rows(got), sort_rows(expect))
    per_rank = gg.last_sharded_levels
    assert all(push + pull == levels + 1 for push, pull in per_rank), (per_rank, levels)  # (+ the level that found nothing)
    assert sum(push for push, _ in per_rank) > 0, per_rank  # the seeds' level is light everywhere
    if E >= 40000:
        assert sum(pull for _, pull in per_rank) > 0, per_rank  # the middle levels cover most of the graph
    got2, _ = gg.bfs_sharded_emulated(shards, sources, -1)  # the push rows are cached on the shard
    assert np.array_equal(sort_rows(got2), sort_rows(expect))
    for c in shards:
        if c is not whole:
            c.close()
    whole.close()
    gg.set_edge_rowid(True)


def test_profile_select_times_only_the_named_kernels(gg, orc):
    vid, src, dst = datagen.small_graph(2000, 30000, 71)
    gg.staging_clear()
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    gg.profile_reset()
    gg.profile_select(["densify_pairs", "expand_mid2"])
    gg.profile(True)
    for _ in range(3):
        c = gg.build_csr()
        gg.expand_khop(c, 1, 2)
        c.close()
    gg.profile(False)
    prof = gg.profile_get()
    assert set(prof) == {"densify_pairs", "expand_mid2"} and prof["densify_pairs"][0] == 3 and prof["expand_mid2"][1] > 0
    gg.profile_reset()
    gg.profile_select(None)
    gg.profile(True)
    c = gg.build_csr()
    c.close()
    gg.profile(False)
    assert {"dict_insert", "densify_pairs", "col_scan", "partition_dual", "sub_sort", "leaf_rows"} <= set(gg.profile_get())
    gg.force_legacy_build(True)
    try:
        gg.profile_reset()
        gg.profile(True)
        c = gg.build_csr()
        c.close()
        gg.profile(False)
        assert {"ht_insert", "densify_hist", "radix_scatter", "scan_chained", "row_offsets"} <= set(gg.profile_get())
    finally:
        gg.force_legacy_build(False)


def test_bfs64_pairs_packed_words_decode_to_the_same_rows(gg, orc):
    vid, src, dst = datagen.small_graph(3000, 40000, 4)
    csr, g = build_both(gg, orc, vid, src, dst)
    sources = datagen.pick_sources(vid, 64, 4)
    rows, _ = gg.bfs64_pairs(csr, sources, -1)
    words = gg.bfs64_pairs_packed(csr, sources, -1)
    lane = (words >> np.uint64(58)).astype(np.int64)
    dist = ((words >> np.uint64(32)) & np.uint64(0x3FFFFFF)).astype(np.int64)
    dense = (words & np.uint64(0xFFFFFFFF)).astype(np.int64)
    decoded = np.stack([sources[lane], vid[dense], dist], axis=1)
    assert np.array_equal(sort_rows(decoded), sort_rows(rows))
    csr.close()
    g.close()


@pytest.mark.parametrize("order", ["src", "dst", "src_dst", "half"])
@pytest.mark.parametrize("V,E", [(65, 3000), (1024, 50_000), (70_000, 600_000), (300_000, 2_000_000)])
@pytest.mark.parametrize("rowid", [False, True])
def test_edge_tables_sorted_by_an_endpoint(gg, orc, order, V, E, rowid):
    """A table sorted by an endpoint (LDBC ships `knows` that way) puts runs of one key side by side: the build's
    counters then take one add per run (gg_runs.h) instead of one per lane.  Same arrays as the oracle, in
    both rank modes, whichever column the table is sorted by -- and when only half of it is."""
    rng = np.random.default_rng(V + E + len(order))
    vid = datagen.person_ids(V, 5)
    # skewed degrees: runs from one entry to thousands
    src = vid[np.minimum((rng.pareto(1.2, E) * V / 50).astype(np.int64), V - 1)]
    dst = vid[rng.integers(0, V, E)]
    if order == "src":
        o = np.argsort(src, kind="stable")
    elif order == "dst":
        o = np.argsort(dst, kind="stable")
    elif order == "src_dst":
        o = np.lexsort((dst, src))
    else:
        o = np.concatenate([np.argsort(src[: E // 2], kind="stable"), np.arange(E // 2, E)])
    src, dst = src[o], dst[o]
    gg.set_edge_rowid(rowid)
    try:
        rc, g = orc.csr_build(vid, src, dst)
        assert rc == 0
        o_off, o_nbr, o_eid, o_vid = g.arrays()
        want = g.khop(1, 2) if E <= 600_000 else None
        for mode in (1, 2):
            gg.rank_mode(mode)
            gg.staging_clear()
            gg.append_vertices(vid)
            gg.append_edges(src, dst)
            csr = gg.build_csr()
            off, nbr, eid, v2 = csr.export()
            assert np.array_equal(off, o_off), mode
            assert np.array_equal(nbr, o_nbr), mode
            assert np.array_equal(v2, o_vid), mode
            assert np.array_equal(eid, o_eid) if rowid else np.all(eid == -1), mode
            if want is not None:
                assert gg.expand_khop(csr, 1, 2) == want, mode  # (the reverse rows)
            csr.close()
            if want is not None and V == 70_000 and not rowid:  # shards of a sorted table (rows marked per direction) add up
                rows, dig, te = [0, 0, 0], [0, 0, 0], 0
                for part in range(3):
                    sh = gg.build_csr_shard(part, 3)
                    st = gg.expand_khop(sh, 1, 2)
                    for h in (1, 2):
                        rows[h] += st["rows"][h]
                        dig[h] = dsum(dig[h], st["digest"][h])
                    te += st["traversed_edges"]
                    sh.close()
                assert rows[1:] == want["rows"][1:3] and dig[1:] == want["digest"][1:3], mode
                assert te == want["traversed_edges"], mode
        g.close()
    finally:
        gg.rank_mode(0)
        gg.set_edge_rowid(True)  # (the session context's default)


def test_run_add_ranks_lane_by_lane(tmp_path):
    """gg::run_add (gg_runs.h) against a sequential count over 4 M entries, from random keys to runs of hundreds
    with invalid lanes and ragged tails; the program touches only its own LDS counters and output slots."""
    import json
    import pathlib
    import shutil
    import subprocess
    root = pathlib.Path(__file__).resolve().parents[1]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "ubench_runadd"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-I", str(root / "duckdb_pgq_amd" / "csrc"),
                    str(root / "scripts" / "ubench_runadd.hip"), "-o", str(exe)], check=True, timeout=300)
    out = subprocess.run([str(exe)], check=True, timeout=120, capture_output=True, text=True).stdout
    res = json.loads(out.strip().splitlines()[-1])
    assert res["bad_ranks"] == 0 and res["bad_counts"] == 0, res
    assert 0 < res["waves_with_runs"] < res["waves"], res


def _ids_for(kind, V, rng):
    if kind == "sparse":      # LDBC-like magnitude: packed 8-byte dictionary slots
        return datagen.person_ids(V, 11)
    if kind == "dense":       # span < 2^20: direct-address array
        return (np.arange(V, dtype=np.int64) * 2 - 77)[rng.permutation(V)]
    if kind == "wide":        # arbitrary 64-bit ids: 16-byte slots
        ids = np.unique(rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, V + V // 8, dtype=np.int64))
        return ids[rng.permutation(ids.size)][:V]
    raise AssertionError(kind)


@pytest.mark.parametrize("kind,V,E", [
    ("sparse", 1, 7), ("sparse", 33, 500), ("sparse", 64, 3000), ("sparse", 65, 3000), ("sparse", 1024, 50_000),
    ("sparse", 1025, 50_000), ("sparse", 70_000, 600_000), ("sparse", 300_000, 2_000_000),
    ("dense", 300_000, 1_000_000), ("wide", 40_000, 400_000),
    ("sparse32", 2_200_000, 3_000_000),   # 22 key bits: buckets of 4096 vertices, unpacked (low, payload) pairs
    ("wide", 1_100_000, 1_500_000),       # 21 key bits + 12 low bits > 32: unpacked pairs, 16-byte slots
])
@pytest.mark.parametrize("rowid", [False, True])
def test_bucketed_build_and_multipass_build_both_equal_the_oracle(gg, orc, kind, V, E, rowid):
    """gg_csr_build has two builds: the bucketed two-level one (gg_csr_fast.hip, <= 2^22 vertices) and the
    multi-pass LSD one (gg_csr.hip).  Both must export the oracle's arrays bit for bit, whichever id
    dictionary the device picks; the reverse CSR is checked through the 2-hop product kernel and the pull
    levels of the BFS."""
    rng = np.random.default_rng(V * 31 + E)
    if kind == "sparse32":  # small span, many vertices: packed slots, but low + key bits > 32
        vid = (np.arange(V, dtype=np.int64) * 3 + 7)[rng.permutation(V)]
    else:
        vid = _ids_for(kind, V, rng)
    src = vid[rng.integers(0, V, E)]
    dst = vid[rng.integers(0, V, E)]
    # a few hubs, self loops, duplicate rows and dangling endpoints
    src[: E // 10] = vid[rng.integers(0, min(V, 3), E // 10)]
    dst[E // 2: E // 2 + E // 20] = vid[0]
    bad = np.array([np.iinfo(np.int64).max - 3, int(vid.min()) - 1 if int(vid.min()) > np.iinfo(np.int64).min else 5], np.int64)
    bad = bad[~np.isin(bad, vid)]
    src = np.concatenate([src, bad, vid[: bad.size]])
    dst = np.concatenate([dst, vid[: bad.size], bad])
    if kind == "sparse" and V in (1024, 70_000):  # an edge table sorted by source: whole waves share a bucket
        order = np.argsort(src, kind="stable")
        src, dst = src[order], dst[order]
    gg.set_edge_rowid(rowid)
    try:
        rc, g = orc.csr_build(vid, src, dst)
        assert rc == 0
        o_off, o_nbr, o_eid, o_vid = g.arrays()
        want = g.khop(1, 2) if E <= 600_000 else None
        sources = vid[:: max(1, V // 64)][:64]
        o_dist, o_st = g.bfs64(g.lookup(sources), 4) if E <= 2_000_000 else (None, None)
        # the bucketed build ranks with ds_add_rtn (1) or match masks (2); "legacy" is the multi-pass build
        for legacy in (1, 2, "legacy"):
            gg.force_legacy_build(legacy == "legacy")
            gg.rank_mode(0 if legacy == "legacy" else legacy)
            gg.staging_clear()
            gg.append_vertices(vid)
            gg.append_edges(src, dst)
            csr = gg.build_csr()
            off, nbr, eid, v2 = csr.export()
            assert csr.V == g.V and csr.E == g.E and csr.dropped == g.dropped, legacy
            assert np.array_equal(off, o_off), legacy
            assert np.array_equal(nbr, o_nbr), legacy
            assert np.array_equal(v2, o_vid), legacy
            assert np.array_equal(eid, o_eid) if rowid else np.all(eid == -1), legacy
            if want is not None:
                assert gg.expand_khop(csr, 1, 2) == want, legacy
            if o_dist is not None:
                dist, st = gg.bfs64(csr, sources, 4)
                assert np.array_equal(dist, o_dist) and st == o_st, legacy
            csr.close()
        g.close()
    finally:
        gg.force_legacy_build(False)
        gg.rank_mode(0)
        gg.set_edge_rowid(True)


def test_a_scan_tile_that_never_publishes_is_an_error_not_a_wrong_csr(gg, orc):
    """The chained prefix scan (multi-pass build, frontier offsets, BFS row compaction) bounds its look-back spin.
    If a predecessor tile never publishes (injected here), the tiles behind it give up — and the call must fail
    with GG_ERR_HIP instead of returning offsets computed from a wrong prefix; kernels after the scan do nothing.
    Afterwards the context works again.  (The bucketed build has no chained scan: it is not affected.)"""
    from duckdb_pgq_amd import GGError

    vid, src, dst = datagen.ldbc_knows(20_000, 1_500_000, 5)
    rc, g = orc.csr_build(vid, src, dst)
    assert rc == 0
    try:
        gg.staging_clear()
        gg.append_vertices(vid)
        gg.append_edges(src, dst)
        gg.scan_fault(spin_limit=64, mute_tile=1)
        csr = gg.build_csr()  # bucketed build: column sums instead of a chained scan
        assert_csr_equal(csr, g)
        with pytest.raises(GGError) as e:  # k >= 3 from a source list: frontier offsets come from the chained scan
            gg.expand_khop(csr, 1, 3, sources=vid[:5000])
        assert e.value.code == -2 and "scan" in str(e.value)
        csr.close()
        gg.force_legacy_build(True)
        with pytest.raises(GGError) as e:
            gg.build_csr()
        assert e.value.code == -2 and "scan" in str(e.value)
        gg.scan_fault()
        csr = gg.build_csr()
        assert_csr_equal(csr, g)
        assert gg.expand_khop(csr, 1, 3, sources=vid[:5000]) == g.khop(1, 3, sources_dense=g.lookup(vid[:5000]).astype(np.uint32))
        csr.close()
    finally:
        gg.scan_fault()
        gg.force_legacy_build(False)
        g.close()


def _endpoint_sets(off, nbr, sources_dense, k_max):
    """CPU restatement: level h = set image of level h - 1 under the oracle's CSR (walk endpoints, no seen mask)."""
    V = off.size - 1
    level = np.zeros(V, bool)
    level[sources_dense] = True
    masks = np.zeros(V, np.int64)
    for h in range(1, k_max + 1):
        nxt = np.zeros(V, bool)
        for v in np.flatnonzero(level):
            nxt[nbr[off[v]:off[v + 1]]] = True
        masks |= nxt.astype(np.int64) << h
        level = nxt
    return masks


@pytest.mark.parametrize("k_max", [1, 2, 3])
def test_walk_endpoints_equal_the_union_of_the_hop_sets(gg, orc, k_max):
    """gg_walk_endpoints — the device form of `SELECT dst ... UNION SELECT e2.dst ...` (friends and friends of
    friends, interactive-complex-3.sql:3-12) — against set images computed from the oracle's CSR: one row per
    endpoint, in vertex order, bit h set iff a walk of exactly h edges ends there (sources on cycles included);
    unknown and duplicate sources, an empty source list, a vertex without edges."""
    vid, src, dst = datagen.small_graph(3000, 40_000, 61, dangling=8, dup_edges=50)
    csr, g = build_both(gg, orc, vid, src, dst)
    o_off, o_nbr, _, o_vid = g.arrays()
    for sources in (vid[[5]], vid[[5, 5, 77, 2999]], np.concatenate([vid[:40], np.array([-5, 1 << 60], np.int64)]),
                    np.zeros(0, np.int64), np.array([-7], np.int64)):
        dense = g.lookup(sources)
        dense = np.unique(dense[dense >= 0])
        want = _endpoint_sets(o_off, o_nbr, dense, k_max)
        ids, masks = gg.walk_endpoints(csr, sources, k_max)
        keep = np.flatnonzero(want)
        assert np.array_equal(ids, o_vid[keep]) and np.array_equal(masks, want[keep])
    csr.close()
    g.close()


def test_randomised_differential_run_against_the_oracle():
    """scripts/fuzz_gg.py for a fixed number of iterations (its own context on cuda:0): random graphs, id shapes, build
    modes, operations and knobs, everything compared with the oracle.  Longer runs: profiles/r04_fuzz.txt."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_gg.py"), "--seed", "7", "--iterations", "60",
                          "--max-rows", "400000", "--max-vertices", "100000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fuzz ok: 60 iterations" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
