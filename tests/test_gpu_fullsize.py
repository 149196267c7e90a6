"""Full-size GPU checks at BASELINE.json's configurations (seeded synthetic LDBC-shaped tables)."""
import numpy as np
import pytest

from duckdb_pgq_amd import datagen, sharding

pytestmark = pytest.mark.gpu


def stage(gg, vid, src, dst):
    gg.staging_clear()
    gg.chunk_rows = 122_880
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    gg.chunk_rows = 0


def test_sf10_two_hop_and_bfs_against_oracle(gg, orc):
    """configs[1]: SF10 Person-KNOWS*1..2-Person — counts, digests, TE bit-exact vs the oracle (1.07 G walks), in count
    mode and materialised; configs[2]-shaped BFS: 64 sources to fixpoint, every distance equal."""
    vid, src, dst = datagen.ldbc("sf10")
    stage(gg, vid, src, dst)
    csr = gg.build_csr()
    rc, g = orc.csr_build(vid, src, dst)
    assert rc == 0
    got = gg.expand_khop(csr, 1, 2)
    assert got == g.khop(1, 2)
    assert got["rows"][2] > 1_000_000_000
    # the same walks materialised (25.5 GB of id columns in HBM, 32 k tiles of k_mat_mid2): every row's hash summed
    # on the device equals the oracle's digest; three middle-vertex ranges with odd bounds partition rows and digest
    res = gg.expand_khop_result(csr, 2)
    assert res.digest(csr, 2) == (got["rows"][2], got["digest"][2])
    res.close()
    V = int(vid.size)
    n_sum = d_sum = 0
    for lo, hi in ((0, 777), (777, V // 2 + 1), (V // 2 + 1, V)):
        part = gg.expand_khop_mid_result(csr, lo, hi, k_min=2)
        n, d = part.digest(csr, 2)
        assert n == part.rows(2)
        n_sum, d_sum = n_sum + n, (d_sum + d) & 0xFFFFFFFF
        part.close()
    assert (n_sum, d_sum) == (got["rows"][2], got["digest"][2])
    sources = datagen.pick_sources(vid, 64, 7)
    dist, st = gg.bfs64(csr, sources, -1)
    o_dist, o_st = g.bfs64(g.lookup(sources), -1)
    assert np.array_equal(dist, o_dist) and st == o_st
    csr.close()
    g.close()


def test_sf100_against_the_oracle_and_size_independent_properties(gg, orc):
    """configs[2] / configs[3] size (SF100, 12.8 G walks): the 2-hop counts, digests and traversed edges and one
    64-source BFS batch to fixpoint against the C oracle's CSR formulation on the same tables (a few seconds of
    host time), then properties that need no oracle run: product kernel == frontier kernels; ownership shards
    add up to the whole; rebuilding gives the same bits."""
    vid, src, dst = datagen.ldbc("sf100")
    stage(gg, vid, src, dst)
    csr = gg.build_csr()
    whole = gg.expand_khop(csr, 1, 2)
    assert whole["rows"][1] == csr.E and whole["traversed_edges"] == whole["rows"][1] + whole["rows"][2]
    rc, g = orc.csr_build(vid, src, dst)
    assert rc == 0
    assert whole == g.khop(1, 2)
    assert whole["rows"][2] > 12 * 10**9
    batch = datagen.pick_sources(vid, 64, 5)
    dist, st = gg.bfs64(csr, batch, -1)
    o_dist, o_st = g.bfs64(g.lookup(batch), -1)
    assert np.array_equal(dist, o_dist) and st == o_st
    del dist, o_dist
    g.close()
    gg.force_frontier(True)
    try:
        assert gg.expand_khop(csr, 1, 2) == whole
    finally:
        gg.force_frontier(False)
    # the mirrored table makes the graph symmetric: 2-hop walks = sum of squared degrees
    off = csr.export()[0]
    deg = np.diff(off)
    assert int((deg.astype(np.int64) ** 2).sum()) == whole["rows"][2]
    csr.close()
    rows = [0, 0, 0]
    dig = [0, 0, 0]
    for part in range(4):
        sh = gg.build_csr_shard(part, 4)
        st = gg.expand_khop(sh, 1, 2)
        for h in (1, 2):
            rows[h] += st["rows"][h]
            dig[h] = sharding.dsum(dig[h], st["digest"][h])
        sh.close()
    assert rows[1:] == whole["rows"][1:3] and dig[1:] == whole["digest"][1:3]
    csr2 = gg.build_csr()
    assert gg.expand_khop(csr2, 1, 2) == whole
    # BFS from 64 sources: an undirected connected component is reached symmetrically
    sources = datagen.pick_sources(vid, 64, 3)
    dist, st = gg.bfs64(csr2, sources, -1, targets=sources)
    assert np.array_equal(dist, dist.T) and st["levels"] >= 3
    csr2.close()


def test_three_hop_product_kernel_against_the_oracle_at_a_million_edges(gg, orc):
    """k_expand_mid3 against the oracle's streaming depth-first count (orc_khop_csr enumerates the join chain's
    walks without storing them) where the frontier kernels are not the yardstick: 1.18 M edge rows, 45 G walks,
    counts, digests and traversed edges of every level for k_min 1..3 — also with the launch split into grids of a
    few thousand workgroups (gg_debug_max_grid_tiles: what a result beyond 2^32 threads per grid takes)."""
    vid, src, dst = datagen.ldbc_knows(40_000, 1_200_000, 0x3A0)
    stage(gg, vid, src, dst)
    csr = gg.build_csr()
    rc, g = orc.csr_build(vid, src, dst)
    assert rc == 0 and csr.E > 1_000_000
    for kmin in (1, 2, 3):
        want = g.khop(kmin, 3)
        assert gg.expand_khop(csr, kmin, 3) == want, kmin
        gg.max_grid_tiles(4099)
        try:
            assert gg.expand_khop(csr, kmin, 3) == want, kmin
        finally:
            gg.max_grid_tiles(0)
    assert want["rows"][3] > 4 * 10**10
    csr.close()
    g.close()


def test_sf10_three_hop_product_kernel_equals_frontier_kernels(gg):
    """288 G three-hop walks at SF10: the product kernel ({2-hop rows ending in b} x out(b)) and the frontier kernels
    (one hash per walk) return the same counts, digests and traversed-edge totals for every k_min; the distinct
    endpoints of 1..2-hop walks from 64 sources (gg_walk_endpoints) are the vertices a bounded BFS reaches, the
    sources that lie on a 2-cycle included."""
    vid, src, dst = datagen.ldbc("sf10")
    stage(gg, vid, src, dst)
    csr = gg.build_csr()
    for kmin in (1, 3):
        got = gg.expand_khop(csr, kmin, 3)
        gg.force_frontier(True)
        try:
            want = gg.expand_khop(csr, kmin, 3)
        finally:
            gg.force_frontier(False)
        assert got == want, kmin
    assert got["rows"][3] > 2 * 10**11
    sources = datagen.pick_sources(vid, 64, 11)
    ids, masks = gg.walk_endpoints(csr, sources, 2)
    dist, _ = gg.bfs64(csr, sources, 2)
    reached = vid[np.flatnonzero(((dist == 1) | (dist == 2)).any(axis=0))]
    # walk endpoints = BFS vertices at distance 1..2, plus sources that end a 2-walk (distance 0 in the BFS)
    extra = np.setdiff1d(ids, reached)
    assert np.isin(reached, ids).all() and np.isin(extra, sources).all()
    assert ((masks & ~0b110) == 0).all() and (masks != 0).all()
    csr.close()
