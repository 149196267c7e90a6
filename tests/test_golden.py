"""Golden vectors produced by the compiled reference (tests/golden/make_golden.py) pin both the C
oracle (CPU tests) and the HIP path (GPU tests)."""
import os

import numpy as np
import pytest

from tests.oracle_lib import sort_rows

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["ldbc_tiny", "ldbc_small", "ldbc_sf0_1"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def hops_of(g, prefix):
    return sorted(int(k[len(prefix):]) for k in g if k.startswith(prefix) and k[len(prefix):].isdigit())


def bfs_cases(g):
    i = 0
    while f"bfs{i}_rel" in g:
        yield g[f"bfs{i}_sources"], int(g[f"bfs{i}_max_hops"][0]), g[f"bfs{i}_rel"]
        i += 1


def dist_to_relation(sources, vid, dist):
    rel = [(int(sources[i]), int(vid[v]), int(dist[i, v])) for i in range(len(sources)) for v in np.nonzero(dist[i] >= 0)[0]]
    return sort_rows(np.array(rel, np.int64).reshape(-1, 3))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_golden(orc, name):
    g = load(name)
    vid, src, dst = g["vid"], g["src"], g["dst"]
    rc, c = orc.csr_build(vid, src, dst)
    assert rc == 0
    kmax = max(hops_of(g, "count"))
    st = c.khop(1, kmax)
    for h in hops_of(g, "count"):
        assert st["rows"][h] == int(g[f"count{h}"][0])
    hr = hops_of(g, "rows")
    if hr:
        j = orc.khop_join(vid, src, dst, 1, max(hr))   # join-chain restatement
        r = c.khop_rows(1, max(hr))                     # CSR formulation
        for h in hr:
            assert np.array_equal(sort_rows(vid[j[h]]), g[f"rows{h}"])
            assert np.array_equal(sort_rows(r[h]), g[f"rows{h}"])
            assert st["digest"][h] == orc.digest_rows(j[h])
    for sources, max_hops, rel in bfs_cases(g):
        if name != "ldbc_sf0_1":
            assert np.array_equal(sort_rows(orc.cte_shortest(vid, src, dst, sources, max_hops)), rel)
        dist, _ = c.bfs64(c.lookup(sources), max_hops)
        assert np.array_equal(dist_to_relation(sources, vid, dist), rel)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_reference_golden(gg, name):
    g = load(name)
    vid, src, dst = g["vid"], g["src"], g["dst"]
    gg.staging_clear()
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    csr = gg.build_csr()
    kmax = max(hops_of(g, "count"))
    hr = hops_of(g, "rows")
    st = gg.expand_khop(csr, 1, kmax)
    for h in hops_of(g, "count"):
        assert st["rows"][h] == int(g[f"count{h}"][0])
    if hr:
        m = gg.expand_khop(csr, 1, max(hr), materialise=True)
        for h in hr:
            assert np.array_equal(sort_rows(m["tables"][h]), g[f"rows{h}"])
    for sources, max_hops, rel in bfs_cases(g):
        dist, _ = gg.bfs64(csr, sources, max_hops)
        assert np.array_equal(dist_to_relation(sources, vid, dist), rel)
    csr.close()
