"""world_size-2 (and 3) `gloo` tests of the N>1 path's host logic on CPU: vertex-ownership sharding +
the single all-reduce that combines per-rank results.  The per-rank compute is the ORACLE restricted to
the walks whose middle vertex the rank owns — the same contract gg_csr_build_shard + gg_expand_khop
fulfil on a GPU (tests/test_gpu_parity.py::test_sharded_build_and_expand_add_up)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from duckdb_pgq_amd import datagen, sharding
from tests import oracle_lib


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard_stats(orc, g, vid, part, n_parts):
    """1..2-hop stats of the walks whose middle vertex (1-hop rows: destination) is owned by `part`."""
    off, nbr, _, _ = g.arrays()
    V = vid.size
    own = sharding.owner_of(vid, n_parts) == part
    rows = [0, 0, 0]
    dig = [0, 0, 0]
    for u in range(V):
        for i in range(off[u], off[u + 1]):
            x = int(nbr[i])
            if not own[x]:
                continue
            rows[1] += 1
            dig[1] = sharding.dsum(dig[1], orc.row_hash([u, x]))
            for j in range(off[x], off[x + 1]):
                rows[2] += 1
                dig[2] = sharding.dsum(dig[2], orc.row_hash([u, x, int(nbr[j])]))
    return {"rows": rows, "digest": dig, "traversed_edges": rows[1] + rows[2],
            "frontier_entries": int(own.sum()) + rows[1]}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.load()
        vid, src, dst = datagen.small_graph(90, 700, 77, dangling=4, dup_edges=9)
        rc, g = orc.csr_build(vid, src, dst)
        assert rc == 0
        local = shard_stats(orc, g, vid, rank, world)
        total = sharding.combine(sharding.stats_to_vec(local), dist)
        whole = g.khop(1, 2)
        assert total == sharding.stats_to_vec(whole), (rank, total, whole)
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_query_combines_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_owner_partition_and_halves():
    vid = datagen.person_ids(10_000, 5)
    for n in (1, 2, 8):
        o = sharding.owner_of(vid, n)
        assert o.min() >= 0 and o.max() < n
        if n > 1:
            counts = np.bincount(o, minlength=n)
            assert counts.min() > 0.8 * vid.size / n  # hash ownership is balanced
    vec = [2**40 + 5, 3, (2**32 - 1), 2**31 + 7, 2**35, 9]
    assert sharding.join_halves(sharding.split_halves(vec)) == vec
    # digest fields are 32-bit sums: they wrap mod 2^32 and never carry into the high half
    a, b = 0xFFFFFFFF, 0x00000001
    parts = [x + y for x, y in zip(sharding.split_halves([0, 0, a, a, 0, 0]), sharding.split_halves([0, 0, b, b, 0, 0]))]
    assert sharding.join_halves(parts)[2] == 0


def _bfs_worker(rank, world, port, q):
    """Source-batch sharding of the 64-lane BFS (bench_bfs.py with N ranks): every rank runs its own
    batches on the whole graph; one all-reduce adds the per-rank statistics.  Per-rank compute = the
    oracle's bitset BFS."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.load()
        vid, src, dst = datagen.small_graph(400, 3000, 91)
        rc, g = orc.csr_build(vid, src, dst)
        assert rc == 0
        per_rank = 2
        stat = lambda b: g.bfs64(g.lookup(datagen.pick_sources(vid, 64, 0x5EED, batch=b)), -1)[1]
        mine = sharding.source_batches(rank, world, per_rank)
        te = sum(stat(b)["traversed_edges"] for b in mine)
        pairs = sum(stat(b)["reached_pairs"] for b in mine)
        t = torch.tensor([te, pairs], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        every = range(world * per_rank)
        assert int(t[0]) == sum(stat(b)["traversed_edges"] for b in every)
        assert int(t[1]) == sum(stat(b)["reached_pairs"] for b in every)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_bfs_source_batches_shard_over_gloo():
    world = 2
    all_batches = sorted(b for r in range(world) for b in sharding.source_batches(r, world, 3))
    assert all_batches == list(range(6))  # disjoint, and together a prefix of the batch sequence
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bfs_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_local_edge_rows_cover_every_shard_need():
    """Hash-partitioning the edge table by endpoint owner: every row lands on the owner of its source and
    on the owner of its destination, nowhere else; about 2/N - 1/N^2 of the table per rank."""
    vid, src, dst = datagen.ldbc_knows(2000, 60_000, 5)
    for n in (1, 2, 8):
        sizes = []
        for part in range(n):
            s, d = sharding.local_edge_rows(src, dst, part, n)
            sizes.append(s.size)
            os_, od = sharding.owner_of(s, n), sharding.owner_of(d, n)
            assert ((os_ == part) | (od == part)).all()
            # nothing this rank needs was left out
            assert s.size == int(((sharding.owner_of(src, n) == part) | (sharding.owner_of(dst, n) == part)).sum())
        if n == 1:
            assert sizes == [src.size]
        else:
            expect = src.size * (2.0 / n - 1.0 / n**2)  # on average; hubs make single ranks deviate
            assert abs(sum(sizes) / n - expect) < 0.1 * expect and max(sizes) < 1.6 * expect


def _graph_sharded_worker(rank, world, port, q):
    """Graph-sharded BFS (bench_bfs.py --graph-sharded, gg_bfs_sharded_*): every rank holds the frontier words
    of all vertices and the in-neighbour lists of the vertices it owns; a level = pull the owned vertices'
    next words, all-reduce(SUM) the word arrays (disjoint supports: the sum is the OR), commit.  Per-rank
    compute here is numpy over the oracle's CSR; the union of the ranks' rows must be the oracle's BFS."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.load()
        vid, src, dst = datagen.small_graph(300, 2200, 57, dangling=3)
        rc, g = orc.csr_build(vid, src, dst)
        assert rc == 0
        off, nbr, _, _ = g.arrays()
        V = vid.size
        # in-neighbour lists of the owned vertices only (what gg_csr_build_shard keeps on this rank)
        own = sharding.owner_of(vid, world) == rank
        rows = np.repeat(np.arange(V), np.diff(off))
        keep = own[nbr]
        order = np.argsort(nbr[keep], kind="stable")
        in_dst, in_src = nbr[keep][order], rows[keep][order]
        roff = np.searchsorted(in_dst, np.arange(V + 1))
        sources = datagen.pick_sources(vid, 20, 9)
        sd = g.lookup(sources)
        front = np.zeros(V, np.uint64)
        seen = np.zeros(V, np.uint64)
        dist_local = {}
        for lane, v in enumerate(sd):
            front[v] |= np.uint64(1) << np.uint64(lane)
            if own[v]:
                seen[v] |= np.uint64(1) << np.uint64(lane)
                dist_local[(lane, int(v))] = 0
        level = 0
        while True:
            level += 1
            nxt = np.zeros(V, np.uint64)
            new = 0
            for w in np.nonzero(own)[0]:
                acc = np.bitwise_or.reduce(front[in_src[roff[w]:roff[w + 1]]]) if roff[w + 1] > roff[w] else np.uint64(0)
                nw = acc & ~seen[w]
                if nw:
                    nxt[w] = nw
                    seen[w] |= nw
                    for lane in range(len(sd)):
                        if (int(nw) >> lane) & 1:
                            dist_local[(lane, int(w))] = level
                            new += 1
            words = torch.from_numpy(nxt.view(np.int64))
            count = torch.tensor([new], dtype=torch.int64)
            dist.all_reduce(words, op=dist.ReduceOp.SUM)  # disjoint supports: SUM == OR
            dist.all_reduce(count, op=dist.ReduceOp.SUM)
            if int(count[0]) == 0:
                break
            front = words.numpy().view(np.uint64).copy()
        # union over ranks == the oracle's whole-graph BFS
        d, _ = g.bfs64(sd, -1)
        mine = sorted((lane, v, h) for (lane, v), h in dist_local.items())
        expect = sorted((int(l), int(v), int(d[l, v])) for l, v in zip(*np.nonzero(d >= 0)) if own[v])
        assert mine == expect, (rank, len(mine), len(expect))
        total = torch.tensor([len(mine)], dtype=torch.int64)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        assert int(total[0]) == int((d >= 0).sum())
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_graph_sharded_bfs_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
