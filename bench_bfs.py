#!/usr/bin/env python3
"""bench_bfs.py — secondary benchmark: LDBC SNB shortest_path(Person, Person), 64-source bitset BFS
(BASELINE.json configs[2]).  Not the driver's bench line (that is bench.py); results go to profiles/.

A step = one 64-source batch run to fixpoint on a prebuilt CSR (distances stay in HBM).  With N ranks
(python -m torch.distributed.run --nproc-per-node N bench_bfs.py), every rank holds the whole CSR and takes
different source batches (source-batch sharding, SURVEY.md §8e: no data-path communication, weak scaling);
one all-reduce adds up the traversed edges and takes the slowest rank's time.
CPU baseline: the C oracle's bitset BFS on one batch; and, where oracle/_ref exists, the reference's
recursive CTE (bi-10 friends/friends_shortest) on a bounded hop count.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def graph_sharded(args, pkg, sharding, gg, device, dist, rank, world, vid, src, dst):
    """one BFS spread over the ranks.  Every rank builds the shard of the vertices it owns
    from its part of the hash-partitioned edge table, holds the whole frontier, pulls the next frontier words
    of its vertices and adds its words to the others' with one all-reduce per level (disjoint supports: SUM is
    OR; 8*V bytes = 3.6 MB at SF100).  Result rows stay sharded."""
    import torch

    gg.set_edge_rowid(False)
    gg.append_vertices(vid)
    s_loc, d_loc = sharding.local_edge_rows(src, dst, rank, world)
    gg.append_edges(s_loc, d_loc)
    shard = gg.build_csr_shard(rank, world) if world > 1 else gg.build_csr()
    on_device = dist is not None and dist.get_backend() == "nccl"

    def run_batch(sources, fetch=False):
        run = gg.bfs_sharded_begin(shard, sources)
        levels = pairs = 0
        while args.max_hops < 0 or levels < args.max_hops:
            new = run.expand()
            if dist is not None:
                if on_device:  # RCCL straight on the words in HBM
                    words = torch.as_tensor(run, device="cuda")
                    count = torch.tensor([new], dtype=torch.int64, device="cuda")
                    dist.all_reduce(words, op=dist.ReduceOp.SUM)
                    dist.all_reduce(count, op=dist.ReduceOp.SUM)
                    torch.cuda.synchronize()
                    new = int(count[0])
                else:  # rehearsal backend: through the host
                    words = torch.from_numpy(run.words().view(np.int64))
                    count = torch.tensor([new], dtype=torch.int64)
                    dist.all_reduce(words, op=dist.ReduceOp.SUM)
                    dist.all_reduce(count, op=dist.ReduceOp.SUM)
                    run.set_words(words.numpy().view(np.uint64))
                    new = int(count[0])
            if new == 0:
                break
            pairs += new
            run.commit()
            levels += 1
        rows = run.pairs() if fetch else None  # (the rows stay on the device unless somebody wants them)
        run.close()
        return levels, pairs, rows

    batches = [pkg.datagen.pick_sources(vid, 64, 0x5EED, batch=b) for b in range(args.batches)]
    run_batch(batches[0])  # warm-up
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    lv = pairs = 0
    for b in batches:
        l, p, _ = run_batch(b)
        lv += l
        pairs += p
    dt = time.perf_counter() - t0
    _, _, rows = run_batch(batches[-1], fetch=True)
    # parity: the union of the ranks' rows of the last batch against the whole-graph BFS (rank 0 builds it):
    # same number of rows and the same sum of all their fields (mod 2^59)
    ok = None
    if not args.no_cpu:
        M = 1 << 59
        mine = torch.tensor([rows.shape[0], int(rows.astype(np.uint64).sum(dtype=np.uint64) % M)], dtype=torch.int64,
                            device="cuda" if on_device else "cpu")
        if dist is not None:
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        if rank == 0:
            g2 = pkg.GG(device)
            g2.set_edge_rowid(False)
            g2.append_vertices(vid)
            g2.append_edges(src, dst)
            whole = g2.build_csr()
            expect, _ = g2.bfs64_pairs(whole, batches[-1], args.max_hops)
            ok = bool(expect.shape[0] == int(mine[0]) and
                      int(expect.astype(np.uint64).sum(dtype=np.uint64) % M) == int(mine[1]) % M)
            whole.close()
            g2.close()
    if rank == 0:
        print(json.dumps({"metric": "64-source bitset BFS, graph-sharded (one BFS over all ranks)", "workload": args.workload,
                          "n_gpus": world, "batches": args.batches, "ms_per_batch": dt / args.batches * 1e3,
                          "levels_per_batch": lv / args.batches, "reached_pairs_per_batch": pairs / args.batches,
                          "exchange": "all-reduce(SUM) of 8*V bytes per level" if dist is not None else "none",
                          "collective_backend": dist.get_backend() if dist is not None else None,
                          "rows_match_whole_graph_bfs": ok}))
    shard.close()
    gg.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sf100")
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--max-hops", type=int, default=-1)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--graph-sharded", action="store_true",
                    help="N ranks share ONE BFS per batch: the graph is vertex-partitioned (gg_csr_build_shard), every "
                         "rank pulls the next frontier words of its vertices, one all-reduce of 8*V bytes per level")
    args = ap.parse_args()
    import duckdb_pgq_amd as pkg
    from duckdb_pgq_amd import sharding

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or "WORLD_SIZE" in os.environ:  # under torch.distributed.run, also with ONE rank (RCCL on one GPU)
        import torch
        import torch.distributed as dist

        # GG_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path
        rehearsal = os.environ.get("GG_BENCH_BACKEND", "nccl") != "nccl"
        local_dev = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_dev)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(os.environ["GG_BENCH_BACKEND"])
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))

    vid, src, dst = pkg.datagen.ldbc(args.workload)
    backend = os.environ.get("GG_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = (local if backend == "nccl" else 0) if dist is not None else 0
    gg = pkg.GG(device)
    if args.graph_sharded:
        graph_sharded(args, pkg, sharding, gg, device, dist, rank, world, vid, src, dst)
        return
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    csr = gg.build_csr()
    batches = [pkg.datagen.pick_sources(vid, 64, 0x5EED, batch=b)
               for b in sharding.source_batches(rank, world, args.batches)]
    gg.bfs64(csr, batches[0], args.max_hops, fetch=False)  # warm-up
    t0 = time.perf_counter()  # wall time without event records (a level is four launches)
    te = act = lv = 0
    for b in batches:
        _, st = gg.bfs64(csr, b, args.max_hops, fetch=False)
        te += st["traversed_edges"]
        act += st["active_vertices"]
        lv += st["levels"]
    dt = time.perf_counter() - t0
    gg.profile_reset()  # kernel times from a second, profiled pass
    gg.profile(True)
    for b in batches:
        gg.bfs64(csr, b, args.max_hops, fetch=False)
    if dist is not None:  # whole-job numbers: edges add up, the slowest rank sets the time
        import torch

        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        sums = torch.tensor([te, act, lv], dtype=torch.int64, device=dev)
        slowest = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        te_job, dt = int(sums[0]), float(slowest[0])
    else:
        te_job = te
    gg.profile(False)
    prof = gg.profile_get()
    V = csr.V
    alg = 8 * V * lv + 16 * act + 24 * te + 24 * V * lv  # SURVEY.md §8d per-level formula summed over levels
    kern_ms = sum(v[1] for k, v in prof.items() if k.startswith("bfs_"))
    line = {
        "metric": "traversed edges/sec, 64-source bitset BFS (shortest path)", "workload": args.workload,
        "n_gpus": world, "scaling": "weak", "batches_per_gpu": args.batches, "max_hops": args.max_hops,
        "collective_backend": dist.get_backend() if dist is not None else None,
        "value": te_job / dt, "unit": "traversed edges/s",
        "ms_per_batch": dt / args.batches * 1e3, "levels_per_batch": lv / args.batches,
        "roofline": {"bound": "hbm", "achieved": alg / (kern_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (kern_ms * 1e-3) / 8e12, "kernel_ms_per_batch": kern_ms / args.batches},
        "kernels": {k: {"launches": v[0], "us_per_batch": v[1] * 1e3 / args.batches} for k, v in prof.items()},
    }
    if not args.no_cpu and rank == 0:
        from tests import oracle_lib

        orc = oracle_lib.load()
        rc, g = orc.csr_build(vid, src, dst)
        t0 = time.perf_counter()
        d, ost = g.bfs64(g.lookup(batches[0]), args.max_hops)
        cdt = time.perf_counter() - t0
        _, gst = gg.bfs64(csr, batches[0], args.max_hops, fetch=False)
        dmat, _ = gg.bfs64(csr, batches[0], args.max_hops)  # (not `dist`: that name is the process group module)
        line["parity_vs_oracle"] = bool(np.array_equal(d, dmat) and ost == gst)
        line["cpu_port"] = {"value": ost["traversed_edges"] / cdt, "unit": "traversed edges/s", "cores": 1,
                            "sample": f"one 64-source batch, C oracle bitset BFS, {cdt:.2f}s"}
        g.close()
        if world == 1:  # the reference's own statement on a bounded sample (bench.py: SF1, 64 seeds, hopCount < 5)
            import bench
            ref = bench._reference_cte_baseline(pkg)
            if ref:
                line["cpu_reference"] = ref
    if rank == 0:
        print(json.dumps(line))
    csr.close()
    gg.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
