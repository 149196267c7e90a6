#!/usr/bin/env python3
"""bench_cs.py — secondary benchmark: Train Benchmark ConnectedSegments at SF1024 (BASELINE.json configs[4]).

The reference's query (benchmark/trainbenchmark/queries/connectedsegments.sql: 11 hash joins) as the GPU path
runs it: CSR of connectsTo + CSR of monitoredBy over one id space (track elements and sensors), 5-hop walks
from the Segment ids, kept where all six segments share a sensor.  A step = both CSR builds + the query,
result rows left in HBM and counted.  Inputs: the SF1 tables the reference ships (tests/golden/
trainbenchmark_sf1/), replicated 1024 times with shifted ids — the result must be the 1024 shifted copies of
the reference's four golden rows (connectedsegments.benchmark:34-38), which rank 0 checks.

With N ranks (python -m torch.distributed.run --nproc-per-node N bench_cs.py) the seed Segments are split
into N slices and both (20 MB) CSRs are replicated — SURVEY.md §8e (iii): no exchange, one all-reduce of
the row counts.  Not the driver's bench line (that is bench.py); results go to profiles/.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import duckdb_pgq_amd as pkg
    from duckdb_pgq_amd import datagen
    from tests import trainbenchmark as tb

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    device = 0
    if world > 1:
        import torch
        import torch.distributed as dist

        rehearsal = os.environ.get("GG_BENCH_BACKEND", "nccl") != "nccl"
        device = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(device)
        dist.init_process_group(os.environ.get("GG_BENCH_BACKEND", "nccl"))

    t = tb.tables()
    base = {"te": tb.load("TrackElement")[:, 0], "sensors": tb.load("Sensor")[:, 0], "seg": t["Segment"][:, 0],
            "ct": t["connectsTo"], "mb": t["monitoredBy"]}
    r = datagen.replicate_tables(base, args.copies)
    seeds = np.array_split(r["seg"], world)[rank]  # this rank's slice of the seed Segments
    gg = pkg.GG(device)
    gg.set_edge_rowid(False)
    vertices = np.concatenate([r["te"], r["sensors"]])

    def step(fetch=False):
        gg.staging_clear()
        gg.append_vertices(vertices)
        gg.append_edges(r["ct"][:, 0], r["ct"][:, 1])
        path_csr = gg.build_csr()
        gg.staging_clear_edges()
        gg.append_edges(r["mb"][:, 0], r["mb"][:, 1])
        filter_csr = gg.build_csr()
        rows = gg.connected_paths_same_neighbour(path_csr, filter_csr, 5, sources=seeds)
        path_csr.close()
        filter_csr.close()
        return rows

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = step()
    dt = time.perf_counter() - t0
    n_rows = rows.shape[0]
    if dist is not None:
        import torch

        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tot = torch.tensor([n_rows], dtype=torch.int64, device=dev)
        slow = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(slow, op=dist.ReduceOp.MAX)
        n_rows, dt = int(tot[0]), float(slow[0])
    # parity on this rank's slice: the shifted golden rows whose first segment is one of its seeds
    shift = np.arange(args.copies, dtype=np.int64) * r["_stride"]
    want = (tb.CONNECTEDSEGMENTS_GOLDEN[None, :, :] + shift[:, None, None]).reshape(-1, 7)
    want = want[np.isin(want[:, 1], seeds)]
    key = lambda a: a[np.lexsort(a.T[::-1])]
    ok = bool(rows.shape == want.shape and np.array_equal(key(rows), key(want)))
    if dist is not None:
        flag = torch.tensor([int(ok)], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(int(flag[0]))
    if rank == 0:
        print(json.dumps({
            "metric": "Train Benchmark ConnectedSegments, whole query per step (2 CSR builds + 5-hop walks + same-sensor filter)",
            "workload": f"SF{args.copies} (SF1 of the reference x{args.copies}, shifted ids)", "n_gpus": world,
            "scaling": "strong", "steps": args.steps, "ms_per_step": dt / args.steps * 1e3,
            "queries_per_s": args.steps / dt, "connectsTo_rows": int(r["ct"].shape[0]),
            "monitoredBy_rows": int(r["mb"].shape[0]), "segments": int(r["seg"].size), "result_rows": n_rows,
            "golden_rows_reproduced": ok,
            "note": "staging of the 1.3 M edge rows (host -> HBM) is inside the step"}))
    gg.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
