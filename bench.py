#!/usr/bin/env python3
"""bench.py — traversed edges/sec on LDBC SNB 2-hop MATCH (Person-KNOWS*1..2-Person), MI355X.

One "step" = one pass of the hot path over the synthetic LDBC-shaped tables already resident in HBM:
    gg_csr_build (densify ids, histogram, scan, stable radix scatter)  +  gg_expand_khop_range(1..2)
i.e. what the reference does per query as hash-join build + probe chain.  With N > 1 ranks the
vertices are hash-partitioned (owner = hash(person id) mod N) and so is the edge table: a rank holds the
`knows` rows whose source or destination it owns (every row on at most two ranks; the 3.6 MB person
table is replicated), builds only the CSR rows of the vertices it owns (gg_csr_build_shard) and produces
the walks whose middle vertex it owns; there is no data-path collective, only one small all-reduce of
(rows, digest, TE) per step, so `value` = total traversed edges of the whole query / max-over-ranks
time ("strong" scaling: the query is fixed, ranks split it).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sf100|sf10|sf1] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (largest total time among densify / radix scatter / expansion; algorithmic bytes per SURVEY.md
§8d divided by the kernel's average duration measured with HIP events on the library's own stream) and
`cpu_baseline` (the compiled reference — oracle/_ref/libduckdb.so — or, if absent, the C oracle,
timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
MASK64 = (1 << 64) - 1


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(vid, src, dst, V, want_seconds=15.0, with_reference=True):
    """Timed CPU path on a bounded sample (sources = first S vertices in table order).

    kind "reference": the compiled reference runs the 1-hop and 2-hop join chains (count(*)) with all
    host threads; its counts are also checked against the GPU's counts for the same source range.
    kind "port": the C oracle's CSR formulation (OpenMP), when oracle/_ref is not present."""
    from oracle import ref_duckdb as R
    from tests import oracle_lib

    cores = os.cpu_count() or 1
    orc = oracle_lib.load()
    out = {}
    # -- C oracle over the FULL workload: parity check of the GPU result + "port" baseline
    t = time.perf_counter()
    rc, g = orc.csr_build(vid, src, dst)
    assert rc == 0
    t_build = time.perf_counter() - t
    t = time.perf_counter()
    ost = g.khop(1, 2)
    t_khop = time.perf_counter() - t
    port = {"value": ost["traversed_edges"] / (t_build + t_khop), "unit": "traversed edges/s",
            "cores": orc.num_threads(), "kind": "port",
            "sample": f"full workload: CSR build {t_build:.2f}s (1 thread) + 1..2-hop count/digest {t_khop:.2f}s (OpenMP)"}
    out["oracle_stats"] = ost
    oracle_off = g.arrays()[0]
    g.close()
    if not with_reference:  # N > 1: only the parity check of the combined result; the baseline is an N = 1 figure
        return out
    if not R.available():
        out["cpu_baseline"] = port
        return out
    # -- the compiled reference on a bounded sample: the first S rows of `knows` as first-hop edges
    #    (a filter on k1.rowid is pushed into the scan, so the sample really bounds the join work; a filter
    #    on the source persons does not — the optimizer still joins knows x knows first)
    db = R.RefDuckDB(threads=cores)
    t = time.perf_counter()
    db.load_ldbc(vid, src, dst)
    t_load = time.perf_counter() - t
    order = np.argsort(vid, kind="stable")
    svid = vid[order]

    def dense(ids):
        pos = np.searchsorted(svid, ids)
        pos[pos >= V] = 0
        ok = svid[pos] == ids
        return np.where(ok, order[pos], -1)

    deg = np.diff(oracle_off)
    E = src.size

    def sample_counts(S):
        u, v = dense(src[:S]), dense(dst[:S])
        ok = (u >= 0) & (v >= 0)
        return int(ok.sum()), int(deg[v[ok]].sum())

    def run(S):
        c1, t1 = db.timed(R.sql_khop(1, where_extra=f"k1.rowid < {S}"))
        c2, t2 = db.timed(R.sql_khop(2, where_extra=f"k1.rowid < {S}"))
        return int(c1[0, 0]), int(c2[0, 0]), t1 + t2

    # two calibration points -> fixed cost (hash-table builds over all of knows) + slope
    S1, S2 = max(1, E // 512), max(2, E // 128)
    _, _, ta = run(S1)
    _, _, tb = run(S2)
    slope = max((tb - ta) / (S2 - S1), 1e-9)
    fixed = max(ta - slope * S1, 0.0)
    S = int(min(E, max(S2, (want_seconds - fixed) / slope))) if want_seconds > fixed else S2
    c1, c2, tt = run(S)
    r1, r2 = sample_counts(S)
    db.close()
    out["cpu_baseline"] = {
        "value": (r1 + r2) / tt, "unit": "traversed edges/s", "cores": cores, "kind": "reference",
        "sample": (f"reference DuckDB (oracle/_ref/libduckdb.so, PRAGMA threads={cores}) count(*) of the 1-hop and 2-hop "
                   f"join chains restricted to the first {S} of {E} knows rows as first-hop edges: TE={r1 + r2} in {tt:.2f}s "
                   f"(its hash-table builds over all knows rows included, ~{fixed:.1f}s; table load {t_load:.1f}s excluded)"),
        "counts_match_oracle": bool(c1 == r1 and c2 == r2),
    }
    out["cpu_port"] = port
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sf100", choices=["sf0.1", "sf1", "sf10", "sf100"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--shard-of", type=int, default=0,
                    help="diagnostic: time rank 0's share of an N-rank run on one GPU (output is not a bench line)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus={args.gpus}; using WORLD_SIZE")
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the gg hot path has no CPU fallback")
    # one process per GPU; GG_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 path
    backend = os.environ.get("GG_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    import duckdb_pgq_amd as pkg
    from duckdb_pgq_amd import sharding

    # ---- synthetic LDBC-shaped tables (identical on every rank), staged to HBM before timing -------
    t0 = time.perf_counter()
    vid, src, dst = pkg.datagen.ldbc(args.workload)
    t_gen = time.perf_counter() - t0
    V, R = vid.size, src.size
    # N > 1: the edge table is hash-partitioned across the GPUs by endpoint ownership — a rank holds the
    # rows whose source or destination vertex it owns (each row on at most two ranks), not a replica of
    # the whole table; the vertex table (3.6 MB) is replicated.  Placement happens here, before timing,
    # like the staging itself.
    parts = args.shard_of if args.shard_of > 1 else world
    src_all, dst_all = src, dst  # the CPU leg (rank 0) checks the combined result against the whole graph
    src, dst = sharding.local_edge_rows(src, dst, 0 if args.shard_of > 1 else rank, parts)
    R_local = src.size
    gg = pkg.GG(device_index)
    # the benchmarked MATCH binds no edge variable: like the reference's build side, carry only the key columns
    gg.set_edge_rowid(False)
    t0 = time.perf_counter()
    gg.chunk_rows = 122_880  # one DuckDB row group per append (storage/table/row_group.hpp:38-39)
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    gg.staging_sync()
    t_stage = time.perf_counter() - t0
    if rank == 0:
        log(f"{args.workload}: V={V} knows rows={R} ({R_local} on this rank); datagen {t_gen:.1f}s, "
            f"staging (PCIe) {t_stage*1e3:.1f} ms")

    def step():
        # all persons are sources; with N ranks this rank builds only the CSR rows of the vertices it
        # owns (owner = hash(id) mod N) and produces the walks whose middle vertex it owns
        if args.shard_of > 1:
            c = gg.build_csr_shard(0, args.shard_of)
        else:
            c = gg.build_csr_shard(rank, world) if world > 1 else gg.build_csr()
        st = gg.expand_khop(c, 1, 2)
        c.close()
        vec = sharding.combine(sharding.stats_to_vec(st), dist, device="cuda")
        return vec, st

    for _ in range(args.warmup):
        step()
    # HIP events around a launch are not free (two records per launch: ~0.3 ms of a 3.9 ms step when all
    # ~35 launches of a step are timed), so the timed region times only the kernels the roofline can name;
    # the per-kernel table of everything else comes from a few extra, untimed steps afterwards.
    ROOFLINE_KERNELS = ["densify_hist", "densify_shard", "radix_scatter", "expand_mid2", "expand_fused2"]
    gg.profile_reset()
    gg.profile_select(ROOFLINE_KERNELS)
    gg.profile(True)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tot, st_local = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gg.profile(False)
    prof = gg.profile_get()
    # untimed: every kernel, for the breakdown (kernels[...]["timed_region"] tells the two apart)
    TABLE_STEPS = 3
    gg.profile_reset()
    gg.profile_select(None)
    gg.profile(True)
    for _ in range(TABLE_STEPS):
        step()
    gg.profile(False)
    prof_all = gg.profile_get()
    barrier()
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    rows1, rows2, dig1, dig2, te_total, fr_total = tot
    ms_per_step = elapsed / args.steps * 1e3
    value = te_total * args.steps / elapsed

    # ---- roofline of the dominant kernel on this rank ---------------------------------------------------
    # dominant kernel = largest total time in the timed region; algorithmic bytes per launch:
    # SURVEY.md §8d's figures — expansion 8*TE + 16*frontier entries; densification 32E + 8V; CSR scatter
    # 16E read + 8E write = 24E per CSR, which our LSD sort spreads over 3 passes (x2 CSRs = 6 launches),
    # so one radix_scatter launch is charged 8E: the multi-pass overhead shows up as a low fraction.
    te_l, fr_l = st_local["traversed_edges"], st_local["frontier_entries"]
    alg = {
        "expand_mid2": 8 * te_l + 16 * fr_l,
        "expand_fused2": 8 * te_l + 16 * fr_l,
        "densify_hist": 32 * R_local + 8 * V,
        "densify_shard": 32 * R_local + 8 * V,
        "radix_scatter": 8 * R_local,
    }
    dom = max((k for k in prof if k in alg), key=lambda k: prof[k][1], default=None)
    launches, total_ms = prof.get(dom, (0, 0.0)) if dom else (0, 0.0)
    traffic_tab = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic_tab = json.load(open(tpath))
        except Exception:
            traffic_tab = {}
    roof = None
    if launches:
        avg_s = total_ms / launches * 1e-3
        achieved = alg[dom] / avg_s
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic_tab.get(f"{args.workload}/{dom}/n{world}"),
                "avg_launch_ms": avg_s * 1e3, "launches_per_step": launches / args.steps,
                "algorithmic_bytes_per_launch": alg[dom]}
    # the same figure per phase of the step (all kernels of the phase together)
    # phase totals: the timed region's figure where a kernel was timed there, the untimed pass's otherwise
    per_step_ms = {k: v[1] / TABLE_STEPS for k, v in prof_all.items()}
    per_step_ms.update({k: v[1] / args.steps for k, v in prof.items()})
    expand_names = {"expand_mid2", "expand_fused2", "reduce_partials", "tile_partition"}
    t_expand = sum(ms for k, ms in per_step_ms.items() if k in expand_names) * 1e-3
    t_build = sum(ms for k, ms in per_step_ms.items() if k not in expand_names) * 1e-3
    alg_build = (32 * R_local + 8 * V) + (32 * R_local + 16 * V)  # densification + one CSR without rowid
    phases = {}
    if t_build > 0:
        phases["csr_build"] = {"kernel_ms": t_build * 1e3, "algorithmic_bytes": alg_build,
                               "achieved": alg_build / t_build / 1e9, "frac": alg_build / t_build / HBM_PEAK,
                               "note": "our build also makes the reverse CSR the product kernel needs; it is not in the algorithmic count"}
    if t_expand > 0:
        a = 8 * te_l + 16 * fr_l
        phases["expand"] = {"kernel_ms": t_expand * 1e3, "algorithmic_bytes": a, "achieved": a / t_expand / 1e9,
                            "frac": a / t_expand / HBM_PEAK,
                            "traffic": traffic_tab.get(f"{args.workload}/expand_mid2/n{world}"),
                            "note": "above 1: the product kernel reads each CSR row once and is VALU-bound (DESIGN.md §4.3)"}
    # the whole step against the same ceiling: SURVEY.md §8d's algorithmic bytes of build + expansion over the
    # measured step time (north_star's target is stated on this workload: >= 50 % of the HBM roofline)
    a_step = alg_build + 8 * te_l + 16 * fr_l
    phases["whole_step"] = {"step_ms": ms_per_step, "algorithmic_bytes": a_step,
                            "achieved": a_step / (ms_per_step * 1e-3) / 1e9,
                            "frac": a_step / (ms_per_step * 1e-3) / HBM_PEAK,
                            "note": "dominated by the expansion's 8 bytes per traversed edge; see roofline_phases.expand"}
    kernels = {k: {"launches": v[0], "avg_us": (v[1] / v[0] * 1e3 if v[0] else 0.0),
                   "us_per_step": v[1] * 1e3 / TABLE_STEPS, "timed_region": False} for k, v in prof_all.items()}
    kernels.update({k: {"launches": v[0], "avg_us": (v[1] / v[0] * 1e3 if v[0] else 0.0),
                        "us_per_step": v[1] * 1e3 / args.steps, "timed_region": True} for k, v in prof.items()})

    if args.shard_of > 1:
        log(f"shard 0 of {args.shard_of}: {ms_per_step:.3f} ms/step; kernels us/step:",
            {k: round(v["us_per_step"]) for k, v in kernels.items()})
        gg.close()
        return
    if rank == 0:
        extra = {}
        if not args.no_cpu:
            extra = cpu_baseline(vid, src_all, dst_all, V, args.cpu_seconds, with_reference=(world == 1))
            ost = extra.pop("oracle_stats")
            parity = (ost["rows"][1] == rows1 and ost["rows"][2] == rows2 and ost["digest"][1] == dig1
                      and ost["digest"][2] == dig2 and ost["traversed_edges"] == te_total)
            extra["parity_vs_oracle"] = bool(parity)
            if not parity:
                log("PARITY FAILURE", ost, tot)
        line = {
            "metric": "traversed edges/sec on LDBC SNB 2-hop MATCH",
            "value": value,
            "unit": "traversed edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int64 ids / u32 dense indices",
            "data": "synthetic (seeded LDBC-shaped person/knows, duckdb_pgq_amd/datagen.py)",
            "config": {"workload": f"LDBC SNB {args.workload.upper()} Person-KNOWS*1..2-Person, all persons as sources: CSR build (no edge-rowid payload) + 2-hop expansion (count + digest)",
                       "vertices": int(V), "knows_rows": int(R), "rows_1hop": int(rows1), "rows_2hop": int(rows2),
                       "traversed_edges": int(te_total), "parallelism": f"vertex-ownership shards x{world} (edge table hash-partitioned by endpoint owner, vertex table replicated, CSR + expansion sharded, no data-path collective)"},
            "roofline": roof,
            "roofline_phases": phases,
            "kernels": kernels,
            "staging_ms_pcie": t_stage * 1e3,
        }
        line.update(extra)
        if "cpu_baseline" not in line:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    gg.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
