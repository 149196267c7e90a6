#!/usr/bin/env python3
"""bench.py — traversed edges/sec on LDBC SNB 2-hop MATCH (Person-KNOWS*1..2-Person), MI355X.

One "step" = one pass of the hot path over the synthetic LDBC-shaped tables already resident in HBM:
    gg_csr_build (densify ids, bucket partition, sub-bucket sort, rows: forward + reverse CSR)  +
    gg_expand_khop_range(1..2)
i.e. what the reference does per query as hash-join build + probe chain.  With N > 1 ranks the
vertices are hash-partitioned (owner = hash(person id) mod N) and so is the edge table: a rank holds the
`knows` rows whose source or destination it owns (every row on at most two ranks; the 3.6 MB person
table is replicated), builds only the CSR rows of the vertices it owns (gg_csr_build_shard) and produces
the walks whose middle vertex it owns; there is no data-path collective, only one small all-reduce of
(rows, digest, TE) per step, so `value` = total traversed edges of the whole query / max-over-ranks
time ("strong" scaling: the query is fixed, ranks split it).  The K timed steps are K builds, K expansions and K
all-reduces inside the timed region; with N > 1 the host launches step i + 1's build before it combines step i's
counts, so the collective runs beside the build's queued kernels (run_steps).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sf100|sf10|sf1] [--no-cpu] [--no-extras]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement):
  roofline           the kernel with the largest total time in the timed region.  HBM-bound kernels (the CSR
                     build): algorithmic bytes per launch / average launch time / 8 TB/s.  The 2-hop
                     product kernel moves 0.6 GB for 12.8 G walks and is bound by vector-ALU issue, so its record is
                     {"bound": "valu"}: one v_xad_u32 per walk against the measured issue rate of that instruction
                     (scripts/ubench_valu.hip).  No record ever divides bytes that are not moved by the HBM peak.
  roofline_kernels   the same record for every kernel of the step that is charged algorithmic work
  roofline_phases    csr_build (all build kernels, HBM) and expand (VALU)
  cpu_baseline       the compiled reference (oracle/_ref/libduckdb.so; else the C oracle) on this box's host cores,
                     benchmark_runner protocol (1 cold + 5 hot runs, median hot: benchmark/benchmark_runner.cpp:132-147)
                     on a bounded sample, plus a threads=1 figure and the CPU model
  materialised, bfs64, connectedsegments   the other BASELINE.json configs on this GPU, each with its own parity
                     boolean (N = 1 only; --no-extras skips them)
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
# v_xad_u32 (xor + add, one per walk in k_expand_mid2): measured issue rate on this part, the counted loop holding
# nothing but the instruction (scripts/ubench_valu.hip -> profiles/r03_ubench_valu.txt): 4.38 cycles per wave-instruction
# and SIMD with 8 waves per SIMD (8.9 with one wave), the same with the state in a scalar register (4.28) — every
# three-source VOP3 runs at that rate (v_add3_u32 4.24), two-source VOP2 instructions at 2.3-2.4 (v_add_u32, v_xor_b32,
# v_fma_f32: the guide's 2-cycle row), so xor + add as two instructions costs 2 x 2.48 = 4.96: the fused form is the
# cheapest way to take one walk.  Peak = 256 CUs x 4 SIMDs x 64 lanes x 2.4e9 / 4.38 = 35.9 T lane-ops/s.
VALU_XAD_CYCLES = 4.38
VALU_XAD_PEAK = 256 * 4 * 64 * 2.4e9 / VALU_XAD_CYCLES
MASK64 = (1 << 64) - 1
PROFILE_TAG = "r02"


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def _probe_rate_record(V, probes, seconds):
    """The densification beside the roofline that actually bounds it: two random 16-byte dictionary probes per edge
    row, against the rate a kernel of the same shape (two id columns streamed in, pairs streamed out) reaches with a
    table of the dictionary's size — scripts/ubench_gather_sizes.hip, output committed under profiles/."""
    pairs = max(512, ((V * 100 + 99) // 100 + 63) // 64 * 64)  # gg_csr_fast.hip: load factor 50 %, 16 bytes per pair
    table_mb = pairs * 16 / 1048576.0
    rec = {"probes_per_launch": int(probes), "achieved": probes / seconds / 1e9, "unit": "G probes/s",
           "dictionary_MB": table_mb, "ceiling": None, "frac": None,
           "note": "ceiling = measured rate of random 16-byte probes beside the same streams for a table of this size "
                   "(profiles/r03_ubench_gather_sizes.txt, interpolated); the HBM fraction above charges SURVEY 8d's "
                   "bytes, which this kernel does not move at HBM's pace because every probe is its own request"}
    path = os.path.join(ROOT, "profiles", "r03_ubench_gather_sizes.txt")
    try:
        pts = {}
        for line in open(path):
            f = line.split()
            if len(f) >= 7 and f[0] == "table" and f[2] == "MB":
                pts[float(f[1])] = max(pts.get(float(f[1]), 0.0), float(f[5]))
        xs = sorted(pts)
        lo = max([x for x in xs if x <= table_mb], default=xs[0])
        hi = min([x for x in xs if x >= table_mb], default=xs[-1])
        c = pts[lo] if hi == lo else pts[lo] + (pts[hi] - pts[lo]) * (table_mb - lo) / (hi - lo)
        rec["ceiling"] = c
        rec["frac"] = rec["achieved"] / c
    except Exception:
        pass
    return rec


def _pmc_traffic(key):
    """HBM-side bytes from the committed counter passes (profiles/pmc_traffic.json), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
    except Exception:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(vid, src, dst, V, want_seconds=6.0, with_reference=True):
    """Timed CPU path on a bounded sample (first-hop edges = the first S rows of knows).

    kind "reference": the compiled reference runs the 1-hop and 2-hop join chains (count(*)) with all host
    threads, 1 cold + 5 hot runs, median of the hot runs (benchmark/benchmark_runner.cpp:132-147); its counts are
    checked against the oracle's for the same rows.  Also one threads=1 figure on a smaller sample.
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
            "cores": orc.num_threads(), "kind": "port", "cpu_model": cpu_model(),
            "sample": f"full workload: CSR build {t_build:.2f}s (<=32 threads) + 1..2-hop count/digest {t_khop:.2f}s (OpenMP)"}
    out["oracle_stats"] = ost
    out["oracle_graph"] = g
    oracle_off = g.arrays()[0]
    if not with_reference:  # N > 1: only the parity check of the combined result; the baseline is an N = 1 figure
        return out
    if not R.available():
        out["cpu_baseline"] = port
        return out
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

    db = R.RefDuckDB(threads=cores)
    t0 = time.perf_counter()
    db.load_ldbc(vid, src, dst)
    t_load = time.perf_counter() - t0

    def leg(threads, seconds, hot_runs=5, calibrate=True):
        """(sample rows S, counts, [cold, hot...] seconds, fixed seconds) at `threads`."""
        db.execute(f"PRAGMA threads={threads}")

        def run(S):  # a filter on k1.rowid is pushed into the scan, so the sample really bounds the join work
            c1, t1 = db.timed(R.sql_khop(1, where_extra=f"k1.rowid < {S}"))
            c2, t2 = db.timed(R.sql_khop(2, where_extra=f"k1.rowid < {S}"))
            return int(c1[0, 0]), int(c2[0, 0]), t1 + t2

        # two calibration points -> fixed cost (hash-table builds over all of knows) + slope
        S1, S2 = max(1, E // 512), max(2, E // 128)
        cal = None
        if calibrate:
            _, _, ta = run(S1)
            _, _, tb = run(S2)
            cal = (S1, ta, S2, tb)
            slope = max((tb - ta) / (S2 - S1), 1e-9)
            fixed = max(ta - slope * S1, 0.0)
            S = int(min(E, max(S2, (seconds - fixed) / slope))) if seconds > fixed else S2
        else:  # one thread: the hash-table builds over all of knows alone take seconds per run; keep the sample small
            S, fixed = S1, float("nan")
        times, counts = [], None
        for _ in range(1 + hot_runs):  # the first run at this sample size is the cold one
            c1, c2, tt = run(S)
            times.append(tt)
            counts = (c1, c2)
        return S, counts, times, fixed, t_load, cal

    S, (c1, c2), times, fixed, t_load, cal = leg(cores, want_seconds)
    r1, r2 = sample_counts(S)
    hot = statistics.median(times[1:])
    S1t, (d1, d2), times1, fixed1, _, _ = leg(1, want_seconds, hot_runs=2, calibrate=False)
    # marginal rate: traversed edges per second of PROBING, from the two calibration samples (their fixed part — the
    # hash-table builds over all of knows — cancels in the difference), and what the full workload would take at it
    (Sa, ta, Sb, tb) = cal
    te_a, te_b = sum(sample_counts(Sa)), sum(sample_counts(Sb))
    marginal = (te_b - te_a) / max(tb - ta, 1e-9)
    te_full = int(out["oracle_stats"]["traversed_edges"])
    full_estimate = fixed + te_full / max(marginal, 1.0)
    db.close()
    q1, q2 = sample_counts(S1t)
    hot1 = statistics.median(times1[1:])
    out["cpu_baseline"] = {
        "value": (r1 + r2) / hot, "unit": "traversed edges/s", "cores": cores, "kind": "reference",
        "cpu_model": cpu_model(),
        "protocol": "1 cold + 5 hot runs, median of the hot runs (benchmark/benchmark_runner.cpp:132-147)",
        "cold_s": times[0], "hot_s": times[1:], "median_hot_s": hot,
        "sample": (f"reference DuckDB (oracle/_ref/libduckdb.so, PRAGMA threads={cores}) count(*) of the 1-hop and 2-hop "
                   f"join chains restricted to the first {S} of {E} knows rows as first-hop edges: TE={r1 + r2} per run "
                   f"(its hash-table builds over all knows rows included, ~{fixed:.1f}s; table load {t_load:.1f}s excluded)"),
        "counts_match_oracle": bool(c1 == r1 and c2 == r2),
        "marginal_value": marginal,
        "marginal_note": (f"slope between two samples ({Sa} and {Sb} first-hop rows: TE {te_a} in {ta:.2f}s, {te_b} in {tb:.2f}s): "
                          "traversed edges per second of probing once the hash tables are built"),
        "full_workload_estimate_s": full_estimate,
        "full_workload_note": f"fixed {fixed:.1f}s (hash-table builds) + {te_full} traversed edges at the marginal rate; not run",
        "threads_1": {"value": (q1 + q2) / hot1, "unit": "traversed edges/s", "cores": 1, "median_hot_s": hot1,
                      "cold_s": times1[0],
                      "hot_s": times1[1:],
                      "sample": f"same statements, PRAGMA threads=1, first {S1t} knows rows as first-hop edges: TE={q1 + q2} per "
                                f"run, 1 cold + 2 hot runs (the hash-table builds over all knows rows dominate a run)",
                      "counts_match_oracle": bool(d1 == q1 and d2 == q2)},
    }
    out["cpu_port"] = port
    return out


# ---- the other BASELINE.json configs (N = 1, after the timed region) --------------------------------------------
def extra_bfs64(pkg, gg, csr, vid, oracle_graph, batches=16):
    """configs[2]: SF100 shortest_path, 64-source bitset BFS to fixpoint, 16 batches on the benchmark's CSR."""
    srcs = [pkg.datagen.pick_sources(vid, 64, 0x5EED, batch=b) for b in range(batches)]
    gg.bfs64(csr, srcs[0], -1, fetch=False)  # warm-up
    # wall time without event records (a level is four launches, two records each would dominate) ...
    t0 = time.perf_counter()
    te = act = lv = 0
    for b in srcs:
        _, st = gg.bfs64(csr, b, -1, fetch=False)
        te += st["traversed_edges"]
        act += st["active_vertices"]
        lv += st["levels"]
    dt = time.perf_counter() - t0
    # ... kernel time from a second, profiled pass over the same batches
    gg.profile_reset()
    gg.profile_select(None)
    gg.profile(True)
    for b in srcs:
        gg.bfs64(csr, b, -1, fetch=False)
    gg.profile(False)
    prof = gg.profile_get()
    kern_ms = sum(v[1] for k, v in prof.items() if k.startswith("bfs_"))
    V = csr.V
    alg = 8 * V * lv + 16 * act + 24 * te + 24 * V * lv  # SURVEY.md §8d per-level formula summed over the levels
    out = {"workload": "LDBC SNB SF100 shortest_path(Person, Person): 64-source bitset BFS to fixpoint",
           "batches": batches, "wall_ms_per_batch": dt / batches * 1e3, "kernel_ms_per_batch": kern_ms / batches,
           "levels_per_batch": lv / batches, "value": te / dt, "unit": "traversed edges/s",
           "roofline": {"bound": "hbm", "achieved": alg / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": alg / (kern_ms * 1e-3) / HBM_PEAK, "traffic": _pmc_traffic("sf100/bfs64_batch/n1"),
                        "traffic_note": "HBM-side bytes per 64-source batch (2*FETCH_SIZE + WRITE_SIZE over the batch's kernels, "
                                        "profiles/pmc_traffic.json: counter passes of an earlier run, not measured in this run)",
                        "algorithmic_bytes_per_batch": alg / batches,
                        "note": "per-level algorithmic bytes 8V + 16Va + 24TE + 24V over the BFS kernels' time"},
           "kernels_us_per_batch": {k: v[1] * 1e3 / batches for k, v in prof.items() if k.startswith("bfs_")}}
    if oracle_graph is not None:
        t0 = time.perf_counter()
        d, ost = oracle_graph.bfs64(oracle_graph.lookup(srcs[0]), -1)
        cdt = time.perf_counter() - t0
        dist, gst = gg.bfs64(csr, srcs[0], -1)
        out["parity"] = bool(np.array_equal(d, dist) and ost == gst)
        out["cpu_port"] = {"value": ost["traversed_edges"] / cdt, "unit": "traversed edges/s", "cores": 1,
                           "sample": f"one 64-source batch, C oracle bitset BFS, {cdt:.2f}s"}
        ref = _reference_cte_baseline(pkg)
        if ref:
            out["cpu_reference"] = ref
    return out


def _reference_cte_baseline(pkg, scale="sf1", n_src=64, max_hops=5):
    """The reference's own shortest-path statement (recursive CTE friends / friends_shortest, bi-10) on a bounded
    sample: SF1, 64 seeds, hopCount < 5, all host threads, 1 cold + 2 hot runs in a child process
    (oracle/ref_cte_bench.py).  Traversed edges of the same search from the C oracle's BFS with the same hop bound."""
    import subprocess
    from oracle import ref_duckdb as R
    from tests import oracle_lib
    if not R.available():
        return None
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_cte_bench.py"), scale, str(n_src), str(max_hops)],
                           capture_output=True, text=True, timeout=300, cwd=ROOT)
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        vid, src, dst = pkg.datagen.ldbc(scale)
        rc, g = oracle_lib.load().csr_build(vid, src, dst)
        seeds = pkg.datagen.pick_sources(vid, n_src, 0x5EED)
        d, ost = g.bfs64(g.lookup(seeds), max_hops)
        rows = int((d >= 0).sum())
        g.close()
        return {"value": ost["traversed_edges"] / line["median_hot_s"], "unit": "traversed edges/s", "kind": "reference",
                "cores": line["threads"], "cold_s": line["cold_s"], "hot_s": line["hot_s"], "median_hot_s": line["median_hot_s"],
                "rows_match_oracle": bool(rows == line["rows"]),
                "sample": f"reference DuckDB (oracle/_ref/libduckdb.so) recursive CTE friends/friends_shortest, LDBC {scale}, "
                          f"{n_src} seeds, hopCount < {max_hops}: {line['rows']} result rows, TE={ost['traversed_edges']} per run "
                          "(the GPU section above is SF100 to fixpoint; the statement does not finish at that size in the bench's time)"}
    except Exception as e:  # the baseline is a report, never a reason to lose the bench line
        return {"error": str(e)[:200]}


def extra_materialised(pkg, orc, device):
    """configs[1]: SF10 Person-KNOWS*2..2-Person with the rows written to HBM as int64 id columns."""
    vid, src, dst = pkg.datagen.ldbc("sf10")
    gg = pkg.GG(device)
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    csr = gg.build_csr()
    gg.expand_khop_result(csr, 2).close()  # warm-up (allocations)
    gg.profile_reset()
    gg.profile_select(None)
    gg.profile(True)
    t0 = time.perf_counter()
    res = gg.expand_khop_result(csr, 2)
    dt = time.perf_counter() - t0
    gg.profile(False)
    prof = gg.profile_get()
    rows = res.rows(2)
    written = rows * 24
    kname = "mat_mid2" if prof.get("mat_mid2", (0, 0.0))[0] else "mat_last"
    k = prof.get(kname, (0, 0.0))
    out = {"workload": "LDBC SNB SF10 Person-KNOWS-Person-KNOWS-Person, all persons as sources, rows materialised in HBM "
                       "(3 int64 id columns)",
           "rows": rows, "bytes_written": written, "wall_ms": dt * 1e3,
           "kernels_ms": {n: v[1] for n, v in prof.items() if v[1] > 0.02}}
    if k[0]:
        out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": written / (k[1] * 1e-3) / 1e9,
                           "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": written / (k[1] * 1e-3) / HBM_PEAK,
                           "traffic": _pmc_traffic("sf10/mat_mid2/n1") if kname == "mat_mid2" else None,
                           "avg_launch_ms": k[1] / k[0],
                           "note": "bytes actually written by the kernel (rows x 3 x 8) over its time",
                           "traffic_note": "2*FETCH_SIZE + WRITE_SIZE of an earlier counter run (profiles/pmc_traffic.json); "
                                           "WRITE_SIZE tallies this kernel's nontemporal 16-byte stores at half their "
                                           "bytes (12.8 GB counted, 25.5 GB written by construction and read back by "
                                           "the digest): the kernel fetches 0.04 GB and writes each row once"}
    # parity over ALL rows: the digest of what was written (gg_result_digest maps every id of every row back to its
    # dense index and sums the row hashes) against the count-mode expansion's and the oracle's digest of the same walks
    rc, g = orc.csr_build(vid, src, dst)
    ost = g.khop(2, 2)
    n_dig, dig = res.digest(csr, 2)
    counted = gg.expand_khop(csr, 2, 2)
    out["parity"] = bool(rows == ost["rows"][2] == n_dig == counted["rows"][2] and
                         dig == ost["digest"][2] == counted["digest"][2])
    out["parity_note"] = ("rows and digest over all materialised rows (device-side, from the id columns in HBM) == "
                          "count-mode expansion == oracle")
    res.close()
    g.close()
    csr.close()
    gg.close()
    return out


def extra_materialised_parts(gg, csr, counted, budget_gb=16.0):
    """SF100 2-hop MATCH with the rows written to HBM: 12.8 G rows x 3 int64 columns = 306 GB do not fit 288 GB, so
    the result is produced part by part — middle-vertex ranges of near-equal work (gg_khop_partition_mid), each
    through the product kernel (gg_expand_khop_mid_result), handed over and freed — the way the substituted join
    streams it (host/gg_operators.cpp).  Timed: expansion calls only (count + materialise per part); the digest of
    every part's rows (gg_result_digest, all rows, device-side) is taken outside the timed region and must add up to
    the count-mode expansion's."""
    total_rows = counted["rows"][2]
    n_parts = max(1, int(np.ceil(total_rows * 24 / (budget_gb * 2**30))))
    bounds = gg.khop_partition_mid(csr, n_parts)
    gg.profile_reset()
    gg.profile_select(["mat_mid2"])
    gg.profile(True)
    rows = dig = 0
    t_total = 0.0
    import torch

    for lo, hi in zip(bounds[:-1], bounds[1:]):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = gg.expand_khop_mid_result(csr, lo, hi, k_min=2)
        torch.cuda.synchronize()  # (the materialising kernel is launched asynchronously)
        t_total += time.perf_counter() - t0
        n, d = res.digest(csr, 2)
        rows += n
        dig = (dig + d) & 0xFFFFFFFF
        res.close()
    gg.profile(False)
    prof = gg.profile_get()
    k = prof.get("mat_mid2", (0, 0.0))
    out = {"workload": "LDBC SNB SF100 Person-KNOWS-Person-KNOWS-Person, all persons as sources, rows materialised in HBM "
                       f"in {n_parts} middle-vertex parts of <= {budget_gb:.0f} GiB (3 int64 id columns each)",
           "parts": n_parts, "rows": rows, "bytes_written": rows * 24, "wall_s": t_total,
           "rows_per_s": rows / t_total, "bytes_per_s": rows * 24 / t_total,
           "parity": bool(rows == total_rows and dig == counted["digest"][2]),
           "parity_note": "sum over the parts of rows and of the device-side digests of all materialised rows == count-mode expansion"}
    if k[0]:
        out["roofline"] = {"bound": "hbm", "kernel": "mat_mid2", "achieved": rows * 24 / (k[1] * 1e-3) / 1e9,
                           "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": rows * 24 / (k[1] * 1e-3) / HBM_PEAK,
                           "traffic": None, "avg_launch_ms": k[1] / k[0], "launches": k[0],
                           "note": "bytes written by the kernel (rows x 3 x 8) over its time, all parts"}
    return out


def extra_connectedsegments(pkg, device, copies=1024, steps=10):
    """configs[4]: Train Benchmark ConnectedSegments at SF1024: both CSR builds + 5-hop walks + same-sensor filter."""
    from duckdb_pgq_amd import datagen
    from tests import trainbenchmark as tb

    t = tb.tables()
    base = {"te": tb.load("TrackElement")[:, 0], "sensors": tb.load("Sensor")[:, 0], "seg": t["Segment"][:, 0],
            "ct": t["connectsTo"], "mb": t["monitoredBy"]}
    r = datagen.replicate_tables(base, copies)
    gg = pkg.GG(device)
    gg.set_edge_rowid(False)
    vertices = np.concatenate([r["te"], r["sensors"]])

    def step():
        gg.staging_clear()
        gg.append_vertices(vertices)
        gg.append_edges(r["ct"][:, 0], r["ct"][:, 1])
        path_csr = gg.build_csr()
        gg.staging_clear_edges()
        gg.append_edges(r["mb"][:, 0], r["mb"][:, 1])
        filter_csr = gg.build_csr()
        rows = gg.connected_paths_same_neighbour(path_csr, filter_csr, 5, sources=r["seg"])
        path_csr.close()
        filter_csr.close()
        return rows

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        rows = step()
    dt = time.perf_counter() - t0
    shift = np.arange(copies, dtype=np.int64) * r["_stride"]
    want = (tb.CONNECTEDSEGMENTS_GOLDEN[None, :, :] + shift[:, None, None]).reshape(-1, 7)
    key = lambda a: a[np.lexsort(a.T[::-1])]  # noqa: E731
    ok = bool(rows.shape == want.shape and np.array_equal(key(rows), key(want)))
    gg.close()
    return {"workload": f"Train Benchmark ConnectedSegments SF{copies} (the reference's SF1 tables x{copies}, shifted ids): "
                        "staging + 2 CSR builds + 5-hop walks + same-sensor filter per step",
            "steps": steps, "ms_per_step": dt / steps * 1e3, "result_rows": int(rows.shape[0]),
            "connectsTo_rows": int(r["ct"].shape[0]), "monitoredBy_rows": int(r["mb"].shape[0]), "parity": ok,
            "parity_note": "result == the reference's four golden rows (connectedsegments.benchmark:34-38), shifted per copy"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sf100", choices=["sf0.1", "sf1", "sf10", "sf100"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the materialised / bfs64 / connectedsegments sections")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="target seconds per reference run (6 + 2 runs per leg)")
    ap.add_argument("--legacy-build", action="store_true", help="diagnostic: the multi-pass LSD build")
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
    # launched by torch.distributed.run (WORLD_SIZE is set): the process group exists even for ONE rank, so that a
    # one-GPU box can run the RCCL calls of the N > 1 path (tests/test_multirank_gpu.py); plain `python bench.py`
    # — the driver's N = 1 run — has no group and no collective
    if world > 1 or "WORLD_SIZE" in os.environ:
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
    if args.legacy_build:
        gg.force_legacy_build(True)
    t0 = time.perf_counter()
    gg.chunk_rows = 122_880  # one DuckDB row group per append (storage/table/row_group.hpp:38-39)
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    gg.staging_sync()
    t_stage = time.perf_counter() - t0
    if rank == 0:
        log(f"{args.workload}: V={V} knows rows={R} ({R_local} on this rank); datagen {t_gen:.1f}s, "
            f"staging (PCIe) {t_stage*1e3:.1f} ms")

    def build():
        # all persons are sources; with N ranks this rank builds only the CSR rows of the vertices it
        # owns (owner = hash(id) mod N) and produces the walks whose middle vertex it owns
        if args.shard_of > 1:
            return gg.build_csr_shard(0, args.shard_of)
        return gg.build_csr_shard(rank, world) if world > 1 else gg.build_csr()

    def run_steps(n):
        """n complete steps — build, expansion, combine of the ranks' results — with the host side pipelined by one
        step when there are several ranks: the NEXT step's build is launched (gg_csr_build returns once its status is
        known, two thirds of its kernels still queued) before the ranks' counts of THIS step are all-reduced, so the
        collective and its host round trips run beside those kernels instead of in front of them.  Every step's
        build, expansion and combine lie inside the caller's timed region; no CSR is built that is not expanded."""
        vec = st = None
        c = build()
        for i in range(n):
            st = gg.expand_khop(c, 1, 2)
            c.close()
            c = build() if i + 1 < n else None
            vec = sharding.combine(sharding.stats_to_vec(st), dist, device="cuda")
        return vec, st

    def step():
        return run_steps(1)

    if args.warmup:
        run_steps(args.warmup)
    # HIP events around a launch are not free (two records per launch: ~0.3 ms of a step when all ~25 launches
    # are timed), so the timed region times only the kernels the roofline can name; the per-kernel table of
    # everything else comes from a few extra, untimed steps afterwards.
    BUILD_KERNELS = ["densify_pairs", "partition_dual", "sub_sort", "leaf_rows",     # bucketed build
                     "densify_hist", "densify_shard", "radix_scatter"]                # multi-pass build (shards, > 2^22 vertices)
    EXPAND_KERNELS = ["expand_mid2", "expand_fused2"]
    gg.profile_reset()
    gg.profile_select(BUILD_KERNELS + EXPAND_KERNELS)
    gg.profile(True)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    tot, st_local = run_steps(args.steps)
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

    # ---- rooflines -------------------------------------------------------------------------------------------------
    # HBM-bound kernels, algorithmic bytes per launch: densification 32E + 8V; one CSR
    # without rowid 32E + 16V, which the bucketed build spreads over three kernels (partition, sub-bucket sort, rows)
    # and the multi-pass build over three scatter passes — each launch is charged a third (the reverse CSR the product
    # kernel needs is built by the same launches and is NOT in the algorithmic count).
    # VALU-bound kernel: the 2-hop product kernel performs one v_xad_u32 per 2-hop walk.
    te_l, fr_l = st_local["traversed_edges"], st_local["frontier_entries"]
    walks2_l = st_local["rows"][2]
    csr_bytes = 32 * R_local + 16 * V
    alg_hbm = {"densify_pairs": 32 * R_local + 8 * V, "densify_hist": 32 * R_local + 8 * V,
               "densify_shard": 32 * R_local + 8 * V, "partition_dual": csr_bytes / 3, "sub_sort": csr_bytes / 3,
               "leaf_rows": csr_bytes / 3, "radix_scatter": 8 * R_local}
    traffic_tab = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic_tab = json.load(open(tpath))
        except Exception:
            traffic_tab = {}
    traffic_src = traffic_tab.get("_source", "profiles/pmc_traffic.json") + " (rocprofv3 --pmc passes of an earlier run of this command, not measured in this run)"

    def record(name):
        launches, total_ms = prof.get(name, (0, 0.0))
        if not launches:
            return None
        avg_s = total_ms / launches * 1e-3
        tr = traffic_tab.get(f"{args.workload}/{name}/n{world}")
        if name in EXPAND_KERNELS:
            ops = walks2_l
            return {"bound": "valu", "kernel": name, "achieved": ops / avg_s / 1e9, "peak": VALU_XAD_PEAK / 1e9,
                    "unit": "Gop/s", "frac": ops / avg_s / VALU_XAD_PEAK, "traffic": tr, "traffic_source": traffic_src if tr else None,
                    "avg_launch_ms": avg_s * 1e3, "launches_per_step": launches / args.steps, "ops_per_launch": int(ops),
                    "note": "one v_xad_u32 per 2-hop walk; peak = measured issue rate of that instruction (scripts/ubench_valu.hip); "
                            "the kernel reads each CSR row once (traffic), so an HBM fraction would not describe it"}
        a = alg_hbm[name]
        return {"bound": "hbm", "kernel": name, "achieved": a / avg_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": a / avg_s / HBM_PEAK, "traffic": tr, "traffic_source": traffic_src if tr else None,
                "avg_launch_ms": avg_s * 1e3, "launches_per_step": launches / args.steps,
                "algorithmic_bytes_per_launch": int(a)}

    recs = {k: record(k) for k in prof}
    recs = {k: v for k, v in recs.items() if v}
    for name in ("densify_pairs", "densify_shard"):
        if name in recs:
            recs[name]["request_rate"] = _probe_rate_record(V, 2 * R_local, recs[name]["avg_launch_ms"] * 1e-3)
    dom = max(recs, key=lambda k: prof[k][1], default=None)
    roof = recs.get(dom)
    # phases: the timed region's figure where a kernel was timed there, the untimed pass's otherwise
    per_step_ms = {k: v[1] / TABLE_STEPS for k, v in prof_all.items()}
    per_step_ms.update({k: v[1] / args.steps for k, v in prof.items()})
    expand_names = {"expand_mid2", "expand_fused2", "reduce_partials", "tile_partition"}
    t_expand = sum(ms for k, ms in per_step_ms.items() if k in expand_names) * 1e-3
    t_build = sum(ms for k, ms in per_step_ms.items() if k not in expand_names) * 1e-3
    alg_build = (32 * R_local + 8 * V) + csr_bytes  # densification + one CSR without rowid
    phases = {}
    if t_build > 0:
        phases["csr_build"] = {"bound": "hbm", "kernel_ms": t_build * 1e3, "algorithmic_bytes": alg_build,
                               "achieved": alg_build / t_build / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": alg_build / t_build / HBM_PEAK,
                               "note": "all build kernels; the build also makes the reverse CSR, which is not in the algorithmic count"}
    if t_expand > 0:
        phases["expand"] = {"bound": "valu", "kernel_ms": t_expand * 1e3, "ops": int(walks2_l),
                            "achieved": walks2_l / t_expand / 1e9, "peak": VALU_XAD_PEAK / 1e9, "unit": "Gop/s",
                            "frac": walks2_l / t_expand / VALU_XAD_PEAK,
                            "traffic": traffic_tab.get(f"{args.workload}/expand_mid2/n{world}")}
    kernels = {k: {"launches": v[0], "avg_us": (v[1] / v[0] * 1e3 if v[0] else 0.0),
                   "us_per_step": v[1] * 1e3 / TABLE_STEPS, "timed_region": False} for k, v in prof_all.items()}
    kernels.update({k: {"launches": v[0], "avg_us": (v[1] / v[0] * 1e3 if v[0] else 0.0),
                        "us_per_step": v[1] * 1e3 / args.steps, "timed_region": True} for k, v in prof.items()})

    if args.shard_of > 1:
        log(f"shard 0 of {args.shard_of}: {ms_per_step:.3f} ms/step; kernels us/step:",
            {k: round(v["us_per_step"]) for k, v in kernels.items()})
        print(json.dumps({"diagnostic": f"shard 0 of {args.shard_of}", "ms_per_step": ms_per_step,
                          "kernels_us_per_step": {k: v["us_per_step"] for k, v in kernels.items()}}), flush=True)
        gg.close()
        return
    if rank == 0:
        extra = {}
        oracle_graph = None
        if not args.no_cpu:
            extra = cpu_baseline(vid, src_all, dst_all, V, args.cpu_seconds, with_reference=(world == 1))
            ost = extra.pop("oracle_stats")
            oracle_graph = extra.pop("oracle_graph")
            parity = (ost["rows"][1] == rows1 and ost["rows"][2] == rows2 and ost["digest"][1] == dig1
                      and ost["digest"][2] == dig2 and ost["traversed_edges"] == te_total)
            extra["parity_vs_oracle"] = bool(parity)
            if not parity:
                log("PARITY FAILURE", ost, tot)
        if world == 1 and not args.no_extras:
            try:
                if args.workload == "sf100":
                    c = gg.build_csr()
                    extra["bfs64"] = extra_bfs64(pkg, gg, c, vid, oracle_graph)
                    extra["materialised_sf100_parts"] = extra_materialised_parts(gg, c, gg.expand_khop(c, 2, 2))
                    c.close()
                if oracle_graph is not None:
                    oracle_graph.close()
                    oracle_graph = None
                from tests import oracle_lib

                extra["materialised"] = extra_materialised(pkg, oracle_lib.load(), device_index)
                extra["connectedsegments"] = extra_connectedsegments(pkg, device_index)
            except Exception as e:  # an extra section must not cost the headline line
                extra["extras_error"] = repr(e)
        if oracle_graph is not None:
            oracle_graph.close()
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
                       "traversed_edges": int(te_total),
                       "collective_backend": dist.get_backend() if dist is not None else None, "parallelism": f"vertex-ownership shards x{world} (edge table hash-partitioned by endpoint owner, vertex table replicated, CSR + expansion sharded, no data-path collective)"},
            "roofline": roof,
            "roofline_kernels": recs,
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
