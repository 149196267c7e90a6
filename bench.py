#!/usr/bin/env python3
"""bench.py — traversed edges/sec on LDBC SNB 2-hop MATCH (Person-KNOWS*1..2-Person), MI355X.

Two timed regions over the synthetic LDBC-shaped tables already resident in HBM, both build-inclusive:

  A  count + checksum (the bench line's `value`, as in every earlier round).  One step = gg_csr_build (densify ids,
     bucket partition, vertex sort, rows: forward + reverse CSR) + the counting 2-hop expansion: what the reference
     does for `SELECT count(*)` over the join chain as hash-join build + probe.  The expansion kernel reads each CSR
     row once and folds a 32-bit checksum per walk: it is bound by vector-ALU issue, NOT by HBM, and carries no HBM
     fraction (config.workload says so).
  B  rows materialised (`match_materialised`, the roofline-bearing region).  One step = gg_csr_build + EVERY 2-hop row
     (person, friend, friend of friend) written to HBM as three int64 id columns, in middle-vertex parts of at most
     --mat-budget-gb (12.8 G rows x 24 B = 306 GB do not fit 288 GB at once; the substituted hash join streams its
     result the same way, host/gg_operators.cpp): what the reference's probe side does when the MATCH returns rows
     (ScanStructure::NextInnerJoin / GatherResult, src/execution/join_hashtable.cpp:442-476).  Dominant kernel
     k_mat_mid2, HBM-bound: bytes written / its time / 8 TB/s.  The top-level `roofline` is the kernel with the
     largest total time over BOTH regions — k_mat_mid2 — with HBM traffic from counter passes made in this run.

With N > 1 ranks the vertices are hash-partitioned (owner = hash(person id) mod N) and so is the edge table: a rank
holds the `knows` rows whose source or destination it owns (every row on at most two ranks; the 3.6 MB person table is
replicated), builds only the CSR rows of the vertices it owns (gg_csr_build_shard) and produces — counts in A, writes
in B — the walks whose middle vertex it owns; no data-path collective.  Region A's per-step result is six words left
on the device (gg_expand_khop_dev), all-reduced in place and copied to the host once (sharding.combine_dev); the
host launches step i + 1's build before it waits for step i's words.  `value` = traversed edges of the whole query /
max-over-ranks time ("strong" scaling: the query is fixed, ranks split it).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sf100|sf10|sf1] [--no-cpu] [--no-extras]
                    [--mat-steps M] [--mat-budget-gb G] [--no-pmc]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement):
  roofline           the kernel with the largest total time over the timed regions (k_mat_mid2, region B): bytes it
                     writes per launch / average launch time / 8 TB/s; `traffic` = 2 x FETCH_SIZE + WRITE_SIZE per
                     launch from two rocprofv3 --pmc passes of this same program run as a child process in this run
                     (--pmc-child; counter rules: profiles/r04_counter_calibration.txt)
  match_materialised region B: TE/s for the same traversed edges, rows/s, bytes/s, parity (device-side digest over
                     all rows of every part, untimed pass), its kernels and the build phase's HBM record
  roofline_count_step / roofline_kernels / roofline_phases   region A: the densification (HBM), the build phase (HBM),
                     the counting expansion (VALU issue: one v_xad_u32 per walk)
  cpu_baseline       the compiled reference (oracle/_ref/libduckdb.so; else the C oracle) on this box's host cores,
                     benchmark_runner protocol (1 cold + 5 hot runs, median hot: benchmark/benchmark_runner.cpp:132-147)
                     on a bounded sample, plus a threads=1 figure, the CPU model, and the one full run committed as
                     profiles/r04_cpu_full_sf100.json
  materialised, bfs64, connectedsegments   the other BASELINE.json configs on this GPU, each with its own parity
                     boolean (N = 1 only; --no-extras skips them)
  build_edge_only    the build the reference's own 2-hop SQL takes (no vertex table in the pattern): endpoint set + CSR
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
    try:  # the one full run of the whole workload on the reference (scripts/cpu_full_sf100.py, committed output)
        full = json.load(open(os.path.join(ROOT, "profiles", "r04_cpu_full_sf100.json")))
        if full.get("traversed_edges") == te_full:
            out["cpu_baseline"]["full_workload_measured"] = {
                "value": full["value"], "unit": full["unit"], "seconds": full["seconds_1hop_plus_2hop"],
                "threads": full["threads"], "cpu_model": full["cpu_model"], "counts_match_oracle": full["counts_match_oracle"],
                "source": "profiles/r04_cpu_full_sf100.json: count(*) of the 1-hop and 2-hop chains over ALL knows rows, run once "
                          "on a box of this pool in round 4 (not in this run); the sample above extrapolates low because the "
                          "probe rate grows with the sample"}
    except Exception:
        pass
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
    import torch

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = gg.expand_khop_result(csr, 2)
    torch.cuda.synchronize()  # (the materialising kernel is launched asynchronously: wait for it before stopping the clock)
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
                           "traffic": None, "avg_launch_ms": k[1] / k[0],
                           "note": "bytes actually written by the kernel (rows x 3 x 8) over its time; counters for this kernel: "
                                   "the SF100 parts of match_materialised (same kernel) and profiles/r04_counter_calibration.txt "
                                   "(WRITE_SIZE of the SF10 launch = 25.52 GB for 25.51 GB of rows)"}
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


MAT_KERNELS = ["mat_mid2", "mat_mid2_prepare", "mat_tile_entries"]


def match_step(gg, build, budget_bytes, verify=False):
    """One step of region B on this rank: build, then every 2-hop row whose middle vertex this rank owns (all of them
    at N = 1) written to HBM as three int64 id columns, part by part under the device-memory budget.  The number of
    rows comes from degrees (gg_khop_count) and sizes the parts; parts are middle-vertex ranges of near-equal work
    (gg_khop_partition_mid), each produced by gg_expand_khop_mid_result, handed over and freed.  verify: every part's
    rows are digested on the device (gg_result_digest maps every id of every row back to its dense index and sums the
    row hashes) — the untimed parity pass."""
    c = build()
    total = gg.khop_count(c, 2, 2)[2]
    n_parts = max(1, int(-(-total * 24 // budget_bytes)))
    bounds = gg.khop_partition_mid(c, n_parts) if n_parts > 1 else [0, c.V]
    rows = dig = 0
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        res = gg.expand_khop_mid_result(c, lo, hi, k_min=2, with_stats=False)  # (rows only: the count came from degrees)
        if verify:
            n, d = res.digest(c, 2)
            assert n == res.rows(2)
            dig = (dig + d) & 0xFFFFFFFF
        rows += res.rows(2)
        res.close()
    c.close()
    return rows, dig, n_parts


def _kernel_short_name(full):
    """'void gg::k_densify_pairs<false>(long const*, ...)' -> 'k_densify_pairs'; '(anonymous namespace)::k_mat_mid2(...)'
    -> 'k_mat_mid2'."""
    name = full.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    name = name.split("(")[0].split("<")[0]
    return name.split("::")[-1].strip()


def _pmc_child(args):
    """--pmc-child: the program the counter passes profile — one GPU, the same tables, one warm-up and one counted step
    of region A and of region B, nothing printed but a marker.  Run as `rocprofv3 --pmc X --kernel-trace -- python3
    bench.py --pmc-child` by measure_pmc_traffic; the counters of every dispatch land in the pass's CSV."""
    import duckdb_pgq_amd as pkg

    vid, src, dst = pkg.datagen.ldbc(args.workload)
    gg = pkg.GG(0)
    gg.set_edge_rowid(False)
    gg.chunk_rows = 122_880
    gg.append_vertices(vid)
    gg.append_edges(src, dst)
    gg.staging_sync()
    budget = int(args.mat_budget_gb * 2**30)
    for _ in range(2):
        c = gg.build_csr()
        gg.expand_khop(c, 1, 2)
        c.close()
        match_step(gg, gg.build_csr, budget)
    gg.close()
    print("pmc-child done", flush=True)


def measure_pmc_traffic(args, timeout_s=300):
    """HBM-side bytes per kernel launch, measured in THIS run: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE — they do
    not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") over `python3 bench.py --pmc-child` as a child
    process.  Units and corrections as calibrated in profiles/r04_counter_calibration.txt on known byte counts: both
    counters are in KiB; WRITE_SIZE is exact for the store forms used here, FETCH_SIZE reports half of the bytes read.
    Returns {kernel: {"launches", "fetch_bytes", "write_bytes", "traffic"}} with per-launch averages, or {"error": ...}."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return {"error": "rocprofv3 not on PATH"}
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return {"error": "already running under a profiler"}
    out = {}
    tmp = tempfile.mkdtemp(prefix="gg_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3",
                   os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", args.workload,
                   "--mat-budget-gb", str(args.mat_budget_gb)]
            env = dict(os.environ, TMPDIR="/tmp")
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                    text=True, start_new_session=True)
            try:
                text, _ = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)  # (the process group this call started, nothing else)
                proc.wait()
                return {"error": f"{counter} pass timed out after {timeout_s}s"}
            if proc.returncode != 0 or "pmc-child done" not in text:
                return {"error": f"{counter} pass failed (rc {proc.returncode}): {text[-300:]}"}
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return {"error": f"{counter} pass wrote no counter file"}
            for f in files:
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != counter:
                        continue
                    name = _kernel_short_name(r["Kernel_Name"])
                    rec = out.setdefault(name, {"launches": {}, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
                    rec[counter] += float(r["Counter_Value"]) * 1024.0
                    rec["launches"][counter] = rec["launches"].get(counter, 0) + 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for name, rec in out.items():
        n = max(rec["launches"].values()) if rec["launches"] else 0
        if not n:
            continue
        fetch = 2.0 * rec["FETCH_SIZE"] / max(rec["launches"].get("FETCH_SIZE", n), 1)
        write = rec["WRITE_SIZE"] / max(rec["launches"].get("WRITE_SIZE", n), 1)
        res[name] = {"launches": n, "fetch_bytes": fetch, "write_bytes": write, "traffic": fetch + write}
    return res


def extra_build_edge_only(pkg, device, src, dst, steps=10):
    """The build the reference's own 2-hop SQL takes (`k1.k_person2id = k2.k_person1id`, benchmark/ldbc/queries/
    interactive-complex-3.sql:11: no vertex table in the pattern): the vertex set is the distinct endpoint ids of the
    edge table (gg_vertices_from_edges: a pair-probed id table filled in one pass over the rows, one bucket sort of the
    ids) and then the CSR build of the headline.  Wall time per statement-side step, per-kernel times, HBM records on
    SURVEY 8d's bytes (endpoint set: both id columns read once, the vertex table written)."""
    import torch

    gg = pkg.GG(device)
    gg.set_edge_rowid(False)
    gg.chunk_rows = 122_880
    gg.append_edges(src, dst)
    gg.staging_sync()

    def step():
        n = gg.vertices_from_edges()
        return n, gg.build_csr()

    for _ in range(3):
        step()[1].close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()[1].close()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    gg.profile_reset()
    gg.profile_select(None)
    gg.profile(True)
    for _ in range(3):
        step()[1].close()
    gg.profile(False)
    prof = gg.profile_get()
    V, c = step()
    E = int(src.size)
    rows2 = gg.khop_count(c, 2, 2)[2]
    c.close()
    gg.close()
    set_ms = sum(v[1] for k, v in prof.items() if k.startswith("set_")) / 3
    all_ms = sum(v[1] for v in prof.values()) / 3
    alg_set = 16 * E + 8 * V
    alg_build = (32 * E + 8 * V) + (32 * E + 16 * V)
    return {"workload": "LDBC SNB SF100 knows table ALONE (edge-only 2-hop idiom): distinct endpoint ids, sorted + CSR build",
            "vertices": int(V), "edge_rows": E, "rows_2hop": int(rows2), "ms_per_step": wall * 1e3,
            "kernels_us_per_step": {k: v[1] / 3 * 1e3 for k, v in prof.items()},
            "endpoint_set": {"bound": "hbm", "kernel_ms": set_ms, "algorithmic_bytes": alg_set,
                             "achieved": alg_set / (set_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                             "frac": alg_set / (set_ms * 1e-3) / HBM_PEAK,
                             "note": "bound by one random 16-byte probe per endpoint into a 6.6 MB table (as the "
                                     "densification of the build that follows), not by the bytes streamed"},
            "whole": {"bound": "hbm", "kernel_ms": all_ms, "algorithmic_bytes": alg_set + alg_build,
                      "achieved": (alg_set + alg_build) / (all_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                      "frac": (alg_set + alg_build) / (all_ms * 1e-3) / HBM_PEAK},
            "parity": bool(V == np.unique(np.concatenate([src, dst])).size),
            "parity_note": "vertex count == numpy's distinct endpoint count (tests/test_gpu_parity.py compares the table and the CSR)"}


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
    ap.add_argument("--mat-steps", type=int, default=5, help="timed steps of region B (build + rows materialised in HBM)")
    ap.add_argument("--mat-budget-gb", type=float, default=40.0,
                    help="device-memory budget of one materialised part (GiB; the operators' GG_RESULT_BUDGET_MB default)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (traffic from profiles/)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return _pmc_child(args)

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
        """n complete steps of region A — build, counting expansion, combine of the ranks' results.  With a process
        group the expansion leaves its six result words on the device (gg_expand_khop_dev: no host synchronisation),
        the NEXT step's build is launched (gg_csr_build returns once its status is known, two thirds of its kernels
        still queued), and only then are the words all-reduced in place and copied to the host once
        (sharding.combine_dev) — the collective and its one round trip run beside the build's queued kernels.  Every
        step's build, expansion and combine lie inside the caller's timed region; no CSR is built that is not
        expanded."""
        vec = st = None
        c = build()
        for i in range(n):
            if dist is None:
                st = gg.expand_khop(c, 1, 2)
                c.close()
                c = build() if i + 1 < n else None
                vec = sharding.stats_to_vec(st)
            else:
                words = gg.expand_khop_dev(c, 1)
                c.close()  # (blocks go back to the pool in stream order; the queued expansion still reads them first)
                c = build() if i + 1 < n else None
                vec = sharding.combine_dev(gg, words, dist)
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

    # ---- region B: build + every 2-hop row written to HBM (this rank: the rows whose middle vertex it owns) ------------
    budget = int(args.mat_budget_gb * 2**30)
    mat = None
    if args.mat_steps > 0 and not args.legacy_build:
        match_step(gg, build, budget)  # warm-up: the pool takes the parts' blocks
        gg.profile_reset()
        gg.profile_select(BUILD_KERNELS + MAT_KERNELS)
        gg.profile(True)
        step_rows = []
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.mat_steps):
            rows_l, _, n_parts_l = match_step(gg, build, budget)
            rt = torch.tensor([rows_l], dtype=torch.int64, device="cuda")
            if dist is not None:
                dist.all_reduce(rt)  # the query's row count, once per step (the rows themselves stay sharded in HBM)
            step_rows.append(rt)
        torch.cuda.synchronize()
        barrier()
        mat_elapsed = time.perf_counter() - t0
        gg.profile(False)
        mat_prof = gg.profile_get()
        if dist is not None:
            tmax = torch.tensor([mat_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            mat_elapsed = float(tmax.item())
        # untimed parity pass: every part's rows digested on the device, summed over parts and ranks
        v_rows, v_dig, _ = match_step(gg, build, budget, verify=True)
        vt = torch.tensor([v_rows, v_dig], dtype=torch.int64, device="cuda")
        if dist is not None:
            dist.all_reduce(vt)
        v_rows, v_dig = int(vt[0].item()), int(vt[1].item()) & 0xFFFFFFFF
        mat = {"elapsed": mat_elapsed, "prof": mat_prof, "rows_local": rows_l, "parts_local": n_parts_l,
               "rows_steps": [int(t.item()) for t in step_rows], "verify_rows": v_rows, "verify_digest": v_dig}

    if st_local is None:  # (device-side combine: the rank's own counts never came to the host inside the timed region)
        c = build()
        st_local = gg.expand_khop(c, 1, 2)
        c.close()
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

    # ---- region B's records ------------------------------------------------------------------------------------------
    match = None
    if mat is not None:
        m_ms = mat["elapsed"] / args.mat_steps * 1e3
        k_l, k_ms = mat["prof"].get("mat_mid2", (0, 0.0))
        rows_l = mat["rows_local"]
        m_kernels = {k: {"launches": v[0], "avg_us": (v[1] / v[0] * 1e3 if v[0] else 0.0),
                         "us_per_step": v[1] * 1e3 / args.mat_steps} for k, v in mat["prof"].items()}
        mb_ms = sum(v[1] for k, v in mat["prof"].items() if k in BUILD_KERNELS) / args.mat_steps
        match = {
            "workload": f"LDBC SNB {args.workload.upper()} Person-KNOWS-Person-KNOWS-Person, all persons as sources: CSR build "
                        f"+ every 2-hop row written to HBM as 3 int64 id columns, in middle-vertex parts of <= "
                        f"{args.mat_budget_gb:g} GiB (gg_expand_khop_mid_result; rows beyond HBM's size stream through, "
                        "as through the hash join they replace)",
            "steps": args.mat_steps, "ms_per_step": m_ms, "parts_per_step_this_rank": mat["parts_local"],
            "rows": mat["rows_steps"][-1], "rows_this_rank": rows_l,
            "value": te_total * args.mat_steps / mat["elapsed"], "unit": "traversed edges/s",
            "rows_per_s": mat["rows_steps"][-1] * args.mat_steps / mat["elapsed"],
            "bytes_written_per_s": mat["rows_steps"][-1] * 24 * args.mat_steps / mat["elapsed"],
            "kernels": m_kernels,
            "placement": dict(zip(("column_sets_placed_by_probing", "fast_pairs_of_last_set"), gg.placement()),
                              note="3 fast pairs = each of the three result columns in its own memory rank class "
                                   "(DESIGN.md 4.2 Placement; GG_PLACE_TRACE=1 prints the probes)"),
            "parity": bool(all(r == rows2 for r in mat["rows_steps"]) and mat["verify_rows"] == rows2 and
                           mat["verify_digest"] == dig2),
            "parity_note": "rows of every timed step, and rows + device-side digest over ALL materialised rows of an untimed "
                           "pass (gg_result_digest per part, summed over parts and ranks), == the counting expansion of "
                           "region A (which parity_vs_oracle compares with the CPU oracle)",
        }
        if k_l:
            avg_s = k_ms / k_l * 1e-3
            bytes_per_launch = rows_l * 24 * args.mat_steps / k_l
            match["roofline"] = {
                "bound": "hbm", "kernel": "mat_mid2", "timed_region": "match_materialised",
                "achieved": bytes_per_launch / avg_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": bytes_per_launch / avg_s / HBM_PEAK, "traffic": None, "avg_launch_ms": avg_s * 1e3,
                "launches_per_step": k_l / args.mat_steps, "algorithmic_bytes_per_launch": int(bytes_per_launch),
                "kernel_share_of_step": k_ms / args.mat_steps / m_ms,
                "note": "algorithmic bytes = rows x 3 columns x 8 B written (SURVEY 8d's 8(k+1) per materialised row; its "
                        "8 B per traversed edge of the counting form are NOT charged: the product kernel reads each CSR row "
                        "once per tile, the counters put that at well under 1 % of the writes)"}
        if mb_ms > 0:
            alg_b = (32 * R_local + 8 * V) + (32 * R_local + 16 * V)
            match["build_phase"] = {"bound": "hbm", "kernel_ms": mb_ms, "algorithmic_bytes": alg_b,
                                    "achieved": alg_b / (mb_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                    "frac": alg_b / (mb_ms * 1e-3) / HBM_PEAK,
                                    "note": "the four large build kernels timed inside region B (the small ones are in `kernels` of region A)"}

    if args.shard_of > 1:
        log(f"shard 0 of {args.shard_of}: count step {ms_per_step:.3f} ms; kernels us/step:",
            {k: round(v["us_per_step"]) for k, v in kernels.items()})
        diag = {"diagnostic": f"shard 0 of {args.shard_of}", "ms_per_step": ms_per_step,
                "kernels_us_per_step": {k: v["us_per_step"] for k, v in kernels.items()}}
        if match is not None:
            log(f"shard 0 of {args.shard_of}: materialised step {match['ms_per_step']:.3f} ms, rows {match['rows_this_rank']}")
            diag["match_materialised"] = {"ms_per_step": match["ms_per_step"], "rows_this_rank": match["rows_this_rank"],
                                          "parts": match["parts_per_step_this_rank"],
                                          "kernels_us_per_step": {k: v["us_per_step"] for k, v in match["kernels"].items()},
                                          "roofline_frac": match.get("roofline", {}).get("frac"),
                                          "rows_digest_of_this_shard": [mat["verify_rows"], mat["verify_digest"]]}
        print(json.dumps(diag), flush=True)
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
                    c.close()
                    extra["build_edge_only"] = extra_build_edge_only(pkg, device_index, src_all, dst_all)
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
        # ---- HBM traffic from counters, measured now: the same program as a child under rocprofv3 --pmc --------------
        pmc = None
        if world == 1 and not args.no_pmc and "extras_error" not in extra:
            gg.close()  # (the child needs the device memory; nothing below uses this context)
            t0 = time.perf_counter()
            pmc = measure_pmc_traffic(args)
            log(f"counter passes: {time.perf_counter() - t0:.1f}s", pmc.get("error", "ok"))
        pmc_src = "2 x FETCH_SIZE + WRITE_SIZE per launch, rocprofv3 --pmc passes of `bench.py --pmc-child` made in this run"
        stale_src = "profiles/pmc_traffic.json (counter passes of an earlier run, NOT of this one)"

        def with_traffic(rec, kernel, stale_key):
            if rec is None:
                return rec
            if pmc and kernel in pmc:
                rec["traffic"] = int(pmc[kernel]["traffic"])
                rec["traffic_detail"] = {k: int(v) for k, v in pmc[kernel].items()}
                rec["traffic_source"] = pmc_src
            else:
                rec["traffic"] = traffic_tab.get(stale_key)
                rec["traffic_source"] = stale_src if rec["traffic"] else None
                if pmc and "error" in pmc:
                    rec["traffic_error"] = pmc["error"]
            return rec

        short = {"densify_pairs": "k_densify_pairs", "partition_dual": "k_partition_dual", "sub_sort": "k_vsort_pipe",
                 "leaf_rows": "k_vrows", "expand_mid2": "k_expand_mid2"}
        for name, rec in recs.items():
            with_traffic(rec, short.get(name, name), f"{args.workload}/{name}/n{world}")
        if match is not None and "roofline" in match:
            with_traffic(match["roofline"], "k_mat_mid2", f"{args.workload}/mat_mid2/n{world}")
        # the bench line's roofline: the kernel with the largest total time over both timed regions
        top = roof
        if match is not None and "roofline" in match:
            t_a = prof.get(dom, (0, 0.0))[1] if dom else 0.0
            t_b = mat["prof"].get("mat_mid2", (0, 0.0))[1]
            if t_b >= t_a:
                top = match["roofline"]
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
            "config": {"workload": f"LDBC SNB {args.workload.upper()} Person-KNOWS*1..2-Person, all persons as sources: CSR build (no edge-rowid "
                                   "payload) + counting 2-hop expansion (row count + 32-bit checksum per walk: a VALU checksum, "
                                   "not an HBM figure — the HBM-shaped form of the same MATCH, rows materialised, is "
                                   "`match_materialised`, which `roofline` describes)",
                       "vertices": int(V), "knows_rows": int(R), "rows_1hop": int(rows1), "rows_2hop": int(rows2),
                       "traversed_edges": int(te_total),
                       "collective_backend": dist.get_backend() if dist is not None else None, "parallelism": f"vertex-ownership shards x{world} (edge table hash-partitioned by endpoint owner, vertex table replicated, CSR + expansion sharded, no data-path collective)"},
            "roofline": top,
            "match_materialised": match,
            "roofline_count_step": roof,
            "roofline_kernels": recs,
            "roofline_phases": phases,
            "kernels": kernels,
            "staging_ms_pcie": t_stage * 1e3,
        }
        line.update(extra)
        if "cpu_baseline" not in line:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if gg.ctx:
        gg.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
