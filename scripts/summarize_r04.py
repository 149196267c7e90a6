#!/usr/bin/env python3
"""Copy what scripts/profile_refresh_r04.sh left in gpurun_out/ into profiles/ (tracked)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def short(full):
    name = full.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0].split("<")[0].split("::")[-1].strip()


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


for src, dst in (("r04_trace", "r04_sf100_kernel_stats.csv"), ("r04_trace_edge_only", "r04_edge_only_sf100_kernel_stats.csv")):
    ks = glob.glob(os.path.join(G, src, "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(P, dst))
lines = ["# rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes) over `python3 bench.py --pmc-child` (SF100, 1 MI355X):",
         "# two steps of region A (build + counting expansion) and of region B (build + every 2-hop row written, 8 parts).",
         "# per kernel: launches, average KiB per launch as the counter reports it; bytes = KiB x 1024, FETCH_SIZE x 2",
         "# (profiles/r04_counter_calibration.txt)"]
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(G, f"r04_pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    lines.append(c)
    for k, v in sorted(acc.items()):
        lines.append("  %-28s n=%4d avg=%16.1f KiB  sum=%18.1f KiB" % (k, len(v), sum(v) / len(v), sum(v)))
open(os.path.join(P, "r04_sf100_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
for name in ("r04_bench_sf100_default.json", "r04_edge_only.json", "r04_sql_sf100.json"):
    p = os.path.join(G, name)
    if os.path.exists(p):
        json.dump(last_json(p), open(os.path.join(P, name if name != "r04_edge_only.json" else "r04_edge_only_sf100.json"), "w"), indent=1)
p = os.path.join(G, "r04_trace.json")
if os.path.exists(p):
    json.dump(last_json(p), open(os.path.join(P, "r04_bench_sf100_under_rocprof.json"), "w"), indent=1)
shards = {"note": "rank 0's share of an N-rank run timed on ONE GPU (bench.py --shard-of N): NOT a scaling curve — only one GPU "
                  "was available to this build; N = 1 from profiles/r04_bench_sf100_default.json"}
base = os.path.join(P, "r04_bench_sf100_default.json")
if os.path.exists(base):
    b = json.load(open(base))
    shards["1"] = {"count_step_ms": b["ms_per_step"], "materialised_step_ms": b["match_materialised"]["ms_per_step"]}
for n in (2, 4, 8):
    p = os.path.join(G, f"r04_shard_of_{n}.json")
    if os.path.exists(p):
        d = last_json(p)
        shards[str(n)] = {"count_step_ms": d["ms_per_step"], "materialised_step_ms": d["match_materialised"]["ms_per_step"],
                          "rows_this_rank": d["match_materialised"]["rows_this_rank"], "parts": d["match_materialised"]["parts"],
                          "mat_mid2_frac_of_hbm": d["match_materialised"]["roofline_frac"],
                          "kernels_us_per_step": d["match_materialised"]["kernels_us_per_step"]}
json.dump(shards, open(os.path.join(P, "r04_shard_of_mat.json"), "w"), indent=1)
print("profiles/ updated:", sorted(f for f in os.listdir(P) if f.startswith("r04_")))
