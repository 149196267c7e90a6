#!/usr/bin/env python3
"""Copy the judged summaries out of gpurun_out/ into profiles/ (tracked):
   summarize_profiles.py <tag> [round]   expects gpurun_out/prof_<tag>, pmc_fetch_<tag>, pmc_write_<tag> (rocprofv3 csv output)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
out = os.path.join(ROOT, "profiles")
ks = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "*", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(out, f"{rnd}_{tag}_sf100_kernel_stats.csv"))
res = {}
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 --no-cpu (SF100, 1 MI355X)",
         "# raw counter values are KB per dispatch (avg over dispatches); gfx950: FETCH_SIZE under-reports wide coalesced streams by 2x (MI355X_MICROARCH.md HBM section)"]
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{'fetch' if name == 'FETCH_SIZE' else 'write'}_{tag}", "*", "*counter_collection.csv"))
    if not fs:
        continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        d[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
    lines.append(name)
    for k, v in d.items():
        lines.append("  %-70s n=%3d avg=%14.1f KB" % (k, len(v), sum(v) / len(v)))
        res.setdefault(k, {})[name] = sum(v) / len(v)
open(os.path.join(out, f"{rnd}_{tag}_sf100_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
traffic = {"_source": f"profiles/{rnd}_{tag}_sf100_pmc_summary.txt", "_note": "HBM-side bytes per launch from rocprofv3 PMC (separate FETCH_SIZE / WRITE_SIZE passes, "
           f"profiles/{rnd}_{tag}_sf100_pmc_summary.txt): (2*FETCH_SIZE + WRITE_SIZE) * 1024; the factor 2 on FETCH_SIZE is the "
           "gfx950 correction of MI355X_MICROARCH.md (upper bound for our 4-byte-per-lane coalesced loads)"}
short = {"k_expand_mid2": "expand_mid2", "k_densify_hist": "densify_hist", "k_radix_scatter": "radix_scatter",
         "k_densify_pairs": "densify_pairs", "k_partition_dual": "partition_dual", "k_sub_sort": "sub_sort", "k_sub_sort_pipe": "sub_sort", "k_leaf_rows": "leaf_rows",
         "k_vsort_pipe": "sub_sort", "k_vrows": "leaf_rows"}
acc = collections.defaultdict(list)
for k, v in res.items():
    n = k.split("::")[-1].split("<")[0]
    if n in short and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        acc[short[n]].append((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
for n, vals in acc.items():  # template instantiations of one kernel name: plain mean per launch
    traffic[f"sf100/{n}/n1"] = int(sum(vals) / len(vals))
# secondary configs: BFS bytes per 64-source BATCH (all bfs kernels of the profiled batches, warm-up batch included in
# both numerator and denominator), materialised rows per mat_mid2 launch
def counter_rows(kind, name):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{kind}_{name}_{tag}", "*", "*counter_collection.csv"))
    return list(csv.DictReader(open(fs[0]))) if fs else []
extra_lines = []
for kind, key, match, per in (("bfs", "sf100/bfs64_batch/n1", "k_bfs", 4 + 1 + 4), ("mat", "sf10/mat_mid2/n1", "k_mat_mid2", None)):
    tot = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = [r for r in counter_rows(kind, cname) if match in r["Kernel_Name"]]
        if not rows:
            tot = {}
            break
        n = per if per else len(rows)
        tot[cname] = sum(float(r["Counter_Value"]) for r in rows) / n
        extra_lines.append("%s %-40s dispatches=%5d per-unit=%14.1f KB" % (cname, key, len(rows), tot[cname]))
    if tot:
        traffic[key] = int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024)
if extra_lines:
    open(os.path.join(out, f"{rnd}_{tag}_sf100_pmc_summary.txt"), "a").write(
        "# secondary configs (bench_bfs.py --no-cpu --batches 4: warm-up + 4 timed + 4 profiled batches; scripts/bench_materialise.py sf10: 3 launches)\n" + "\n".join(extra_lines) + "\n")
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
