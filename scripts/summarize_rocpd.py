#!/usr/bin/env python3
"""rocprofv3's sqlite (rocpd) output -> the summaries kept under profiles/.
   summarize_rocpd.py <tag>   expects gpurun_out/prof_<tag>/*.db (--kernel-trace --stats) and, optionally,
                              gpurun_out/pmc_fetch_<tag>/*.db, gpurun_out/pmc_write_<tag>/*.db (--pmc X)
Writes profiles/r01_<tag>_sf100_kernel_stats.csv, profiles/r01_<tag>_sf100_pmc_summary.txt and refreshes
profiles/pmc_traffic.json (HBM bytes per launch, gfx950 correction of MI355X_MICROARCH.md applied)."""
import collections
import glob
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, "profiles")


def short(name):
    """k_<kernel> out of a demangled (gg::k_x<...>(...)) or mangled (_ZN2gg<len>k_x...) symbol"""
    m = re.search(r"(\d+)k_", name)
    if name.startswith("_Z") and m:
        return name[m.end() - 2:m.end() - 2 + int(m.group(1))]
    n = name.split("(")[0]
    return n.split("::")[-1].split("<")[0]


def dispatches(db):
    con = sqlite3.connect(db)
    q = ("select s.kernel_name, d.start, d.end, d.event_id from rocpd_kernel_dispatch d "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid")
    return con, list(con.execute(q))


dbs = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "*.db"))
if dbs:
    _, rows = dispatches(dbs[0])
    agg = collections.defaultdict(list)
    for name, start, end, _ in rows:
        agg[name.split("(")[0]].append(end - start)
    total = sum(sum(v) for v in agg.values())
    with open(os.path.join(out, f"r01_{tag}_sf100_kernel_stats.csv"), "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (name, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total,
                                                      min(v), max(v)))
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 --no-cpu (SF100, 1 MI355X)",
         "# raw counter values are KB per dispatch (avg over dispatches); gfx950: FETCH_SIZE under-reports wide coalesced streams by 2x (MI355X_MICROARCH.md HBM section)"]
res = {}
for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    dbs = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{sub}_{tag}", "*.db"))
    if not dbs:
        continue
    con, rows = dispatches(dbs[0])
    by_event = {ev: name for name, _, _, ev in rows}
    d = collections.defaultdict(list)
    q = ("select e.event_id, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id and e.guid = p.guid "
         "where p.name = ?")
    per_event = collections.defaultdict(float)
    for ev, value in con.execute(q, (cname,)):
        per_event[ev] += value  # one row per counter instance (XCD): add them up
    for ev, value in per_event.items():
        if ev in by_event:
            d[by_event[ev].split("(")[0][:70]].append(value)
    lines.append(cname)
    for k, v in d.items():
        lines.append("  %-70s n=%3d avg=%14.1f KB" % (k, len(v), sum(v) / len(v)))
        res.setdefault(k, {})[cname] = sum(v) / len(v)
if len(lines) > 2:
    open(os.path.join(out, f"r01_{tag}_sf100_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
    traffic = {"_note": "HBM-side bytes per launch from rocprofv3 PMC (separate FETCH_SIZE / WRITE_SIZE passes, "
               f"profiles/r01_{tag}_sf100_pmc_summary.txt): (2*FETCH_SIZE + WRITE_SIZE) * 1024; the factor 2 on FETCH_SIZE is the "
               "gfx950 correction of MI355X_MICROARCH.md (upper bound for our 4-byte-per-lane coalesced loads)"}
    names = {"k_expand_mid2": "expand_mid2", "k_densify_hist": "densify_hist", "k_radix_scatter": "radix_scatter"}
    acc = collections.defaultdict(list)
    for k, v in res.items():
        n = short(k)
        if n in names and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            acc[names[n]].append((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    for n, vals in acc.items():
        traffic[f"sf100/{n}/n1"] = int(sum(vals) / len(vals))
    json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
