#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel: pmc_summary.py <dir> [<dir> ...] (csv output directories)"""
import collections
import csv
import glob
import os
import sys

for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("#", f)
        for k, cs in acc.items():
            print("  %-60s n=%d " % (k, len(next(iter(cs.values())))) + " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
