#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for p in 1 2; do echo "process $p"; ./build/ubench_fill_candidates 6 || exit 1; done > $O/r4_fill_candidates2.txt 2>&1
grep -v cand $O/r4_fill_candidates2.txt
for g in 1 16 256; do for b in 16 40; do
  GG_MAT_GROUPS=$g python3 bench.py --no-cpu --no-extras --no-pmc --steps 5 --mat-budget-gb $b > $O/r4_groups_${g}_$b.json 2> $O/r4_groups_${g}_$b.log || { tail -5 $O/r4_groups_${g}_$b.log; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$O/r4_groups_${g}_$b.json").read().strip().splitlines()[-1]); m=d["match_materialised"]
print("groups $g budget $b:", round(m["ms_per_step"],2), "ms/step, parts", m["parts_per_step_this_rank"], "mat_mid2 frac", round(m["roofline"]["frac"],3), "TB/s wall", round(m["bytes_written_per_s"]/1e12,2), "parity", m["parity"])
PY
done; done
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/r4_gpu_suite1.log 2>&1; echo "suite rc=$?"; tail -5 $O/r4_gpu_suite1.log
