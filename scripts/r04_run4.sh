#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for b in 16 24 40 64; do
  python3 bench.py --no-cpu --no-extras --no-pmc --steps 5 --mat-budget-gb $b > $O/r4_budget_$b.json 2> $O/r4_budget_$b.log || { tail -5 $O/r4_budget_$b.log; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$O/r4_budget_$b.json").read().strip().splitlines()[-1]); m=d["match_materialised"]
print("budget $b:", round(m["ms_per_step"],2), "ms/step, parts", m["parts_per_step_this_rank"], "mat_mid2 frac", round(m["roofline"]["frac"],3), "TB/s wall", round(m["bytes_written_per_s"]/1e12,2), "parity", m["parity"], "count step", round(d["ms_per_step"],3))
PY
done
python3 -m pytest tests/test_gpu_parity.py -x -q -k "materialised or mid_ranges or digest" 2>&1 | tail -3
