#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
export GG_PLACE_TRACE=1
for b in 16 40; do
  python3 bench.py --no-cpu --no-extras --no-pmc --steps 5 --mat-budget-gb $b > $O/r4_sets_$b.json 2> $O/r4_sets_$b.log || { tail -5 $O/r4_sets_$b.log; exit 1; }
  grep "column set" $O/r4_sets_$b.log | cut -c1-400
  python3 - <<PY
import json
d=json.loads(open("$O/r4_sets_$b.json").read().strip().splitlines()[-1]); m=d["match_materialised"]
print("budget $b:", round(m["ms_per_step"],2), "ms/step, parts", m["parts_per_step_this_rank"], "mat_mid2 frac", round(m["roofline"]["frac"],3), "TB/s wall", round(m["bytes_written_per_s"]/1e12,2), "parity", m["parity"])
PY
done
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "materialised or mid_ranges or digest or sf10 or SF10" 2>&1 | tail -3
