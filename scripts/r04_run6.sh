#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
export GG_PLACE_TRACE=1
for pr in 1 6; do for b in 16 40; do
  GG_PLACE_PROBES=$pr python3 bench.py --no-cpu --no-extras --no-pmc --steps 5 --mat-budget-gb $b > $O/r4_place_${pr}_$b.json 2> $O/r4_place_${pr}_$b.log || { tail -5 $O/r4_place_${pr}_$b.log; exit 1; }
  grep "placed" $O/r4_place_${pr}_$b.log
  python3 - <<PY
import json
d=json.loads(open("$O/r4_place_${pr}_$b.json").read().strip().splitlines()[-1]); m=d["match_materialised"]
print("probes $pr budget $b:", round(m["ms_per_step"],2), "ms/step, parts", m["parts_per_step_this_rank"], "mat_mid2 frac", round(m["roofline"]["frac"],3), "TB/s wall", round(m["bytes_written_per_s"]/1e12,2), "parity", m["parity"])
PY
done; done
