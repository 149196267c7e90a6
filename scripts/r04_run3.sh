#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for p in 1 2 3; do echo "process $p"; ./build/ubench_fill_candidates 8 || exit 1; done > $O/r4_fill_candidates.txt 2>&1
cat $O/r4_fill_candidates.txt | grep -v cand
for n in 2 4 8; do
  python3 bench.py --shard-of $n --no-cpu --no-extras --no-pmc > $O/r4_shard_of_$n.json 2> $O/r4_shard_of_$n.log || { tail -5 $O/r4_shard_of_$n.log; exit 1; }
  tail -2 $O/r4_shard_of_$n.log
done
