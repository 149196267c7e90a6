#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -m pytest tests/test_gpu_parity.py -x -q -k "walk_counts or shard or mid_ranges or repeatable or materialised" > $O/r4_t2.log 2>&1; rc=$?; tail -5 $O/r4_t2.log
[ $rc -ne 0 ] && exit $rc
python3 bench.py > $O/r4_bench1.json 2> $O/r4_bench1.log; rc=$?; tail -12 $O/r4_bench1.log
[ $rc -ne 0 ] && exit $rc
for n in 2 4 8; do
  python3 bench.py --shard-of $n --no-cpu --no-extras --no-pmc > $O/r4_shard_of_$n.json 2> $O/r4_shard_of_$n.log || { tail -5 $O/r4_shard_of_$n.log; exit 1; }
  tail -2 $O/r4_shard_of_$n.log
done
