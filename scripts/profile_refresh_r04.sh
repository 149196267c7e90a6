#!/bin/bash
# Round 4: every figure the bench line and DESIGN.md quote, reproduced in one GPU call.  On the box:
#   bash scripts/profile_refresh_r04.sh            then, back in the repo:  python3 scripts/summarize_r04.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (1) kernel trace of the default bench (both timed regions + the secondary configs), counter passes left out here
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_trace -- python3 $R/bench.py --no-pmc > $O/r04_trace.json 2> $O/r04_trace.log || { tail -5 $O/r04_trace.log; exit 1; }
# (2) the counter passes the bench makes itself, kept as files: one counter per pass over `bench.py --pmc-child`
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/r04_pmc_$c -- python3 $R/bench.py --pmc-child > $O/r04_pmc_$c.log 2>&1 || { tail -5 $O/r04_pmc_$c.log; exit 1; }
done
# (3) the edge-only build (the reference's own 2-hop idiom): kernel trace + its own line
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_trace_edge_only -- python3 $R/scripts/bench_edge_only.py sf100 10 > $O/r04_edge_only.json 2> $O/r04_edge_only.log || { tail -5 $O/r04_edge_only.log; exit 1; }
cd $R
# (4) the driver's command
python3 bench.py > $O/r04_bench_sf100_default.json 2> $O/r04_bench_sf100_default.log || { tail -5 $O/r04_bench_sf100_default.log; exit 1; }
# (5) one rank's share of an N-rank run, both regions (NOT a scaling curve: one GPU)
for n in 2 4 8; do
  python3 bench.py --shard-of $n --no-cpu --no-extras --no-pmc > $O/r04_shard_of_$n.json 2> $O/r04_shard_of_$n.log || { tail -5 $O/r04_shard_of_$n.log; exit 1; }
done
# (6) whole SQL statements inside the reference (PCIe and table scans included)
python3 scripts/bench_sql.py --scale sf100 --skip-cpu --gpu-runs 5 > $O/r04_sql_sf100.json 2> $O/r04_sql_sf100.log || { tail -5 $O/r04_sql_sf100.log; exit 1; }
find $O/r04_trace $O/r04_trace_edge_only $O/r04_pmc_WRITE_SIZE $O/r04_pmc_FETCH_SIZE -name '*kernel_trace.csv' -size +4M -delete 2>/dev/null
echo "profile_refresh_r04 done"
