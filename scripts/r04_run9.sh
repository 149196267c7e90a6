#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -m pytest tests/test_gpu_parity.py -x -q -k "vertices_from_edges" > $O/r4_t9.log 2>&1; rc=$?; tail -5 $O/r4_t9.log
[ $rc -ne 0 ] && exit $rc
python3 scripts/bench_edge_only.py sf100 10 > $O/r4_edge_only.json 2> $O/r4_edge_only.log || { tail $O/r4_edge_only.log; exit 1; }
cat $O/r4_edge_only.json
python3 scripts/bench_edge_only.py sf10 10 > $O/r4_edge_only_sf10.json 2>> $O/r4_edge_only.log; cat $O/r4_edge_only_sf10.json
