#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for p in 1 2; do echo "process $p"; ./build/ubench_fill_map 208 || exit 1; done > $O/r4_fill_map.txt 2>&1
timeout -k 10 900 python3 -m pytest tests/test_duckdb_extension.py tests/test_reference_callouts.py -x -q -m gpu > $O/r4_ext1.log 2>&1; echo "ext rc=$?"; tail -8 $O/r4_ext1.log
