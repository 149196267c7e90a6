#!/bin/bash
# rocprofv3 counter passes over scripts/build_once.py (one --pmc group per run): pmc_passes.sh <tag> <lib|-> <group>...
# a group is a comma-separated counter list
tag=$1; lib=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i
  rocprofv3 --pmc ${grp//,/ } --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/scripts/build_once.py sf100 $lib 2 > $out.log 2>&1 || { echo "pass $i failed"; tail -5 $out.log; exit 1; }
done
cd $GRAFT_REPO_ROOT && python3 scripts/pmc_summary.py $(for j in $(seq 1 $i); do echo gpurun_out/pmc_${tag}_$j; done) > gpurun_out/pmc_${tag}.txt
