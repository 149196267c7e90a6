#!/bin/bash
# Kernel trace + the two HBM counter passes of the default bench (run on the GPU box): profile_refresh.sh <tag>
# then, back in the repo: python scripts/summarize_profiles.py <tag> <round>
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-extras --steps 20 > $R/gpurun_out/prof_$tag.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$tag -- python3 $R/bench.py --no-extras --no-cpu --steps 2 --warmup 1 > $R/gpurun_out/pmc_fetch_$tag.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$tag -- python3 $R/bench.py --no-extras --no-cpu --steps 2 --warmup 1 > $R/gpurun_out/pmc_write_$tag.log 2>&1
echo "profile_refresh rc=$?"
