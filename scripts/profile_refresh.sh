#!/bin/bash
# Kernel trace + the two HBM counter passes of the default bench (run on the GPU box): profile_refresh.sh <tag>
# then, back in the repo: python scripts/summarize_profiles.py <tag> <round>
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-extras --steps 20 > $R/gpurun_out/prof_$tag.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$tag -- python3 $R/bench.py --no-extras --no-cpu --steps 2 --warmup 1 > $R/gpurun_out/pmc_fetch_$tag.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$tag -- python3 $R/bench.py --no-extras --no-cpu --steps 2 --warmup 1 > $R/gpurun_out/pmc_write_$tag.log 2>&1 && \
for c in FETCH_SIZE WRITE_SIZE; do  # the secondary configs: 64-source BFS batches (SF100), materialised 2-hop rows (SF10)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bfs_${c}_$tag -- python3 $R/bench_bfs.py --no-cpu --batches 4 > $R/gpurun_out/pmc_bfs_${c}_$tag.log 2>&1 && \
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mat_${c}_$tag -- python3 $R/scripts/bench_materialise.py sf10 > $R/gpurun_out/pmc_mat_${c}_$tag.log 2>&1 || break
done
echo "profile_refresh rc=$?"
