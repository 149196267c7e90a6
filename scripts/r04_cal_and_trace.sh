#!/bin/bash
# round 4, first GPU call: (1) counter calibration on known byte counts, (2) kernel trace of the default bench
# including the SF100 materialised parts, (3) WRITE_SIZE / FETCH_SIZE passes of the same command
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
$R/build/ubench_write_cal > $O/r4_cal_plain.txt 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r4_cal_write -- $R/build/ubench_write_cal > $O/r4_cal_write.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r4_cal_fetch -- $R/build/ubench_write_cal > $O/r4_cal_fetch.log 2>&1 && \
(cd $R && python3 scripts/pmc_summary.py gpurun_out/r4_cal_write gpurun_out/r4_cal_fetch > $O/r4_cal_summary.txt) && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_trace0 -- python3 $R/bench.py --no-cpu --steps 5 > $O/r4_trace0.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r4_pmcw0 -- python3 $R/bench.py --no-cpu --steps 2 --warmup 1 > $O/r4_pmcw0.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r4_pmcf0 -- python3 $R/bench.py --no-cpu --steps 2 --warmup 1 > $O/r4_pmcf0.log 2>&1
rc=$?
cd $R && python3 scripts/pmc_summary.py gpurun_out/r4_pmcw0 gpurun_out/r4_pmcf0 > $O/r4_pmc0_summary.txt 2>&1
echo "rc=$rc"; cat $O/r4_cal_summary.txt
# the raw counter CSVs are large; keep the summaries and the stats only
find $O/r4_pmcw0 $O/r4_pmcf0 $O/r4_cal_write $O/r4_cal_fetch -name '*.csv' -size +2M -delete 2>/dev/null
find $O/r4_trace0 -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null
exit $rc
