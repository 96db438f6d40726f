#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run15; mkdir -p $OUT
cd $R
(timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8) > $OUT/gputest.log
cat $OUT/gputest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dstats -- python $R/tools/prof_describe.py > $OUT/describe_leg.log 2>&1
cp $(find $OUT/dstats -name "*kernel_stats.csv" | head -1) $OUT/describe_leg_kernel_stats.csv; rm -rf $OUT/dstats
head -8 $OUT/describe_leg_kernel_stats.csv | cut -c1-120
