#!/bin/bash
# round 5, second GPU call: parity (all -m gpu tests), describe leg kernel statistics, the foreign-MFMA experiment, the bench line
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run2; mkdir -p $OUT
cd $R
(timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $OUT/gputest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dstats -- python $R/tools/prof_describe.py > $OUT/describe_leg.log 2>&1
cp $(find $OUT/dstats -name "*kernel_stats.csv" | head -1) $OUT/describe_leg_kernel_stats.csv; rm -rf $OUT/dstats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg -- python $R/tools/prof_detect.py 16 > $OUT/detect_leg.log 2>&1
cp $(find $OUT/leg -name "*kernel_stats.csv" | head -1) $OUT/detect_leg_kernel_stats.csv; rm -rf $OUT/leg
cd $R
(timeout 400 python tools/exp_foreign_mfma.py 5 2>&1 | tail -20) > $OUT/foreign_mfma.log
(timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1) > $OUT/bench.json
cat $OUT/gputest.log $OUT/foreign_mfma.log; head -12 $OUT/describe_leg_kernel_stats.csv | cut -c1-150; cut -c1-400 $OUT/bench.json
