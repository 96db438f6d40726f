#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run21; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PYR_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg -- python $R/tools/prof_detect.py 16 > $OUT/detect_leg.log 2>&1
cp $(find $OUT/leg -name "*kernel_stats.csv" | head -1) $OUT/detect_leg_kernel_stats.csv; rm -rf $OUT/leg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg -- python $R/tools/prof_detect.py 16 > $OUT/detect_leg_two_streams.log 2>&1
cp $(find $OUT/leg -name "*kernel_stats.csv" | head -1) $OUT/detect_leg_two_streams_kernel_stats.csv; rm -rf $OUT/leg
grep -a "^pyramid\|^blur\|^nms\|^batch" $OUT/detect_leg.log $OUT/detect_leg_two_streams.log
