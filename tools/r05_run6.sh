#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run6; mkdir -p $OUT
cd $R
(timeout 900 python -m pytest tests/test_gpu_describe.py tests/test_gpu_pair.py -x -q -m gpu -k "graph_replay or pipeline or bench or batch16" 2>&1 | tail -8) > $OUT/gputest.log
cat $OUT/gputest.log
STAGES=pyramid bash tools/trace_pyr.sh > $OUT/trace.log 2>&1
cat $OUT/trace.log
