#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run11; mkdir -p $OUT
cd $R
(timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -25) > $OUT/gputest.log
cat $OUT/gputest.log
bash tools/r05_run10.sh
