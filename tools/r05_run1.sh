#!/bin/bash
# round 5, first GPU call: parity of the cleaned-up library, blur variants, the foreign-MFMA experiment, the new bench line
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run1; mkdir -p $OUT
cd $R
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $OUT/gputest.log
bash tools/run_blur_variants.sh r05_run1/blur X=1 base swz colj75 colj86 colj11 > /dev/null 2>&1
cd $R
(timeout 300 python tools/exp_foreign_mfma.py 5 2>&1 | tail -20) > $OUT/foreign_mfma.log
(timeout 600 python bench.py 2>&1 | tail -3) > $OUT/bench.log
cat $OUT/gputest.log $OUT/blur/variants.log $OUT/foreign_mfma.log; tail -c 6000 $OUT/bench.log
