#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run7; mkdir -p $OUT
cd $R
timeout 1200 python tools/exp_graph.py > $OUT/graph.log 2>&1
cat $OUT/graph.log
bash tools/r05_ab_pyr.sh r05_run7/ab r04 X=1 prio0 occ74 > /dev/null 2>&1
cat $OUT/ab/ab.log | sed 's/kps \[[^]]*\]//'
