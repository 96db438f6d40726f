#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run13; mkdir -p $OUT
cd $R
(MODS_LIB=$R/mods-light-zmq_amd/_variants/libmodsgpu_nmspf1.so timeout 600 python -m pytest tests/test_gpu_detect.py -x -q -m gpu 2>&1 | tail -4) > $OUT/gputest_nmspf1.log
cat $OUT/gputest_nmspf1.log
bash tools/r05_ab_pyr.sh r05_run13/ab X=1 nmspf1 > /dev/null 2>&1
cat $OUT/ab/ab.log | sed 's/kps \[[^]]*\]//'
