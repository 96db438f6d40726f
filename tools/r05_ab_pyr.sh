#!/bin/bash
# round 5: the whole scale space as one scope (MODS_STAGE_PYRAMID, both streams) for library variants on ONE box, alternating:
#   tools/r05_ab_pyr.sh <out> <variant|X=1> ...     (X=1 = the shipped library)
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd $R
for rep in 1 2 3; do
  for v in "$@"; do
    if [[ "$v" == *=* ]]; then L=""; else L=$R/mods-light-zmq_amd/_variants/libmodsgpu_$v.so; fi
    a=$(MODS_LIB=$L STAGES=pyramid timeout 120 python tools/prof_detect.py 16 2>&1 | grep -a "^pyramid\|^batch" | tr '\n' ' ')
    b=$(MODS_LIB=$L STAGES=blur,blur_small,resize,nms timeout 120 python tools/prof_detect.py 16 2>&1 | grep -a "^blur \|^nms\|^resize\|^blur_small" | awk '{printf "%s %s  ", $1, $2}')
    echo "rep $rep $v: $a | $b" >> $OUT/ab.log
  done
done
cat $OUT/ab.log
