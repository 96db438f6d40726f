#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run17; mkdir -p $OUT
cd $R
( echo "# MODS_BENCH_SHARE_GPU=1: N ranks of bench.py on ONE MI355X over gloo (tools/run_share_gpu.sh): the multi-rank control flow and the host side of a rank"
  echo "# on the box's 16 usable cores (cgroup quota); the aggregate is bounded by the one GPU (a single rank with 6 GPU workers: 836-909 pairs/s) - what the"
  echo "# lines show is whether N ranks' host threads fit the cores: GPU workers per rank chosen so that the GPU sees ~6-8 worker contexts in all"
  for n in 2 4 8; do
    w=3; [ $n = 8 ] && w=1; [ $n = 4 ] && w=2
    echo "# $n ranks, $w GPU worker(s) per rank:"
    bash tools/run_share_gpu.sh r05_run17/share$n $n 0-255 --pairs-per-step 96 --gpu-workers $w 2>&1 | tail -1
    grep -a "out of memory\|Error" gpurun_out/r05_run17/share$n/err.log | head -2
  done ) > $OUT/share_gpu_ranks.log 2>&1
cat $OUT/share_gpu_ranks.log | cut -c1-600
(timeout 900 python -m pytest tests/test_gpu_describe.py -q -m gpu -k "unaligned" 2>&1 | tail -4)
