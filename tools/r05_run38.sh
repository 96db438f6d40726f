#!/bin/bash
# auxiliary profiles of the final library: N ranks on one GPU, the other configurations, the 0.4-inlier-ratio line, concurrency stress
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run38; mkdir -p $OUT
cd $R
( echo "# MODS_BENCH_SHARE_GPU=1: N ranks of bench.py on ONE MI355X over gloo (tools/run_share_gpu.sh): the multi-rank control flow and the host side of a rank"
  echo "# on the box's 16 usable cores (cgroup quota); the aggregate is bounded by the one GPU (a single rank with 6 GPU workers: 881-906 pairs/s) - what the"
  echo "# lines show is whether N ranks' host threads fit the cores: GPU workers per rank chosen so that the GPU sees ~6-8 worker contexts in all"
  echo "# columns: aggregate pairs/s | per rank | host | shape | process CPU ms per pair (of one rank)"
  for n in 2 4 8; do
    w=3; [ $n = 8 ] && w=1; [ $n = 4 ] && w=2
    echo "# $n ranks, $w GPU worker(s) per rank:"
    bash tools/run_share_gpu.sh r05_run38/share$n $n 0-255 --pairs-per-step 96 --gpu-workers $w 2>&1 | tail -1
    grep -a "out of memory\|Error" gpurun_out/r05_run38/share$n/err.log | head -2
  done ) > $OUT/share_gpu_ranks.log 2>&1
cat $OUT/share_gpu_ranks.log | cut -c1-500
bash tools/run_configs.sh r05_run38 | cut -c1-200
timeout 500 python bench.py --inlier-ratio 0.4 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $OUT/bench_inlier_ratio_0.4.json; cut -c1-160 $OUT/bench_inlier_ratio_0.4.json
