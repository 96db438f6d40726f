#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run16; mkdir -p $OUT
cd $R
( echo "# final round-5 library (graph replay in the pipeline's workers, round-5 kernels)"
  echo "# tools/stress_pipeline.py 30: the benchmark pipeline (6 GPU workers x 8 pairs + 8 verify workers, 8-bit host input) 30 x 48 pairs against the oracle chain"
  timeout 900 python tools/stress_pipeline.py 30 2>&1 | grep -v amdgpu.ids | tail -2
  echo "# tools/stress_match.py 6 1500: six contexts repeat detect + describe + match of three 1080p pairs, every result compared with its first"
  timeout 600 python tools/stress_match.py 6 1500 2>&1 | grep -v amdgpu.ids | tail -2 ) > $OUT/concurrency_stress.log 2>&1
cat $OUT/concurrency_stress.log
( echo "# MODS_BENCH_SHARE_GPU=1: N ranks of bench.py on ONE MI355X over gloo (tools/run_share_gpu.sh): the multi-rank control flow and the host side of a rank; the aggregate is bounded by the one GPU"
  for n in 2 4 8; do
    w=6; [ $n = 8 ] && w=3; [ $n = 4 ] && w=4
    echo "# $n ranks on the box's 16 usable cores, $w GPU workers per rank:"
    bash tools/run_share_gpu.sh r05_run16/share $n 0-255 --pairs-per-step 96 --gpu-workers $w 2>&1 | tail -1
  done ) > $OUT/share_gpu_ranks.log 2>&1
cat $OUT/share_gpu_ranks.log | cut -c1-400
bash tools/run_configs.sh r05_run16 > /dev/null 2>&1
cut -c1-500 $OUT/configs_c3_c4_c5.log
