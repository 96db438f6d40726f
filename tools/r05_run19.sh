#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run19; mkdir -p $OUT
cd $R
for cfg in "6 8 8" "8 8 8" "5 8 8" "6 6 8" "6 12 8" "4 16 8" "6 8 12" "7 8 10" "6 8 8"; do
  set -- $cfg
  v=$(timeout 300 python bench.py --no-cpu-baseline --no-match-leg --gpu-workers $1 --pairs-per-batch $2 --verify-workers $3 --steps 10 --warmup 2 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_cpu']['process_cpu_ms_per_pair'], d['config']['detect_describe_batches_replayed_as_graph'])")
  echo "workers $1 ppb $2 verify $3 -> $v" | tee -a $OUT/sweep.log
done
