#!/bin/bash
ulimit -c 0
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_run41
for shape in "6 8 8" "4 8 8" "8 8 8" "6 12 8" "6 16 8" "8 4 8" "10 8 8" "6 8 4" "6 8 8"; do
  set -- $shape
  timeout 300 python bench.py --no-cpu-baseline --no-match-leg --steps 6 --gpu-workers $1 --pairs-per-batch $2 --verify-workers $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$shape', d['value'], d['host_cpu']['process_cpu_ms_per_pair'], d['latency_ms_single_pair']['pipeline_one_in_flight_host_u8']['median'])"
done
