#!/bin/bash
ulimit -c 0
cd $GRAFT_REPO_ROOT
for q in 4 16 32 4 16 32 4 16 24; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-match-leg --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q', d['value'], d['host_cpu']['process_cpu_ms_per_pair'], d['latency_ms_single_pair']['one_call_hbm_f32']['median'], d['roofline_pyramid']['one_scope']['ms'])"
done
