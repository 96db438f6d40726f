#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run10; mkdir -p $OUT
cd $R
for m in flag sleep flag sleep; do
  MODS_SYNC=$m timeout 300 python bench.py --no-cpu-baseline --no-match-leg 2>&1 | tail -1 > $OUT/bench_$m.json
  python3 - $OUT/bench_$m.json $m <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print("sync", sys.argv[2], d['value'], d['host_cpu']['process_cpu_ms_per_pair'], d['host_cpu']['by_thread_name_ms_per_pair'], d['roofline_pyramid']['one_scope']['ms'], d['latency_ms_single_pair'])
PY
done
STAGES=pyramid bash tools/trace_pyr.sh > $OUT/trace.log 2>&1
cat $OUT/trace.log | tail -45
cd $R
(timeout 300 python bench.py --config c3 --ladder hessian 2>&1 | tail -1) > $OUT/c3_hessian.json; cut -c1-900 $OUT/c3_hessian.json
(timeout 300 python bench.py --config c3 2>&1 | tail -1) > $OUT/c3_full.json; cut -c1-900 $OUT/c3_full.json
(timeout 300 python bench.py --inlier-ratio 0.4 --no-cpu-baseline --no-match-leg 2>&1 | tail -1) > $OUT/bench_inlier_ratio_0.4.json; cut -c1-300 $OUT/bench_inlier_ratio_0.4.json
