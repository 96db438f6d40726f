#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run18; mkdir -p $OUT
cd $R
for g in 1 0 1 0 1 0; do
  MODS_GRAPHS=$g timeout 300 python bench.py --no-cpu-baseline --no-match-leg 2>&1 | tail -1 > $OUT/bench_g$g.json
  python3 - $OUT/bench_g$g.json $g <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print("graphs", sys.argv[2], d['value'], "replayed", d['config']['detect_describe_batches_replayed_as_graph'], "cpu/pair", d['host_cpu']['process_cpu_ms_per_pair'], {k:v['total'] for k,v in d['host_cpu']['by_thread_name_ms_per_pair'].items()}, "frac_in_pipeline", d['roofline']['frac_in_pipeline'])
PY
done
