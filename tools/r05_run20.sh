#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run20; mkdir -p $OUT
cd $R
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-match-leg 2>&1 | tail -1 > $OUT/bench$i.json
python3 - $OUT/bench$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
p=d['roofline_pyramid']
print(d['value'], p['stage_ms'], p['one_scope'], p['one_scope_graph_replay'], p['one_scope_eager_host_clock'])
PY
done
(timeout 600 python -m pytest tests/test_gpu_detect.py tests/test_gpu_describe.py -x -q -m gpu 2>&1 | tail -3)
