#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run5; mkdir -p $OUT
cd $R
(timeout 900 python -m pytest tests/test_gpu_describe.py tests/test_gpu_pair.py -x -q -m gpu -k "graph_replay or pipeline or bench or batch16" 2>&1 | tail -8) > $OUT/gputest.log
cat $OUT/gputest.log
bash tools/r05_ab_pyr.sh r05_run5/ab r04 X=1 th16j occ66 occ74 > /dev/null 2>&1
cat $OUT/ab/ab.log | sed 's/kps \[[^]]*\]//' 
for g in 1 0 1 0; do
  MODS_GRAPHS=$g timeout 300 python bench.py --no-cpu-baseline --no-match-leg 2>&1 | tail -1 > $OUT/bench_g$g.json
  python3 - $OUT/bench_g$g.json $g <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print("graphs", sys.argv[2], d['value'], d['host_cpu']['process_cpu_ms_per_pair'], d['host_cpu']['by_thread_name_ms_per_pair'], d['roofline_pyramid']['one_scope']['ms'])
PY
done
