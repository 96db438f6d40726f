#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run33; mkdir -p $OUT
cd $R
for m in default sleep:50 sleep:100 sleep:200 default; do
  if [ $m = default ]; then unset MODS_SYNC; else export MODS_SYNC=$m; fi
  timeout 600 python bench.py --no-cpu-baseline --steps 10 2> $OUT/bench_$m.err > $OUT/bench_$m.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_$m.json"))
print("$m", d["value"], "pairs/s", d["host_cpu"]["process_cpu_ms_per_pair"], d["host_cpu"]["verify_workers_cpu_ms_per_pair"], d["host_cpu"]["gpu_workers_cpu_ms_per_pair"], d["host_cpu"]["by_thread_name_ms_per_pair"]["python"]["busiest"])
PY
done
