#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run37; mkdir -p $OUT
cd $R
for m in 1 -1 0 -1 1; do
  export MODS_GRAPHS=$m
  timeout 600 python bench.py --no-cpu-baseline --steps 10 2> $OUT/bench_$m.err > $OUT/bench_$m.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_$m.json"))
print("MODS_GRAPHS=$m", d["value"], "pairs/s", d["host_cpu"]["process_cpu_ms_per_pair"], d["host_cpu"]["verify_workers_cpu_ms_per_pair"], d["host_cpu"]["gpu_workers_cpu_ms_per_pair"], d["host_cpu"]["by_thread_name_ms_per_pair"]["python"], d["latency_ms_single_pair"]["pipeline_one_in_flight_host_u8"]["median"])
PY
done
