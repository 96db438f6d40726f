#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run35; mkdir -p $OUT
cd $R
(timeout 1500 python -m pytest tests/test_gpu_pair.py -q -m gpu -x 2>&1 | tail -3) | tee $OUT/tests.txt
for m in default default; do
  if [ $m = default ]; then unset MODS_SYNC; else export MODS_SYNC=$m; fi
  timeout 600 python bench.py --no-cpu-baseline --steps 10 2> $OUT/bench_$m.err > $OUT/bench_$m.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_$m.json"))
print("$m", d["value"], "pairs/s", d["host_cpu"]["process_cpu_ms_per_pair"], d["host_cpu"]["verify_workers_cpu_ms_per_pair"], d["host_cpu"]["gpu_workers_cpu_ms_per_pair"], d["host_cpu"]["by_thread_name_ms_per_pair"], d["latency_ms_single_pair"]["pipeline_one_in_flight_host_u8"])
PY
done
