#!/bin/bash
# whole GPU suite + concurrency stress + bench twice (library without legacy-stream copies, LO memo)
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run32; mkdir -p $OUT
cd $R
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40) > $OUT/tests_full.txt; tail -3 $OUT/tests_full.txt
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline 2> $OUT/bench_$i.err > $OUT/bench_$i.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_$i.json"))
print(d["value"], "pairs/s", d["host_cpu"]["process_cpu_ms_per_pair"], d["host_cpu"]["verify_workers_cpu_ms_per_pair"], d["host_cpu"]["by_thread_name_ms_per_pair"]["python"]["busiest"], d["latency_ms_single_pair"]["one_call_hbm_f32"]["median"])
PY
done
