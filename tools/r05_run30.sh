#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run30; mkdir -p $OUT
cd $R
(timeout 2400 python -m pytest tests/test_gpu_ransac.py tests/test_gpu_ransac_f.py tests/test_gpu_pair.py tests/test_gpu_cli.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error" | tail -8) | tee $OUT/tests.txt
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline 2> $OUT/bench_$i.err > $OUT/bench_$i.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_$i.json"))
print(d["value"], "pairs/s", d["host_cpu"]["process_cpu_ms_per_pair"], d["host_cpu"]["verify_workers_cpu_ms_per_pair"], d["host_cpu"]["by_thread_name_ms_per_pair"]["python"]["busiest"])
PY
done
