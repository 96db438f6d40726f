#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run40; mkdir -p $OUT
cd $R
(timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_zmq.py tests/test_gpu_distributed.py -q -m gpu -x 2>&1 | tail -3) | tee $OUT/tests.txt
( echo "# final round-5 library (one stream per pipeline worker, event waits, LO memo, round-5 kernels)"
  echo "# tools/stress_pipeline.py 30: the benchmark pipeline (6 GPU workers x 8 pairs + 8 verify workers, 8-bit host input) 30 x 48 pairs against the oracle chain"
  STRESS_VERBOSE=1 timeout 500 python tools/stress_pipeline.py 30 2>&1 | grep -v amdgpu.ids | tail -1
  echo "# the same with MODS_PIPELINE_STREAMS=2 (side stream + graph replay in the workers), 10 x 48 pairs"
  STRESS_VERBOSE=1 MODS_PIPELINE_STREAMS=2 timeout 400 python tools/stress_pipeline.py 10 2>&1 | grep -v amdgpu.ids | tail -1
  echo "# tools/stress_match.py 6 1500: six contexts repeat detect + describe + match of three 1080p pairs, every result compared with its first"
  timeout 500 python tools/stress_match.py 6 1500 2>&1 | grep -v amdgpu.ids | tail -1 ) > $OUT/concurrency_stress.log 2>&1
cat $OUT/concurrency_stress.log
timeout 400 python bench.py --no-cpu-baseline --steps 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['host_cpu']['process_cpu_ms_per_pair'], d['latency_ms_single_pair']['one_call_hbm_f32']['median'])"
