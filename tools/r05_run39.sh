#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run39; mkdir -p $OUT
cd $R
( echo "# final round-5 library (one stream per pipeline worker, event waits, LO memo, round-5 kernels)"
  echo "# tools/stress_pipeline.py 30: the benchmark pipeline (6 GPU workers x 8 pairs + 8 verify workers, 8-bit host input) 30 x 48 pairs against the oracle chain"
  timeout 700 python tools/stress_pipeline.py 30 2>&1 | grep -v amdgpu.ids | tail -2
  echo "# the same with MODS_PIPELINE_STREAMS=2 (side stream + graph replay in the workers), 10 x 48 pairs"
  MODS_PIPELINE_STREAMS=2 timeout 400 python tools/stress_pipeline.py 10 2>&1 | grep -v amdgpu.ids | tail -2
  echo "# tools/stress_match.py 6 1500: six contexts repeat detect + describe + match of three 1080p pairs, every result compared with its first"
  timeout 500 python tools/stress_match.py 6 1500 2>&1 | grep -v amdgpu.ids | tail -2 ) > $OUT/concurrency_stress.log 2>&1
cat $OUT/concurrency_stress.log
