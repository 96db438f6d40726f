#!/bin/bash
# blur table of extract_small + orientation vote loop (mask prefetch, bin table in LDS): parity tests, then the describe leg of the
# new library and of the library with HEAD's describe.hip (same box)
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run26; mkdir -p $OUT
cd $R
(timeout 1500 python -m pytest tests/test_gpu_describe.py tests/test_gpu_pair.py tests/test_gpu_deep.py tests/test_gpu_zmq.py tests/test_gpu_distributed.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error" | tail -8) | tee $OUT/tests.txt
cd /tmp; export TMPDIR=/tmp
for v in new orihead; do
  L=""; [ $v = orihead ] && L=$R/mods-light-zmq_amd/_variants/libmodsgpu_orihead.so
  rm -rf /tmp/p_$v
  MODS_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o d -- python $R/tools/prof_describe.py > $OUT/prof_$v.log 2>&1
  f=$(find /tmp/p_$v -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -14 $f > $OUT/describe_leg_$v.csv
  echo "== $v"; cut -d, -f1-4 $OUT/describe_leg_$v.csv | cut -c1-150
done
