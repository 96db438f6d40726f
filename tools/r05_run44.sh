#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run44; mkdir -p $OUT
cd $R
(timeout 900 python -m pytest tests/test_gpu_describe.py -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -2)
cd /tmp; export TMPDIR=/tmp
for v in tier2 tier4864 tier4056 tier2 tier4864; do
  L=$R/mods-light-zmq_amd/_variants/libmodsgpu_$v.so
  rm -rf /tmp/p_$v
  MODS_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o d -- python $R/tools/prof_describe.py > $OUT/prof_$v.log 2>&1
  f=$(find /tmp/p_$v -name '*kernel_stats.csv' | head -1)
  python3 - $f $v <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 4 / 1e6
es = [r for r in rows if "extract_small" in r["Name"]][0]
print("%-10s leg %.3f ms per batch | extract_small: %s launches, %.1f us per batch (avg %.1f, min %.1f, max %.1f)" % (sys.argv[2], tot, es["Calls"], float(es["TotalDurationNs"]) / 4 / 1e3, float(es["AverageNs"]) / 1e3, float(es["MinNs"]) / 1e3, float(es["MaxNs"]) / 1e3))
PY
done
