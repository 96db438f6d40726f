#!/bin/bash
# host-side verification changes (first batch of 8 samples, no design matrix of all correspondences, no residual rows, moment sums
# and the checks behind the model through the SIMD table) + the blur table of extract_small: the whole GPU suite, the host LO timings on
# the box's cores, then the bench line twice
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run28; mkdir -p $OUT
cd $R
python tools/bench_host_lo.py > $OUT/host_lo.txt 2>&1; cat $OUT/host_lo.txt
(timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error" | tail -8) | tee $OUT/tests.txt
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline 2> $OUT/bench_$i.err > $OUT/bench_$i.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_$i.json"))
print(d["value"], "pairs/s", d["host_cpu"])
PY
done
