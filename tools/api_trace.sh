#!/bin/bash
# HIP / HSA API call statistics of a short benchmark run (which runtime calls the host threads spend their time in)
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --hip-trace --hsa-trace --stats --output-format csv -d $OUT/t -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-match-leg "$@" > $OUT/run.log 2>&1
for f in $(find $OUT/t -name "*_stats.csv"); do cp $f $OUT/$(basename $f | sed 's/^[0-9]*_//'); done
rm -rf $OUT/t
for f in $OUT/hip_api_stats.csv $OUT/hsa_api_stats.csv; do echo "== $f"; head -22 $f | cut -c1-150; done
grep -o '"value": [0-9.]*' $OUT/run.log | head -1
