#!/bin/bash
# arbitrary PMC passes over the describe leg: usage tools/pmc_passes.sh <out tag> "<counters of pass 1>" "<counters of pass 2>" ...
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/passes.txt
i=0
for c in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p$i -- python $R/tools/prof_describe.py > $OUT/p$i.log 2>&1
  python3 $R/tools/pmc_summary.py $(find $OUT/p$i -name "*counter_collection.csv" | head -1) >> $OUT/passes.txt
  rm -rf $OUT/p$i
done
grep -A12 "extract_pipe_kernel<6\|extract_small\|baumberg_kernel<2\|orient_kernel" $OUT/passes.txt | head -150
