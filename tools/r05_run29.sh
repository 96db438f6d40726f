#!/bin/bash
# per-kernel durations of the match stage at 60 156 x 47 177 (bench_match.py --c5only) from a kernel trace
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run29; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python $R/tools/bench_match.py --make > /dev/null 2>&1
rm -rf /tmp/pm; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o m -- python $R/tools/bench_match.py --c5only > $OUT/match.log 2>&1
grep "C5" $OUT/match.log
f=$(find /tmp/pm -name '*kernel_stats.csv' | head -1); cp $f $OUT/match_c5_kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/match_c5_kernel_stats.csv")):
    if "match" in r["Name"] or "copy" in r["Name"] or "fill" in r["Name"]:
        print("%-50s %5s avg %8.1f us min %8.1f"%(r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
