cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pm1
mkdir -p $OUT
python $R/tools/bench_match.py 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/tools/bench_match.py > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "match_" in r["Name"]:
        print("%-40s calls %4s avg %9.1f us min %8.1f max %8.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
