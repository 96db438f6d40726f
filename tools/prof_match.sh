# rocprofv3 kernel stats of the matcher kernels in a bench config (default c5: 4096 x 4096 pair)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pm}
CFG=${2:-c5}
mkdir -p $OUT
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --config $CFG --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench.log 2>&1
echo rc=$?
grep '^{"metric"' $OUT/bench.log | cut -c1-400
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "match_" in r["Name"]:
        print("%-40s calls %4s avg %9.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
