# rocprofv3 kernel stats of N detect+describe calls on a synthetic 1080p pair (tools/prof_sift.py); extra env passes through
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof_desc}
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/tools/prof_sift.py > $OUT/run.log 2>&1
echo rc=$?
tail -2 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total ms %.2f" % (tot / 1e6))
for r in rows[:16]:
    print("%-44s calls %4s avg %9.1f us %5.1f%%" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
