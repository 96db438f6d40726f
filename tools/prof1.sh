# rocprofv3 kernel stats of bench.py with one GPU worker (per-kernel time per pair in batch mode); args: out-dir name, extra bench args
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof1}
shift
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --no-cpu-baseline --gpu-workers 1 --steps 6 --warmup 1 "$@" > $OUT/bench.log 2>&1
echo rc=$?
grep '^{"metric"' $OUT/bench.log | cut -c1-200
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
N = 136.0   # 7 steps x 16 pairs + the roofline leg (6 x 4 pairs)
print("total ms %.2f  per pair %.3f" % (tot / 1e6, tot / 1e6 / N))
for r in rows[:22]:
    print("%-44s calls %5s avg %9.1f us %5.1f%%  per pair %.3f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]), int(r["TotalDurationNs"]) / 1e6 / N))
PY
