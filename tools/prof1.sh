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
import json
line = [l for l in open(sys.argv[1] + "/bench.log") if l.startswith('{"metric"')][0]
d = json.loads(line)
pps = d["config"].get("pairs_per_step", 1)
ppb = int(d["config"]["overlap"].split(" x ")[1].split()[0]) if " x " in d["config"].get("overlap", "") else 1
N = float((d["steps"] + d["warmup"]) * pps + 8 * ppb)   # timed + warm-up pairs + the roofline leg (8 passes of one batch)
print("total ms %.2f  per pair %.3f" % (tot / 1e6, tot / 1e6 / N))
for r in rows[:22]:
    print("%-44s calls %5s avg %9.1f us %5.1f%%  per pair %.3f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]), int(r["TotalDurationNs"]) / 1e6 / N))
PY
