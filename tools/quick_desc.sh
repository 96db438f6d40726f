# GPU box: descriptor parity tests, then kernel statistics of the describe leg (16 images of 1080p per batch, one stream)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-quick}
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then (cd $R && timeout 900 python -m pytest tests/test_gpu_describe.py -x -q -m gpu 2>&1 | tail -5); fi
rm -rf $OUT/dstats
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dstats -- python $R/tools/prof_describe.py > $OUT/run.log 2>&1
cp $(find $OUT/dstats -name "*kernel_stats.csv" | head -1) $OUT/describe_leg_kernel_stats.csv
rm -rf $OUT/dstats
python3 - $OUT/describe_leg_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total %.2f ms per batch of 16 images" % (tot / 4e6))
for r in rows[:14]:
    print("%-40s calls %4s  %8.3f ms/batch %5.1f%%" % (r["Name"].replace("mods::", "").replace("void ", "")[:40], r["Calls"], int(r["TotalDurationNs"]) / 4e6, float(r["Percentage"])))
PY
