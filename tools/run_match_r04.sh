# round 4: matcher A/B on the GPU box - parity tests first, then the kernel times of the shipped library and of the variants
# under mods-light-zmq_amd/_variants/ (MODS_LIB selects one) on the configs[4]-sized lists of tools/bench_match.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_match
mkdir -p $OUT $R/tools/_cache
cd $R
[ -n "$SKIP_TESTS" ] && echo skipped > $OUT/tests.log || (timeout 900 python -m pytest tests/test_gpu_match.py -x -q -m gpu 2>&1 | tail -5) > $OUT/tests.log
python tools/bench_match.py --make > /dev/null 2>&1
cd /tmp
for v in $R/mods-light-zmq_amd/libmodsgpu.so $R/mods-light-zmq_amd/_variants/libmodsgpu_*.so; do
  for blocks in ${MATCH_BLOCKS_LIST:-default}; do
    echo "== $(basename $v) blocks=$blocks"
    [ "$blocks" = default ] && unset MODS_MATCH_BLOCKS || export MODS_MATCH_BLOCKS=$blocks
    rm -rf /tmp/ks; MODS_LIB=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/tools/bench_match.py 2>&1 | grep "^C[25]"
    python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "match_" in r["Name"]: print("   %-28s calls %4s avg %8.1f min %8.1f max %8.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  done
done > $OUT/variants.log 2>&1
cat $OUT/tests.log; cat $OUT/variants.log
