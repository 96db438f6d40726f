cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/tools/_cache
[ -f $R/tools/_cache/match_fixture.npz ] || python $R/tools/bench_match.py --make > /dev/null 2>&1
for v in libmodsgpu.so _variants/libmodsgpu_mfmaonly.so; do
for mode in "" --rand --const128 --zero; do
  echo "== $v data=$mode"
  rm -rf /tmp/ks; MODS_LIB=$R/mods-light-zmq_amd/$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/tools/bench_match.py --c5only $mode > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "match_nn1" in r["Name"]: print("   %-28s calls %4s avg %8.1f min %8.1f max %8.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
done
