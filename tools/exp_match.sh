# development aid: matcher variants (libmodsgpu_v*.so built with other -D flags): stage time and nn1 / pass-2 kernel times
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in $R/mods-light-zmq_amd/libmodsgpu.so $R/mods-light-zmq_amd/libmodsgpu_v*.so; do
  echo "== $v"
  rm -rf /tmp/ks; MODS_LIB=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/tools/bench_match.py 2>&1 | grep "^C[25]"
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "match_nn1" in r["Name"] or "match_fginn" in r["Name"]: print("   %s min %.1f max %.1f us" % (r["Name"][6:22], float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
