# avg duration of kernels matching $1 in tools/prof_sift.py (6 detect+describe calls on a 1080p pair); MODS_LIB may select a variant
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python $GRAFT_REPO_ROOT/tools/prof_sift.py > /tmp/pk.log 2>&1
python3 - "$1" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pk/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[1] in r["Name"]:
        print("%-50s calls %4s avg %9.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
