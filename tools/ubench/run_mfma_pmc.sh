cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/u1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/u1 -- $R/tools/ubench/mfma_i8.bin > /tmp/u1.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/u1/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:40], r["Grid_Size"])
    rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
t = glob.glob("/tmp/u1/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(t)):
    dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (d, n, g), c in rows.items():
    us = dur.get(d, 0)
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("disp %3d %-34s grid %8s  %8.1f us  clock %.2f GHz  mfma busy %.2f of SIMD cycles" % (d, n, g, us, cyc / us / 1e3 if us else 0, c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc))
PY
