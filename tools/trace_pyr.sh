cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -- python $R/tools/prof_detect.py 16 > /tmp/tp.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/tp/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full detect call: find the last u8/first blur of a call: take the last 60 kernels
names = [r["Kernel_Name"] for r in rows]
# find index of last occurrence of the kernel with '<1, 32' hmm: pick last 'accept_kernel' and go back to previous 'accept_kernel'
acc = [i for i, n in enumerate(names) if "accept_kernel" in n]
a, b = acc[-2], acc[-1]
t0 = int(rows[a + 1]["Start_Timestamp"])
for r in rows[a + 1: b + 1]:
    print("%8.1f %8.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-60:]))
PY
