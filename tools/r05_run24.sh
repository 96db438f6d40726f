#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run24; mkdir -p $OUT
cd $R
for v in bprof sprof oprof; do
  MODS_LIB=$R/mods-light-zmq_amd/_variants/libmodsgpu_$v.so timeout 300 python tools/prof_describe.py 2>&1 | grep -a "prof:" > $OUT/$v.log
  echo "== $v: $(wc -l < $OUT/$v.log) lines"
  python3 - $OUT/$v.log <<'PY'
import re,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for l in open(sys.argv[1], errors='ignore'):
    m=re.match(r"(.*?) prof: (.*)", l)
    if not m: continue
    name=m.group(1).strip(); rest=m.group(2)
    cnt[name]+=1
    for k,v in re.findall(r"([A-Za-z_+ ]+?) (\d+)(?= |$)", rest.split("cycles",1)[-1]):
        acc[name][k.strip(": ")]+=float(v)
for name in acc:
    tot=sum(acc[name].values())
    print(name, cnt[name], {k: "%.0f (%.0f%%)" % (v/cnt[name], 100*v/tot) for k,v in acc[name].items()})
PY
done
