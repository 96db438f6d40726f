#!/bin/bash
# which thread of the benchmark process burns a core: runs bench.py in the background, finds its busiest thread over two seconds and
# prints that thread's backtrace (rocgdb).  usage (GPU box): bash tools/who_spins.sh <out tag> [env=value ...]
ulimit -c 0
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for kv in "$@"; do export $kv; done
python bench.py --steps 2500 --warmup 3 --no-cpu-baseline --no-match-leg $BENCH_ARGS > $OUT/bench.log 2>&1 &
PID=$!
sleep 40
python3 - $PID > $OUT/threads.txt <<'PY'
import os, sys, time
pid = sys.argv[1]
def snap():
    d = {}
    for tid in os.listdir("/proc/%s/task" % pid):
        try:
            d[tid] = (int(open("/proc/%s/task/%s/schedstat" % (pid, tid)).read().split()[0]), open("/proc/%s/task/%s/comm" % (pid, tid)).read().strip())
        except Exception:
            pass
    return d
a = snap(); time.sleep(2.0); b = snap()
rows = sorted(((b[t][0] - a.get(t, (0,))[0], t, b[t][1]) for t in b), reverse=True)
for ns, t, name in rows[:12]:
    print("%s %-16s %.3f cores" % (t, name, ns / 2e9))
PY
head -3 $OUT/threads.txt
TID=$(head -1 $OUT/threads.txt | cut -d' ' -f1)
echo "pid $PID busiest $TID"
python3 - $PID $TID <<'PY'
import sys, time
pid, tid = sys.argv[1:3]
def ticks():
    f = open("/proc/%s/task/%s/stat" % (pid, tid)).read().rsplit(")", 1)[1].split()
    return int(f[11]), int(f[12])          # utime, stime (clock ticks)
u0, s0 = ticks(); time.sleep(2.0); u1, s1 = ticks()
print("busiest thread over 2 s: user %d ticks, system %d ticks" % (u1 - u0, s1 - s0))
for name in ("wchan", "syscall"):
    try: print(name, open("/proc/%s/task/%s/%s" % (pid, tid, name)).read().strip())
    except Exception as e: print(name, "?", e)
try: print(open("/proc/%s/task/%s/stack" % (pid, tid)).read()[:600])
except Exception as e: print("stack ?", e)
PY
kill $PID; sleep 1; kill -9 $PID 2>/dev/null
tail -2 $OUT/bench.log | cut -c1-300
