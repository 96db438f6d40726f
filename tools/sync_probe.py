#!/usr/bin/env python3
"""How a host thread waits for the GPU on this box: CPU time of the waiting thread and of the runtime's helper threads during a
~1 s stream synchronisation, with the default device flags and with hipDeviceScheduleBlockingSync (argument: spin | blocking | yield)."""
import ctypes, os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "spin"
import torch
hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = ctypes.CDLL(line.split()[-1]); break
flags = {"spin": 1, "yield": 2, "blocking": 4}[mode]
print("hipSetDeviceFlags(%s) ->" % mode, hip.hipSetDeviceFlags(flags), flush=True)
def threads():
    d = {}
    for tid in os.listdir("/proc/self/task"):
        try: d[tid] = int(open("/proc/self/task/%s/schedstat" % tid).read().split()[0])
        except Exception: pass
    return d
x = torch.randn(8192, 8192, device="cuda")
torch.cuda.synchronize()
for rep in range(2):
    a = threads(); t0 = time.perf_counter(); c0 = time.thread_time()
    for _ in range(40): y = x @ x
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter(); c1 = time.thread_time(); b = threads()
    others = sorted(((b[t] - a.get(t, 0)) * 1e-9, t) for t in b if t != str(os.getpid()))[-3:]
    print("%s: queued in %.3f s, waited %.3f s; waiting thread used %.3f s of CPU; busiest other threads %s" % (mode, t1 - t0, t2 - t1, c1 - c0, [(round(v, 3), t) for v, t in others]), flush=True)
