#!/usr/bin/env python3
"""Which asynchronous operation keeps the HIP runtime's helper thread polling: ~0.3 s of GPU work is queued, followed by (a) nothing,
(b) an asynchronous device-to-host copy into pinned memory, (c) an event record, (d) a second stream waiting on an event; the main
thread then SLEEPS (no runtime call) and the CPU time of every other thread over the sleep is printed."""
import os, sys, time
import torch
def threads():
    d = {}
    for tid in os.listdir("/proc/self/task"):
        try: d[tid] = int(open("/proc/self/task/%s/schedstat" % tid).read().split()[0])
        except Exception: pass
    return d
x = torch.randn(8192, 8192, device="cuda")
out = torch.empty(1 << 20, device="cuda")
host = torch.empty(1 << 20, pin_memory=True)
s2 = torch.cuda.Stream()
torch.cuda.synchronize()
for what in ("kernels only", "copy D2H pinned", "event record", "second stream waits on event", "copy H2D pinned", "kernels only"):
    for _ in range(40): y = x @ x
    if what == "copy D2H pinned": host.copy_(out, non_blocking=True)
    if what == "copy H2D pinned": out.copy_(host, non_blocking=True)
    if what == "event record": ev = torch.cuda.Event(); ev.record()
    if what == "second stream waits on event":
        ev = torch.cuda.Event(); ev.record(); s2.wait_event(ev)
        with torch.cuda.stream(s2): out.add_(1.0)
    a = threads(); t0 = time.perf_counter()
    time.sleep(0.25)
    b = threads(); dt = time.perf_counter() - t0
    busy = sorted(((b[t] - a.get(t, 0)) * 1e-9 / dt, t) for t in b)[-2:]
    torch.cuda.synchronize()
    print("%-32s busiest other threads over a %.2f s sleep: %s" % (what, dt, [(round(v, 2), t) for v, t in busy]), flush=True)
