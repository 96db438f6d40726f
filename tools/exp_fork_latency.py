#!/usr/bin/env python3
"""Scale space of a 16-image batch on a context of its own: one stream against the side stream (fork / join), as one timed scope and
as wall time per detect + describe call, with and without an idle pipeline in the process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import synth
pkg = ge.load_package()
W, H, B = 1920, 1080, 16
imgs = []
for i in range(B // 2):
    a, b, _ = synth.pair(W, H, seed=2000 + (i % 2)); imgs += [a, b]
t = torch.from_numpy(np.stack(imgs)).cuda()
ctx = pkg.Context(0, W, H, B)
def leg(tag):
    for n in (1, 2, 1, 2):
        ctx.pyramid_streams(n)
        for _ in range(2): ctx.detect_describe_dev(t.data_ptr(), B, W, H)
        ctx.timing_enable(["pyramid"]); ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(8): ctx.detect_describe_dev(t.data_ptr(), B, W, H)
        wall = (time.perf_counter() - t0) / 8
        ms = ctx.timing_read("pyramid")[0] / 8
        ctx.timing_enable([])
        t0 = time.perf_counter()
        for _ in range(8): ctx.detect_describe_dev(t.data_ptr(), B, W, H)
        wall2 = (time.perf_counter() - t0) / 8
        print("%-28s streams %d: pyramid scope %.3f ms, call %.3f ms (%.3f ms without the scope)" % (tag, n, ms, wall * 1e3, wall2 * 1e3), flush=True)
leg("context alone")
pipe = pkg.Pipeline(0, W, H, pkg.PairParams.default(), 6, 8, 8)
leg("idle pipeline in the process")
pipe.close()
# second part: a context created AFTER a pipeline, and measured after the pipeline has worked
pipe = pkg.Pipeline(0, W, H, pkg.PairParams.default(), 6, 8, 8)
ctx.close()
ctx = pkg.Context(0, W, H, B)
leg("context created after the pipeline")
pair = t[:2].contiguous()
for rep in range(2):
    pend = 0
    for i in range(200):
        if pend >= pipe.capacity - 1: pipe.next(); pend -= 1
        pipe.submit(pair.data_ptr(), i); pend += 1
    while pend: pipe.next(); pend -= 1
    leg("after %d pairs through the pipeline" % (200 * (rep + 1)))
pipe.close()
leg("pipeline closed")
