#!/usr/bin/env python3
"""Per-stage HIP-event timing of detect+describe at 1080p, batch 2 (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import synth
pkg = ge.load_package()
W, H, B = 1920, 1080, 2
a, b, _ = synth.pair(W, H, seed=2000)
t = torch.from_numpy(np.stack([a, b])).cuda()
torch.cuda.synchronize()
ctx = pkg.Context(0, W, H, B)
stages = ["blur", "response", "resize", "nms", "localize", "baumberg", "sort", "orient", "describe", "match"]
for it in range(3):
    ctx.detect_describe_dev(t.data_ptr(), B, W, H)
    ctx.match_dev(0, 1)
N = 10
t0 = time.time()
for it in range(N):
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), B, W, H)
ctx.sync()
print("detect_describe: %.3f ms per pair" % ((time.time() - t0) / N * 1e3), nd, nr)
t0 = time.time()
for it in range(N):
    tc, _ = ctx.match_dev(0, 1)
print("match: %.3f ms per pair, %d tentatives" % ((time.time() - t0) / N * 1e3, len(tc)))
ctx.timing_enable(stages)
ctx.timing_reset()
for it in range(N):
    ctx.detect_describe_dev(t.data_ptr(), B, W, H)
    ctx.match_dev(0, 1)
for s in stages:
    ms, n, by = ctx.timing_read(s)
    print("%-10s %8.3f ms/pair  %4d scopes" % (s, ms / N, n // N))
