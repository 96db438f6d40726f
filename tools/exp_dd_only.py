#!/usr/bin/env python3
"""Development aid: detect + orient + describe only, N contexts on N host threads, 32 images per call (the pipeline's launch shape):
the rate the GPU stages in front of the matcher can reach without the matcher and the verifier behind them."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import synth

pkg = ge.load_package()
W, H, B = 1920, 1080, 32
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 4
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 12
imgs = []
for i in range(B // 2):
    a, b, _ = synth.pair(W, H, seed=2000 + (i % 2))
    imgs += [a, b]
t = torch.from_numpy(np.stack(imgs)).cuda()
ctxs = [pkg.Context(0, W, H, B) for _ in range(NW)]
for c in ctxs:
    c.detect_describe_dev(t.data_ptr(), B, W, H); c.sync()
def work(c):
    for _ in range(REP):
        c.detect_describe_dev(t.data_ptr(), B, W, H)
    c.sync()
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t0
print("contexts %d: %.1f pairs/s (%.3f ms per 32-image call per context)" % (NW, NW * REP * B / 2 / dt, dt / REP * 1e3))
