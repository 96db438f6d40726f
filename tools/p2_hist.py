#!/usr/bin/env python3
"""Window sizes of the measurement regions of the describe-leg workload (tools/prof_describe.py): histogram of P2 = 2 ceil(s mrSize) + 3
per size class of the extraction, and the sample counts per class ((P2)^2 bilinear taps per region)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import synth

pkg = ge.load_package()
W, H = 1920, 1080
a, b, _ = synth.pair(W, H, seed=2000)
t = torch.from_numpy(np.stack([a, b])).cuda()
ctx = pkg.Context(0, W, H, 2)
ctx.detect_describe_dev(t.data_ptr(), 2, W, H)
r = ctx.regions_fetch(0)
s = np.asarray(r["s"], dtype=np.float64)
mr = 3.0 * math.sqrt(3.0)
P2 = 2 * np.ceil(s * mr).astype(int) + 3
scale = (P2 - 2) / 41.0
direct = scale <= 0.4
print("regions", len(P2), "direct branch", int(direct.sum()))
edges = [0, 24, 32, 40, 48, 64, 80, 128, 256, 512, 1024, 1 << 20]
tot = float((P2[~direct].astype(float) ** 2).sum())
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (~direct) & (P2 > lo) & (P2 <= hi)
    if m.sum():
        print("P2 in (%4d, %7d]: %5d regions  %5.1f %% of the samples  mean P2 %.1f" % (lo, hi, m.sum(), 100 * (P2[m].astype(float) ** 2).sum() / tot, P2[m].mean()))
