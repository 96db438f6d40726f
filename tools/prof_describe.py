#!/usr/bin/env python3
"""Detect + orient + describe of one batch of 16 synthetic 1080p images (the unit of rounds 2-5; `prof_describe.py 32` = the launch size
of the round-6 pipeline), a few repetitions on one stream: the workload of the describe-stage profiles (tools/refresh_profiles.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import synth

pkg = ge.load_package()
W, H, B = 1920, 1080, (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
imgs = []
for i in range(B // 2):
    a, b, _ = synth.pair(W, H, seed=2000 + (i % 2))
    imgs += [a, b]
t = torch.from_numpy(np.stack(imgs)).cuda()
ctx = pkg.Context(0, W, H, B)
for it in range(4):
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), B, W, H)
ctx.sync()
print("regions per image", nr[:4])
