#!/usr/bin/env python3
"""Scale space of a 1080p batch as a function of the images per launch (is a smaller working set served from the Infinity Cache?):
the whole pyramid as one scope (MODS_STAGE_PYRAMID) and the blur launches, per image, for 1 / 2 / 4 / 8 / 16 images per call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import synth

pkg = ge.load_package()
W, H, B = 1920, 1080, 16
imgs = []
for i in range(2):
    a, b, _ = synth.pair(W, H, seed=2000 + i)
    imgs += [a, b]
t = torch.from_numpy(np.stack([imgs[i % 4] for i in range(B)])).cuda()
for n in (16, 8, 4, 2, 1):
    ctx = pkg.Context(0, W, H, n)
    for streams in (2, 1):
        ctx.pyramid_streams(streams)
        for _ in range(3):
            ctx.detect_dev(t.data_ptr(), n, W, H) if hasattr(ctx, "detect_dev") else ctx.detect_describe_dev(t.data_ptr(), n, W, H)
        ctx.timing_enable(["pyramid"]); ctx.timing_reset()
        reps = 8
        for _ in range(reps):
            for g in range(B // n):
                ctx.detect_describe_dev(t[g * n:(g + 1) * n].data_ptr(), n, W, H)
        ms, cnt, _ = ctx.timing_read("pyramid")
        ctx.timing_enable(["blur", "nms"]); ctx.timing_reset()
        for _ in range(reps):
            for g in range(B // n):
                ctx.detect_describe_dev(t[g * n:(g + 1) * n].data_ptr(), n, W, H)
        bms = ctx.timing_read("blur")[0]; nms = ctx.timing_read("nms")[0]
        ctx.timing_enable([])
        print("images per call %2d streams %d: pyramid scope %.4f ms per 16 images (%d scopes), blur launches %.4f, nms %.4f" % (n, streams, ms / reps, cnt, bms / reps, nms / reps))
    ctx.close()
