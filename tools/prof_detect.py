#!/usr/bin/env python3
"""Per-stage HIP-event timing of the detector at 1080p (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import synth
pkg = ge.load_package()
W, H, B = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 2
imgs = np.stack([synth.texture(W, H, seed=100 + i) for i in range(B)])
t = torch.from_numpy(imgs).cuda()
ctx = pkg.Context(0, W, H, B)
# PYR_STREAMS=1: every launch of the scale space alone on the GPU (what bench.py's "isolated" events and the roofline of the blur kernel
# are about); default 2 = as shipped, the small octaves on the side stream next to the large octaves' last level and NMS
ctx.pyramid_streams(int(os.environ.get("PYR_STREAMS", "2")))
stages = os.environ.get("STAGES", "blur,blur_small,response,resize,nms,pyramid,localize,baumberg,sort").split(",")
for it in range(3):
    ctx.detect_hessian_affine_dev(t.data_ptr(), B, W, H, fetch=False)
ctx.sync()
N = 10
t0 = time.time()
for it in range(N):
    counts = ctx.detect_hessian_affine_dev(t.data_ptr(), B, W, H, fetch=False)
ctx.sync()
dt = (time.time() - t0) / N
print("batch %d: %.3f ms per call, %.3f ms per image, kps %s" % (B, dt * 1e3, dt * 1e3 / B, counts))
ctx.timing_enable(stages)
ctx.timing_reset()
for it in range(N):
    ctx.detect_hessian_affine_dev(t.data_ptr(), B, W, H, fetch=False)
for s in stages:
    ms, n, by = ctx.timing_read(s)
    print("%-10s %8.3f ms/call  %4d scopes/call  %s" % (s, ms / N, n // N, ("%.1f GB/s" % (by / N / (ms / N) / 1e6)) if by else ""))
