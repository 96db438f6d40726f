#!/usr/bin/env python3
"""One mods_match_pair_dev call at a time on a 1080p pair in HBM: wall time and the stage times the call reports."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import synth
pkg = ge.load_package()
W, H = 1920, 1080
a, b, _ = synth.pair(W, H, seed=2000)
t = torch.from_numpy(np.stack([a, b])).cuda(); torch.cuda.synchronize()
ctx = pkg.Context(0, W, H, 2)
par = pkg.PairParams.default()
pkg.ransac_pin_seed(12345)
for n in (2, 1):
    ctx.pyramid_streams(n)
    for _ in range(3): pkg.match_pair_dev(ctx, t.data_ptr(), W, H, par)
    ts, st = [], []
    for _ in range(15):
        t0 = time.perf_counter()
        res, _ = pkg.match_pair_dev(ctx, t.data_ptr(), W, H, par)
        ts.append((time.perf_counter() - t0) * 1e3)
        st.append((res.ms_detect_describe, res.ms_match, res.ms_duplicates, res.ms_ransac))
    st = np.median(np.array(st), axis=0)
    print("pyramid streams %d: call median %.3f ms min %.3f | detect+describe %.3f match %.3f duplicates %.3f ransac %.3f" % (n, np.median(ts), min(ts), *st))
