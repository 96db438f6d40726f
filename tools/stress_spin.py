#!/usr/bin/env python3
"""Contexts running the matcher (MODS_MATCH_MASK=2: match_nn1_kernel only) next to waves that do nothing but park constants in
v0..v95 and LDS (tools/ubench/spin_victim.hip): do the matcher's neighbours lose register or LDS contents?
usage: [MODS_LIB=...] [MODS_MATCH_MASK=2] python tools/stress_spin.py <aggressor threads> <victim launches>"""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import synth

pkg = ge.load_package()
n_aggr = int(sys.argv[1]) if len(sys.argv) > 1 else 3
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 300
W, H = 1920, 1080
a, b = synth.pair(W, H, seed=2000)[:2]
dev = torch.from_numpy(np.stack([a, b]).astype(np.float32)).cuda()
torch.cuda.synchronize()
stop = threading.Event()


def aggress():
    if os.environ.get("AGGR_TOP"):        # synthetic aggressor: tools/ubench/vgpr_top.hip as a library, mode = AGGR_TOP
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libvgprtop.so"))
        while not stop.is_set():
            if lib.top_launch(int(os.environ["AGGR_TOP"]), 4096, 400):
                print("top_launch failed"); return
        return
    if os.environ.get("AGGR_MFMA"):       # synthetic aggressor: tools/ubench/mfma_aggr.hip, nothing but MFMA streams (mode, data = AGGR_MFMA, AGGR_DATA)
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libmfmaaggr.so"))
        while not stop.is_set():
            if lib.aggr_launch(int(os.environ["AGGR_MFMA"]), int(os.environ.get("AGGR_DATA", "1")), 1024, 2000):
                print("aggr_launch failed"); return
        return
    kind = os.environ.get("AGGR", "match")     # what the aggressor contexts run: match | pair | mser | view | dog
    d = pkg.view_ctx_dims(W, H) if kind == "view" else (W, H)
    ctx = pkg.Context(0, d[0], d[1], 2)
    ctx.detect_describe_dev(dev.data_ptr(), 2, W, H)
    while not stop.is_set():
        try:
            if kind == "pair":        # detect + describe + match + duplicate filter + LO-RANSAC (GPU scoring kernels)
                pkg.match_pair_dev(ctx, dev.data_ptr(), W, H, max_matches=1 << 16)
            elif kind == "mser":
                ctx.detect_describe_dev(dev.data_ptr(), 2, W, H, det=pkg.HessAffParams.mser())
            elif kind == "dog":
                ctx.detect_describe_dev(dev.data_ptr(), 2, W, H, det=pkg.HessAffParams.dog())
                ctx.detect_describe_dev(dev.data_ptr(), 2, W, H, det=pkg.HessAffParams.harris())
            elif kind == "view":
                ctx.detect_describe_view_dev(dev.data_ptr(), W, H, 4.0, 0.6)
            else:
                ctx.match_dev(0, 1)
        except Exception as e:
            if kind != "match":
                print("aggressor failed:", e); return
    ctx.close()


spin = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libspin.so"))
out = (ctypes.c_uint * 3)()
ths = [threading.Thread(target=aggress) for _ in range(n_aggr)]
for t in ths: t.start()
time.sleep(1.0)
t0 = time.time()
tot = [0, 0, 0]
try:
  for i in range(launches):
    if os.environ.get("SPIN_SVDVAR"):
        rc = spin.svdvar_launch(int(os.environ["SPIN_SVDVAR"]), 2048, 300, out)
        tot = [x + y for x, y in zip(tot, out)] if i else list(out)
        out[0], out[1], out[2] = tot[0], tot[1], tot[2]
    elif os.environ.get("SPIN_PART"):
        part = int(os.environ["SPIN_PART"])
        rc = spin.part_launch(part, 2048, 300, out)
        tot = [x + y for x, y in zip(tot, out)] if i else list(out)
        out[0], out[1], out[2] = tot[0], tot[1], tot[2]
    elif os.environ.get("SPIN_SVD"):
        rc = spin.svd_launch(2048, 300, out)            # waves that repeat the detector's 2x2 fp64 Jacobi SVD
    elif os.environ.get("SPIN_PK"):
        rc = spin.pk_launch(8192, 2000, out)            # waves that keep running packed-fp32 chains
    elif os.environ.get("SPIN_FP64"):
        rc = spin.fp64_launch(8192, 400, out)           # waves that keep computing fp64 / fp32 chains (counts: fp64 in the first, fp32 in the second column)
    elif os.environ.get("SPIN_LOADS"):
        rc = spin.load_launch(8192, 4000, out)          # waves that keep loading known LDS / global words
    else:
        rc = spin.spin_launch(2048, int(os.environ.get("SPIN_ITERS", "300000")), out)
    assert rc == 0, rc
finally:
  stop.set()
for t in ths: t.join()
print("%d aggressor contexts, %d victim waves: %d wrong register / global values, %d wrong LDS words, %.1f s"
      % (n_aggr, out[2], out[0], out[1], time.time() - t0))
