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
    ctx = pkg.Context(0, W, H, 2)
    ctx.detect_describe_dev(dev.data_ptr(), 2, W, H)
    while not stop.is_set():
        try:
            ctx.match_dev(0, 1)
        except Exception:
            pass
    ctx.close()


spin = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libspin.so"))
out = (ctypes.c_uint * 3)()
ths = [threading.Thread(target=aggress) for _ in range(n_aggr)]
for t in ths: t.start()
time.sleep(1.0)
t0 = time.time()
try:
  for i in range(launches):
    if os.environ.get("SPIN_FP64"):
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
