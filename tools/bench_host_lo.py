#!/usr/bin/env python3
"""Host-side pieces of the homography verification, timed on this machine's cores (no GPU needed): u2h on 4500 inliers (its moment
sums through the SIMD table and in scalar code) and the checks behind the model (scalar statement against the lanes-wide form)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
from test_cpu_lo_fast import _points, _frames, P

pkg = ge.load_package()
M = pkg.lib()
M.mods_test_host_hchecks.restype = C.c_int


def best(fn, reps=200, rounds=5):
    b = 1e9
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        b = min(b, (time.perf_counter() - t0) / reps * 1e6)
    return b


u, h = _points(6000, 11)
n = 4500
inl = np.ascontiguousarray(np.sort(np.random.default_rng(1).permutation(6000)[:n]).astype(np.int32))
H, Cv = np.zeros(9), np.zeros(81)
print(os.popen("grep -m1 'model name' /proc/cpuinfo").read().strip())
for form, name in ((1, "lin_hgN + cov_mat as written"), (2, "30 folded sums, scalar"), (101, "table of 1 lane"), (104, "table of 4 lanes"), (108, "table of 8 lanes"), (0, "product")):
    if M.mods_test_host_cov(P(u), P(inl), n, form, P(Cv)) != 0:
        continue
    print("normu + moment matrix of %d inliers, %-30s %6.1f us" % (n, name, best(lambda: M.mods_test_host_cov(P(u), P(inl), n, form, P(Cv)))))
for form in (1, 0):
    print("u2h form %d %6.1f us" % (form, best(lambda: M.mods_test_host_u2h(P(u), P(inl), n, form, P(H)))))
laf = _frames(u, 3)
err = np.zeros(6000)
M.mods_test_host_errfn(0, P(u), 6000, P(h), 0, P(err))
flags = np.ascontiguousarray((err <= 16.0).astype(np.uint8))
par = pkg.RansacParams.default()
mask, Ho = np.zeros(6000, np.uint8), np.zeros(9)
for lanes in (0, 1, 4, 8):
    if M.mods_test_host_hchecks(P(u), P(laf), 6000, P(flags), P(h), C.byref(par), lanes, P(mask), P(Ho)) < 0:
        continue
    print("checks behind the model, %d inliers, lanes %d: %6.1f us" % (int(flags.sum()), lanes, best(lambda: M.mods_test_host_hchecks(P(u), P(laf), 6000, P(flags), P(h), C.byref(par), lanes, P(mask), P(Ho)), 100)))
