#!/usr/bin/env python3
"""Repeats the body of tests/test_gpu_pair.py::test_bench_pipeline_path_vs_oracle and reports HOW a result differs from the oracle
chain when it does (development aid for an intermittent failure)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import pipeline_oracle as po

pkg = ge.load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
partial = len(sys.argv) > 2 and sys.argv[2] == "partial"      # two motions per pair (bench.py --inlier-ratio): ~60 RANSAC samples per pair
w, h = 1920, 1080
n_pairs, n_sub = (3, 48) if partial else (6, 48)
if partial:
    import synth
    oracle = []
    for i in range(n_pairs):
        a, b, Ht = synth.pair_partial(w, h, seed=2300 + i, frac=0.55)
        a, b = np.round(a).clip(0, 255).astype(np.float32), np.round(b).clip(0, 255).astype(np.float32)   # the pipeline is fed 8-bit images
        oracle.append((a, b, Ht, po.match_pair(a, b, seed_time=12345)))
else:
    oracle = [po.cached_pair(w, h, 2000 + i, 12345) for i in range(n_pairs)]
pinned = []
for a, b, _, _ in oracle:
    buf = pkg.PinnedBuffer((2, h, w), np.uint8)
    buf.array[...] = np.stack([a, b]).astype(np.uint8)
    pinned.append(buf)
par = pkg.PairParams.default()
pkg.ransac_pin_seed(12345)
bad = 0
if os.environ.get("STRESS_VERBOSE"): print("oracle ready", flush=True)
t0 = time.time()
for rep in range(reps):
    pipe = pkg.Pipeline(0, w, h, par, 6, 8, 8)
    got, pending = [], 0
    for i in range(n_sub):
        if pending >= pipe.capacity - 1:
            got.append(pipe.next_matches()); pending -= 1
        pipe.submit_host(pinned[i % n_pairs].ptr.value, i, u8=True); pending += 1
    while pending:
        got.append(pipe.next_matches()); pending -= 1
    for i, (res, tag, m) in enumerate(got):
        want = oracle[i % n_pairs][3]
        exp_m = want["u6"][want["mask"]][:, [0, 1, 3, 4]]
        facts = dict(tag=tag, nd=list(res.n_detected), nr=list(res.n_described), nt=res.n_tentatives, nu=res.n_unique,
                     stats=[res.ransac_samples, res.ransac_lo, res.ransac_rejects], ni=res.n_inliers)
        wf = dict(tag=i, nd=want["n_detected"], nr=want["n_described"], nt=want["n_tentatives"], nu=want["n_unique"], stats=want["stats"],
                  ni=want["n_inliers"])
        if facts != wf or not np.array_equal(m, exp_m):
            bad += 1
            print("rep %d submission %d differs: got %s want %s" % (rep, i, facts, wf))
            if m.shape == exp_m.shape:
                d = np.nonzero(np.any(m != exp_m, axis=1))[0]
                print("  rows that differ: %d of %d, first %s" % (len(d), len(m), d[:8]))
                for r in d[:4]:
                    print("   got", m[r], "want", exp_m[r])
                print("  same set, other order:", sorted(map(tuple, m)) == sorted(map(tuple, exp_m)))
            else:
                print("  shapes", m.shape, exp_m.shape)
    pipe.close()
    if os.environ.get("STRESS_VERBOSE"): print("rep %d done at %.1f s" % (rep, time.time() - t0), flush=True)
print("%d repetitions, %d differing results, %.1f s" % (reps, bad, time.time() - t0))
