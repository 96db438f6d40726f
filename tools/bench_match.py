#!/usr/bin/env python3
"""Matcher micro-benchmark (development aid): BASELINE configs[4]-sized descriptor lists through mods_match_reps, HIP-event
time of the match stage.  The lists are the RootSIFT regions of six synthetic 1080p pairs; they are cached in
tools/_cache/match_fixture.npz (not committed) because generating the images takes longer than everything else.

  python tools/bench_match.py [--make]      --make: (re)build the cache on a GPU box (writes gpurun_out/match_fixture.npz too)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

CACHE = os.path.join(ROOT, "tools", "_cache", "match_fixture.npz")


def main():
    import torch
    pkg = ge.load_package()
    W, H = 1920, 1080
    ctx = pkg.Context(0, W, H, 2)
    if "--make" in sys.argv or not os.path.exists(CACHE):
        import synth
        qs, ts = [], []
        for i in range(6):
            a, b, _ = synth.pair(W, H, seed=2000 + i)
            t = torch.from_numpy(np.stack([a, b])).cuda()
            ctx.detect_describe_dev(t.data_ptr(), 2, W, H)
            qs.append(ctx.regions_fetch(0)); ts.append(ctx.regions_fetch(1))
        q, t = np.concatenate(qs), np.concatenate(ts)
        os.makedirs(os.path.dirname(CACHE), exist_ok=True)
        np.savez(CACHE, q=q, t=t)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "match_fixture.npz"), q=q, t=t)
    d = np.load(CACHE)
    q, t = d["q"], d["t"]
    # power experiments: the same lists with constant / uniformly random descriptors (the tentatives are meaningless)
    if "--zero" in sys.argv:
        q["desc"][:] = 0; t["desc"][:] = 0
    if "--const128" in sys.argv:
        q["desc"][:] = 128; t["desc"][:] = 128
    if "--rand" in sys.argv:
        rng = np.random.default_rng(1)
        q["desc"][:] = rng.integers(0, 256, q["desc"].shape, dtype=np.uint8); t["desc"][:] = rng.integers(0, 256, t["desc"].shape, dtype=np.uint8)
    rq, rt = pkg.ImgRep(ctx, 1 << 17), pkg.ImgRep(ctx, 1 << 17)
    rq.append_host(q); rt.append_host(t)
    for name, (a, b) in (("C5", (rq, rt)),):
        tent, _, _ = pkg.match_reps(ctx, a, b)
        ctx.timing_enable(["match"]); ctx.timing_reset()
        reps = 10
        t0 = time.time()
        for _ in range(reps):
            tent, _, _ = pkg.match_reps(ctx, a, b)
        wall = (time.time() - t0) / reps
        ms, n, _ = ctx.timing_read("match")
        ops = 2.0 * len(a) * len(b) * 128
        if hasattr(pkg.lib(), "mods_debug_match_stats"):    # library built with -DMATCH_STATS
            import ctypes
            st = (ctypes.c_ulonglong * 2)()
            pkg.lib().mods_debug_match_stats(st)
            print("nn1 exact path: %d of %d (query block, tile) pairs = %.3f" % (st[1], st[0], st[1] / max(1, st[0])))
        print("%s: %d x %d, %d tentatives, match stage %.4f ms (%.1f TOP/s, %.3f of 5 POP/s), wall %.3f ms, checksum %d"
              % (name, len(a), len(b), len(tent), ms / reps, ops / (ms / reps * 1e-3) / 1e12, ops / (ms / reps * 1e-3) / 5e15, wall * 1e3,
                 int(tent["t"].astype(np.int64).sum() + tent["q"].astype(np.int64).sum())))
    if "--c5only" in sys.argv:
        return
    # one 1080p pair (configs[1] size)
    n1, n2 = 10040, 8969
    r1, r2 = pkg.ImgRep(ctx, 1 << 15), pkg.ImgRep(ctx, 1 << 15)
    r1.append_host(q[:n1]); r2.append_host(t[:n2])
    pkg.match_reps(ctx, r1, r2)
    ctx.timing_enable(["match"]); ctx.timing_reset()
    for _ in range(20):
        tent, _, _ = pkg.match_reps(ctx, r1, r2)
    ms, n, _ = ctx.timing_read("match")
    print("C2: %d x %d, %d tentatives, match stage %.4f ms" % (n1, n2, len(tent), ms / 20))


if __name__ == "__main__":
    main()
