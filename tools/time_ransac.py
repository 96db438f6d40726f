"""Wall time of the verification stage: libmodsgpu (GPU-scored) vs the reference's own degensac (oracle/_ref,
one CPU thread) on the same correspondences and seed.  Prints one line per case."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import __graft_entry__ as ge, fsynth, refdeg
from test_gpu_ransac import make_corr
pkg = ge.load_package()
def t(fn, reps=3):
    fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
print("case, n, ours_ms, reference_ms, samples, inliers")
for n, ratio in ((2000, 0.5), (6000, 0.8), (24000, 0.9), (24000, 0.3)):
    u = make_corr(n, ratio, 0.6, 1)
    a, ra = t(lambda: pkg.ransac_h(u, 16.0, seed_time=5)); b, rb = t(lambda: refdeg.ransac_h(u, 16.0, seed_time=5))
    assert ra["I"] == rb["I"] and ra["samples"] == rb["samples"]
    print("H, %d (%.0f%% inliers), %.2f, %.2f, %d, %d" % (n, 100 * ratio, a, b, ra["samples"], ra["I"]))
for n, ratio, plane in ((2000, 0.5, 0.0), (6000, 0.7, 0.5), (24000, 0.9, 1.0), (24000, 0.4, 0.3)):
    u, _, _ = fsynth.two_view(n, ratio, plane, 0.6, seed=2, size=(4096, 4096))
    a, ra = t(lambda: pkg.ransac_f(u, 16.0, sym_check=1, seed_time=5)); b, rb = t(lambda: refdeg.ransac_f(u, 16.0, sym_check=1, seed_time=5))
    assert ra["I"] == rb["I"] and ra["samples"] == rb["samples"], (ra["I"], rb["I"], ra["samples"], rb["samples"])
    print("F, %d (%.0f%% inliers, %.0f%% on a plane), %.2f, %.2f, %d, %d" % (n, 100 * ratio, 100 * plane, a, b, ra["samples"], ra["I"]))
