import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as ge, synth, orc
import test_gpu_match as tm
pkg = ge.load_package()
ctxA = pkg.Context(0, 1920, 1080, 2)
q, t = tm._rand_regions(1000, 6), tm._rand_regions(1500, 106)
g, _ = ctxA.match_fginn(q, t, 0.8); print('A', len(g))
a, b, H = synth.pair(1280, 960, seed=5)
tt = torch.from_numpy(np.stack([a, b])).cuda()
ctx2 = pkg.Context(0, 1280, 960, 2)
nd, nr = ctx2.detect_describe_dev(tt.data_ptr(), 2, 1280, 960)
print(nd, nr)
ra, rb = ctx2.regions_fetch(0), ctx2.regions_fetch(1)
got, u6 = ctx2.match_dev(0, 1)
print('dev', len(got))
got2, _ = ctx2.match_fginn(ra, rb)
print('host', len(got2))
got, u6 = ctx2.match_dev(0, 1)
print('dev again', len(got))
