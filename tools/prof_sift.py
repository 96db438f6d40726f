"""rocprof helper: N detect+describe calls on a synthetic 1080p pair; PHOTO=0/1 selects photometric normalisation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import __graft_entry__ as ge, synth
pkg = ge.load_package()
a, b, _ = synth.pair(1920, 1080, seed=2000)
t = torch.from_numpy(np.stack([a, b])).cuda(); torch.cuda.synchronize()
ctx = pkg.Context(0, 1920, 1080, 2)
desc = pkg.DescribeParams.default()
desc.photoNorm = int(os.environ.get("PHOTO", "1"))
for _ in range(6):
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), 2, 1920, 1080, None, desc)
ctx.sync(); print(nd, nr)
