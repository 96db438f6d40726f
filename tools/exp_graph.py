#!/usr/bin/env python3
"""Round 5: where does the replayed detect + describe graph (mods_ctx_graphs) work?  One configuration per subprocess (a GPU fault
ends the process): image size x batch x stream kind x pyramid streams; prints OK (regions identical to eager launches) / the error."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import __graft_entry__ as ge, synth
pkg = ge.load_package()
w, h, n, nonblock, streams = [int(x) for x in sys.argv[1:6]]
imgs = [np.stack([synth.texture(w, h, seed=500 + 7 * j + i) for i in range(n)]) for j in range(2)]
buf = torch.from_numpy(imgs[0]).cuda(); torch.cuda.synchronize()
def run(ctx, im):
    buf.copy_(torch.from_numpy(im)); torch.cuda.synchronize()
    ctx.detect_describe_dev(buf.data_ptr(), n, w, h)
    return [ctx.regions_fetch(i) for i in range(n)]
e = pkg.Context(0, w, h, n, nonblocking=bool(nonblock)); e.pyramid_streams(streams)
want = [run(e, im) for im in imgs]; e.close()
c = pkg.Context(0, w, h, n, nonblocking=bool(nonblock)); c.pyramid_streams(streams); c.graphs(True)
for rep in range(3):
    for im, exp in zip(imgs, want):
        got = run(c, im)
        for a, b in zip(got, exp):
            assert len(a) == len(b) and np.array_equal(a["desc"], b["desc"]) and np.array_equal(a["x"], b["x"]), "regions differ"
print("OK replays", c.graph_replays())
''' % (ROOT, ROOT)
open("/tmp/exp_graph_child.py", "w").write(CHILD)
# usage: exp_graph.py [ENV=VALUE ...]   (e.g. DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: the runtime's replay of recorded AQL packets off)
env = dict(os.environ)
for a in sys.argv[1:]:
    k, _, v = a.partition("=")
    env[k] = v
print("environment:", " ".join(sys.argv[1:]) or "(default)")
for (w, h) in ((1280, 720), (1920, 1080)):
    for n in (2, 16):
        for nonblock in (0, 1):
            for streams in (1, 2):
                p = subprocess.run([sys.executable, "/tmp/exp_graph_child.py", str(w), str(h), str(n), str(nonblock), str(streams)],
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
                out = p.stdout.decode().strip().split("\n")
                print("%dx%d n=%2d nonblocking=%d streams=%d: %s" % (w, h, n, nonblock, streams, out[-1][:160]), flush=True)
