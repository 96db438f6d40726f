"""-m gpu: the view-sharded ladder with two ranks (gloo rendezvous, both ranks computing on GPU 0, exchange
staged through the host as the gloo backend requires) gives exactly the single-GPU result."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import synth
    from test_gpu_views import _hard_pair
    pkg = ge.load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("shard", os.path.join(%r, "mods-light-zmq_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec); spec.loader.exec_module(shard)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, h = 480, 360
    a, b, _ = _hard_pair(w, h, seed=21)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    steps = [pkg.LadderStep.make(tl, ph) for tl, ph in (((1,), 360.0), ((1, 2, 4), 360.0), ((1, 2, 4), 120.0))]
    got = shard.match_ladder_distributed(pkg, ctx, t.data_ptr(), w, h, steps, dist, "cuda:0", seed_time=31)
    if rank == 0:
        rep1, rep2 = pkg.ImgRep(ctx), pkg.ImgRep(ctx)
        pkg.ransac_pin_seed(31)
        res, m = pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, steps, rep1, rep2, max_matches=100000)
        assert got["steps_done"] == res.steps_done and got["n_views"] == res.n_views
        assert got["n_described"] == list(res.n_described)
        assert got["n_tentatives"] == res.n_tentatives and got["n_unique"] == res.n_unique
        assert got["n_inliers"] == res.n_inliers >= 15
        assert got["stats"] == [res.ransac_samples, res.ransac_lo, res.ransac_rejects]
        assert np.array_equal(got["matches"], m)
        assert np.array_equal(got["H"], np.array(res.H))
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok", got["n_inliers"])
''') % (ROOT, ROOT, ROOT)


@pytest.mark.parametrize("world", [1, 2])
def test_view_sharded_ladder(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o
