"""-m gpu: the view-sharded ladder with two ranks (gloo rendezvous, both ranks computing on GPU 0, exchange
staged through the host as the gloo backend requires) gives exactly the single-GPU result."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
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
    backend = os.environ.get("MODS_TEST_BACKEND", "gloo")      # nccl (= RCCL): one GPU per rank
    dev = int(os.environ["RANK"]) if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, h = 480, 360
    a, b, _ = _hard_pair(w, h, seed=21)
    t = torch.from_numpy(np.stack([a, b])).cuda(dev)
    torch.cuda.synchronize()
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(dev, d[0], d[1], 1)
    pkg.lib().mods_ransac_set_device(dev)
    # (the last two steps run the orientation in doHalfSIFT mode, as the HessianAffine steps of iters_MODS.ini do)
    steps = [pkg.LadderStep.make(tl, ph, half_orientation=ho) for tl, ph, ho in (((1,), 360.0, 0), ((1, 2, 4), 360.0, 1), ((1, 2, 4), 120.0, 1))]
    got = shard.match_ladder_distributed(pkg, ctx, t.data_ptr(), w, h, steps, dist, "cuda:" + str(dev), seed_time=31)
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


def _run_ranks(tmp_path, world, backend):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MODS_TEST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o


def _need_gpus(pkg, n):
    have = pkg.lib().mods_device_count()
    if have < n:
        pytest.skip("needs %d GPUs, this box has %d: RCCL between distinct devices is only exercised on a multi-GPU node "
                    "(on one GPU the same decomposition runs with device 0 listed repeatedly / over gloo, see the tests beside this one)" % (n, have))


@pytest.mark.parametrize("world", [1, 2])
def test_view_sharded_ladder(tmp_path, world):
    _run_ranks(tmp_path, world, "gloo")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_view_sharded_ladder_over_rccl(pkg, tmp_path, world):
    """The same ladder with one rank per GPU and the exchange as RCCL collectives on device buffers (backend nccl): identical to the
    one-GPU result.  Runs only where `world` GPUs are visible."""
    _need_gpus(pkg, world)
    _run_ranks(tmp_path, world, "nccl")


@pytest.mark.parametrize("n_dev", [2, 4, 8])
def test_multi_gpu_ladder_in_cpp_on_distinct_devices(pkg, n_dev):
    """mods_match_ladder_multi over n_dev DISTINCT devices: ncclCommInitAll + one grouped ncclAllGather per step (csrc/multi.hip),
    the branch that a repeated device 0 bypasses.  Identical to the one-GPU ladder; runs only where n_dev GPUs are visible."""
    _need_gpus(pkg, n_dev)
    _check_multi_ladder(pkg, list(range(n_dev)), expect_rccl=True)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_gpu_ladder_in_cpp(pkg, devices):
    """mods_match_ladder_multi (C++ host, csrc/multi.hip): views sharded over the listed devices, ONE exchange of the regions
    per step - ncclAllGather on device buffers when the devices are distinct ([0]: a one-rank communicator), plain device
    copies when a device is listed more than once (the 2- and 3-way sharding exercised on this one-GPU box) - and the query
    rows split over the devices.  Identical to the one-GPU ladder."""
    _check_multi_ladder(pkg, devices, expect_rccl=(len(devices) == 1))


def _check_multi_ladder(pkg, devices, expect_rccl):
    import torch
    from test_gpu_views import _hard_pair
    w, h = 480, 360
    a, b, _ = _hard_pair(w, h, seed=21)
    steps = [pkg.LadderStep.make((1,), 360.0), pkg.LadderStep.make((1, 2, 4), 360.0), pkg.LadderStep.make((1, 2, 4), 120.0)]
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    rep1, rep2 = pkg.ImgRep(ctx), pkg.ImgRep(ctx)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(31)
    want, wm = pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, steps, rep1, rep2, max_matches=100000)
    ra, rb = rep1.fetch(), rep2.fetch()
    multi = pkg.Multi(devices, w, h)
    assert multi.uses_rccl == expect_rccl
    pkg.ransac_pin_seed(31)
    got, gm = multi.match_ladder(a, b, steps, max_matches=100000)
    for f in ("steps_done", "n_views", "n_tentatives", "n_unique", "n_inliers", "ransac_samples", "ransac_lo", "ransac_rejects"):
        assert getattr(got, f) == getattr(want, f), f
    assert list(got.n_described) == list(want.n_described) and list(got.n_detected) == list(want.n_detected)
    assert list(got.H) == list(want.H) and np.array_equal(gm, wm) and got.n_inliers >= 15
    for bank, exp in ((multi.bank(0), ra), (multi.bank(1), rb)):
        assert np.array_equal(bank["desc"], exp["desc"]) and np.array_equal(bank["x"], exp["x"]) and np.array_equal(bank["id"], exp["id"])
    multi.close(); rep1.close(); rep2.close(); ctx.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_multi_gpu_ladder_half_lists_and_two_detectors(pkg, devices):
    """The whole step loop on several devices (mods_match_ladder_groups_multi): a HessianAffine detector whose steps ask for
    RootSIFT and HalfRootSIFT lists (iters_MODS.ini: Descriptors = RootSIFT, HalfRootSIFT; the HalfRootSIFT twins of a view travel
    behind its RootSIFT regions in the step's one exchange) next to a DoG detector that only has views in the first step;
    identical to mods_match_ladder_groups_dev on one GPU, field by field, banks included."""
    import torch
    from test_gpu_views import _hard_pair
    w, h = 480, 360
    a, b, _ = _hard_pair(w, h, seed=23)
    mk = pkg.LadderStep.make
    det_steps = [[mk((1,), 360.0), None, None],
                 [mk((1,), 360.0, half_orientation=1, fginn_half=0.8), mk((1, 2), 360.0, half_orientation=1, fginn_half=0.8),
                  mk((1, 2, 4), 120.0, half_orientation=1, fginn_half=0.8)]]
    det_params = [pkg.HessAffParams.dog(), pkg.HessAffParams.default()]
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    reps1, reps2 = [pkg.ImgRep(ctx), pkg.ImgRep(ctx)], [pkg.ImgRep(ctx), pkg.ImgRep(ctx)]
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(31)
    want, wm = pkg.match_ladder_dets_dev(ctx, t.data_ptr(), w, h, det_steps, det_params, reps1, reps2, min_matches=100000, max_matches=100000)
    banks = [[r.fetch() for r in reps1], [r.fetch() for r in reps2]]
    multi = pkg.Multi(devices, w, h)
    pkg.ransac_pin_seed(31)
    got, gm = multi.match_ladder_dets(a, b, det_steps, det_params, min_matches=100000, max_matches=100000)
    assert want.steps_done == 3 and want.n_tentatives > 100
    for f in ("steps_done", "n_views", "n_tentatives", "n_unique", "n_inliers", "ransac_samples", "ransac_lo", "ransac_rejects"):
        assert getattr(got, f) == getattr(want, f), f
    assert list(got.n_described) == list(want.n_described) and list(got.n_detected) == list(want.n_detected)
    assert list(got.n_unoriented) == list(want.n_unoriented)
    assert list(got.H) == list(want.H) and np.array_equal(gm, wm) and got.n_inliers >= 15
    for im in (0, 1):
        for det in (0, 1):
            bank, exp = multi.bank(im, det), banks[im][det]
            assert len(bank) == len(exp) > 30
            assert np.array_equal(bank["desc"], exp["desc"]) and np.array_equal(bank["x"], exp["x"]) and np.array_equal(bank["id"], exp["id"])
    multi.close()
    for r in reps1 + reps2:
        r.close()
    ctx.close()


def test_multi_gpu_rejects_images_larger_than_created(pkg):
    """mods_multi_create sizes the device buffers; a later call with larger images is refused instead of overflowing them."""
    multi = pkg.Multi([0, 0], 320, 240)
    big = np.zeros((480, 640), np.float32)
    with pytest.raises(pkg.ModsError, match="exceed"):
        multi.match_ladder(big, big, [pkg.LadderStep.make((1,), 360.0)])
    multi.close()


@pytest.mark.parametrize("iters", ["iters_ladder.ini", "iters_mser.ini"])
def test_cli_on_several_devices(tmp_path, iters):
    """MODS_DEVICES: the command line runs the whole configuration (HessianAffine steps with RootSIFT + HalfRootSIFT lists:
    tests/configs/iters_ladder.ini; MSER steps next to a HessianAffine step: iters_mser.ini) through
    mods_match_ladder_groups_multi and writes the SAME files as on one GPU."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mods = os.path.join(root, "mods-light-zmq_amd", "mods")
    cfg = os.path.join(root, "tests", "configs")
    g1, g6 = (os.path.join(root, "tests", "golden", n) for n in ("graf1.png", "graf6.png"))
    outs = {}
    for name, env in (("one", {}), ("multi", {"MODS_DEVICES": "0,0"})):
        wd = tmp_path / name
        wd.mkdir()
        args = [mods, g1, g6, "o1.png", "o2.png", "k1.txt", "k2.txt", "m.txt", "log.txt", "0", "0", "H.txt",
                os.path.join(cfg, "classic.ini"), os.path.join(cfg, iters)]
        p = subprocess.run(args, cwd=wd, env=dict(os.environ, MODS_RANSAC_SEED="4242", **env), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()
        outs[name] = {f: (wd / f).read_text() for f in ("m.txt", "k1.txt", "k2.txt", "H.txt")}
    assert len(outs["multi"]["m.txt"].splitlines()) >= 15
    for f in ("m.txt", "k1.txt", "k2.txt", "H.txt"):
        assert outs["multi"][f] == outs["one"][f], f
