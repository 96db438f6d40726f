"""-m gpu: AffNet / OriNet in the place of Baumberg / the dominant orientation (imagerepresentation.cpp:786-856, 874-900).
The networks sit behind a callback; here they are deterministic stand-ins (plain numpy functions of the patch), so that the GPU
path (patches from the extraction kernels, keypoint bookkeeping around the callback) can be compared exactly with the oracle
driven by the same functions.  A second test runs the real models of the daemon over ZMQ."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_int))
MR = 3.0 * np.sqrt(3.0)


def shape_net(p):
    """(n, ps, ps) float32 -> (n, 3) float32: a smooth, well conditioned function of the patch."""
    p = p.astype(np.float32)
    m = p.reshape(len(p), -1).mean(1) + np.float32(1.0)
    top = p[:, : p.shape[1] // 2].reshape(len(p), -1).mean(1)
    left = p[:, :, : p.shape[2] // 2].reshape(len(p), -1).mean(1)
    a11 = np.float32(1.0) + np.float32(0.4) * (top / m - np.float32(0.5))
    a21 = np.float32(0.6) * (left / m - np.float32(0.5))
    a22 = np.float32(1.0) / a11 + np.float32(0.2) * (left / m - np.float32(0.5))
    return np.stack([a11, a21, a22], 1).astype(np.float32)


def ori_net(p):
    p = p.astype(np.float32)
    h = p.shape[1] // 2
    y = p[:, :h].reshape(len(p), -1).mean(1) - p[:, h:].reshape(len(p), -1).mean(1)
    x = p[:, :, :h].reshape(len(p), -1).mean(1) - p[:, :, h:].reshape(len(p), -1).mean(1)
    return np.stack([y, x + np.float32(1e-3)], 1).astype(np.float32)


def _hook(net, dim, seen):
    def fn(user, patches, n, ps, out, cap, dim_out):
        a = np.ctypeslib.as_array(patches, shape=(n, ps, ps)).copy()
        seen.append(a)
        r = net(a)
        assert r.shape == (n, dim) and cap >= n * dim
        np.ctypeslib.as_array(out, shape=(n * dim,))[:] = r.reshape(-1)
        dim_out[0] = dim
        return 0
    return FN(fn)


@pytest.mark.parametrize("w,h,seed", [(480, 360, 3), (640, 400, 9)])
def test_affnet_orinet_hooks_equal_oracle(pkg, w, h, seed):
    img = synth.texture(w, h, seed=seed)
    det = pkg.HessAffParams.default()
    det.doBaumberg = 0
    ctx = pkg.Context(0, w, h, 1)
    keys = ctx.detect_hessian_affine(img, det)
    assert len(keys) > 200 and np.all(keys["a12"] == 0)
    seen_s, seen_o = [], []
    hs, ho = _hook(shape_net, 3, seen_s), _hook(ori_net, 2, seen_o)
    ctx.set_external_shape(C.cast(hs, C.c_void_p).value, None, MR, 32)
    ctx.set_external_orientation(C.cast(ho, C.c_void_p).value, None, MR, 32)
    got = ctx.orient_describe(img, keys)
    ctx.set_external_shape(None, None); ctx.set_external_orientation(None, None)
    # the oracle, step by step as imagerepresentation.cpp does it
    regs = orc.regions_from_keys(keys)
    p1 = orc.extract_patches_column(img, regs, MR, 32)
    assert np.array_equal(seen_s[0].view(np.uint32), p1.view(np.uint32))
    regs = orc.affnet_apply(regs, shape_net(p1), w, h, MR)
    assert 0 < len(regs) < len(keys)                       # some keypoints fail the border / anisotropy tests
    regs = orc.filter_centres_inside(regs, w, h)
    p2 = orc.extract_patches_column(img, regs, MR, 32)
    assert np.array_equal(seen_o[0].view(np.uint32), p2.view(np.uint32))
    regs = orc.orinet_apply(regs, ori_net(p2))
    regs = orc.filter_touch_boundary(regs, w, h)
    want = orc.describe_rootsift(img, regs)
    assert len(got) == len(want) > 100
    for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["desc"], want["desc"])
    ctx.close()


def test_affnet_orinet_hardnet_daemons(pkg):
    """The three daemons of the deep configuration (random-weight AffNet, OriNet, HardNet on the MI355X) behind the hooks."""
    from test_cpu_zmq import _stop
    from test_gpu_zmq import _free_port, LIB, DAEMON
    wire = C.CDLL(LIB)
    procs = []
    try:
        eps = {}
        for model in ("affnet", "orinet", "hardnet"):
            port = _free_port()
            ep = "tcp://127.0.0.1:%d" % port
            d = subprocess.Popen([sys.executable, DAEMON, "--model", model, "--bind", ep, "--device", "cuda", "--seed", "5"], stderr=subprocess.PIPE)
            line = ""
            for _ in range(20):
                line = d.stderr.readline().decode()
                if "serving" in line or not line:
                    break
            assert "serving" in line, line
            procs.append((d, port)); eps[model] = C.create_string_buffer(ep.encode())
        time.sleep(0.3)
        w, h = 480, 360
        img = synth.texture(w, h, seed=13)
        det = pkg.HessAffParams.default()
        det.doBaumberg = 0
        ctx = pkg.Context(0, w, h, 1)
        keys = ctx.detect_hessian_affine(img, det)
        hook = C.cast(wire.mods_zmq_descriptor_hook, C.c_void_p).value
        ctx.set_external_shape(hook, C.addressof(eps["affnet"]), MR, 32)
        ctx.set_external_orientation(hook, C.addressof(eps["orinet"]), MR, 32)
        ctx.set_external_descriptor(hook, C.addressof(eps["hardnet"]), MR, 32)
        regs = ctx.orient_describe(img, keys)
        ctx.set_external_shape(None, None); ctx.set_external_orientation(None, None); ctx.set_external_descriptor(None, None)
        assert 50 < len(regs) <= len(keys)
        assert np.all(np.isfinite(regs["a11"])) and np.all(regs["a11"] * regs["a22"] - regs["a12"] * regs["a21"] > 0)
        assert regs["desc"].std() > 5                      # the descriptors vary
        tent, _ = ctx.match_fginn(regs, regs, 0.8)
        assert np.array_equal(tent["q"], tent["t"]) and len(tent) > 0.5 * len(regs)
        ctx.close()
    finally:
        for d, port in procs:
            _stop(wire, d, port)
