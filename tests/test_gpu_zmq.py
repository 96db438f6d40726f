"""-m gpu: the ZMQ descriptor path end to end - ExtractPatchesColumn patches from the GPU (bit exact vs the oracle, even patch
size), sent by the C client to the daemon (HardNet on the MI355X through PyTorch-ROCm), descriptors written back into the
regions and used by the matcher."""
import ctypes as C
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mods-light-zmq_amd", "libmodszmq.so")
DAEMON = os.path.join(ROOT, "mods-light-zmq_amd", "zmq_daemon.py")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_patch_columns_bit_exact_and_hardnet_daemon(pkg):
    sys.path.insert(0, os.path.join(ROOT, "mods-light-zmq_amd"))
    import zmq_daemon as zd
    from test_cpu_zmq import _describe, _stop
    wire = C.CDLL(LIB)
    wire.mods_zmq_last_error.restype = C.c_char_p
    port = _free_port()
    endpoint = ("tcp://127.0.0.1:%d" % port).encode()
    d = subprocess.Popen([sys.executable, DAEMON, "--model", "hardnet", "--bind", endpoint.decode(), "--device", "cuda", "--seed", "5"],
                         stderr=subprocess.PIPE)
    line = ""
    for _ in range(20):                     # the ROCm runtime may print a warning first
        line = d.stderr.readline().decode()
        if "serving" in line or not line:
            break
    assert "serving" in line and "cuda" in line, line
    time.sleep(0.2)
    try:
        w, h = 480, 360
        img = synth.texture(w, h, seed=13)
        ctx = pkg.Context(0, w, h, 1)
        keys = ctx.detect_hessian_affine(img)
        # 1. RootSIFT run: the region list (orientation etc.) is the same in both modes
        sift_regs = ctx.orient_describe(img, keys)
        # 2. external descriptor through the ZMQ client hook
        hook = C.cast(wire.mods_zmq_descriptor_hook, C.c_void_p).value
        ep = C.create_string_buffer(endpoint)
        ctx.set_external_descriptor(hook, C.addressof(ep), 3.0 * np.sqrt(3.0), 32)
        regs = ctx.orient_describe(img, keys)
        assert len(regs) == len(sift_regs) > 100
        for f in ("x", "y", "s", "a11", "a12", "a21", "a22"):
            assert np.array_equal(regs[f], sift_regs[f])
        # the patches that went out = ExtractPatchesColumn of the oracle, bit for bit (32 = even patch size rule)
        got_p = ctx.patches_fetch(0, 32)
        want_p = orc.extract_patches_column(img, sift_regs, 3.0 * np.sqrt(3.0), 32)
        assert got_p.shape == want_p.shape and np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32))
        # the descriptors = the daemon's answer for those patches (8-bit quantised request, HardNet forward on the GPU)
        direct = _describe(wire, endpoint.decode(), want_p)
        assert direct.shape == (len(regs), 128)
        # (two forward passes of the same network on the GPU may pick different convolution algorithms: the integer
        # quantisation of the daemon can then flip by one step in a few entries)
        dq = np.clip(np.rint(direct), 0, 255).astype(np.int16)
        assert np.max(np.abs(regs["desc"].astype(np.int16) - dq)) <= 1 and np.mean(regs["desc"] == dq) > 0.999
        # and the GPU forward agrees with the same network on the CPU up to the quantisation step
        cpu = zd.build_model("hardnet", None, 5, "cpu")(np.clip(np.rint(want_p), 0, 255).astype(np.float32).reshape(-1, 1, 32, 32))
        assert np.mean(np.abs(direct - cpu) <= 1.0) > 0.999
        # matching works on these descriptors like on any others: an image against itself matches 1:1
        ctx.set_external_descriptor(None, None)
        tent, _ = ctx.match_fginn(regs, regs, 0.8)
        assert np.array_equal(tent["q"], tent["t"]) and len(tent) > 0.5 * len(regs)
        ctx.close()
    finally:
        _stop(wire, d, port)


@pytest.mark.parametrize("model", ["affnet", "orinet"])
def test_daemon_with_reference_weights_on_the_gpu(model):
    """The AffNet / OriNet daemon on the MI355X (PyTorch-ROCm forward) with the reference's own weights, over the wire, against
    the outputs of the reference's network classes on the CPU (tests/golden/nets.npz, tools/gen_nets_golden.py)."""
    from test_cpu_zmq import _describe, _stop
    wire = C.CDLL(LIB)
    wire.mods_zmq_last_error.restype = C.c_char_p
    g = np.load(os.path.join(ROOT, "tests", "golden", "nets.npz"))
    port = _free_port()
    endpoint = "tcp://127.0.0.1:%d" % port
    d = subprocess.Popen([sys.executable, DAEMON, "--model", model, "--bind", endpoint, "--device", "cuda", "--weights",
                          os.path.join(ROOT, "tests", "golden", "nets.npz")], stderr=subprocess.PIPE)
    line = ""
    for _ in range(20):
        line = d.stderr.readline().decode()
        if "serving" in line or not line:
            break
    assert "serving" in line and "cuda" in line, line
    time.sleep(0.2)
    try:
        got = _describe(wire, endpoint, g["patches"].astype(np.float32))
        want = g[model + "_out"]
        assert got.shape == want.shape
        assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-2)) < 1e-4     # fp32 convolutions: GPU vs the reference on the CPU
    finally:
        _stop(wire, d, port)
