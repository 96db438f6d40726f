"""CPU tests of the ZMQ descriptor-daemon protocol: message bodies (PNG column of 8-bit patches, float32 replies),
request splitting at 2000 patches, the daemon's REP loop with a deterministic model and with the HardNet / AffNet /
OriNet architectures on the CPU."""
import ctypes as C
import io
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mods-light-zmq_amd", "libmodszmq.so")
DAEMON = os.path.join(ROOT, "mods-light-zmq_amd", "zmq_daemon.py")


@pytest.fixture(scope="module")
def wire(pkg):
    assert os.path.exists(LIB), "libmodszmq.so not built"
    lib = C.CDLL(LIB)
    lib.mods_zmq_last_error.restype = C.c_char_p
    return lib


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _describe(lib, endpoint, patches, cap_dim=512, timeout_ms=60000):
    p = np.ascontiguousarray(patches, np.float32)
    n, ps = p.shape[0], p.shape[1]
    out = np.zeros(n * cap_dim, np.float32)
    dim = C.c_int()
    rc = lib.mods_zmq_describe(endpoint.encode(), p.ctypes.data_as(C.c_void_p), n, ps, out.ctypes.data_as(C.c_void_p),
                               C.c_size_t(out.size), C.byref(dim), timeout_ms)
    assert rc == 0, lib.mods_zmq_last_error().decode()
    return out[:n * dim.value].reshape(n, dim.value)


def _start(model, port, extra=()):
    p = subprocess.Popen([sys.executable, DAEMON, "--model", model, "--bind", "tcp://127.0.0.1:%d" % port, "--device", "cpu"] + list(extra),
                         stderr=subprocess.PIPE)
    line = ""
    for _ in range(20):                       # "serving ..." is printed once the model is built (runtime warnings may come first)
        line = p.stderr.readline().decode()
        if "serving" in line or not line:
            break
    assert "serving" in line, line
    time.sleep(0.2)
    return p


def _stop(lib, p, port):
    # zero-byte request = shutdown message of mods_zmq_serve
    import ctypes.util
    z = C.CDLL("/opt/conda/lib/libzmq.so.5")
    z.zmq_ctx_new.restype = C.c_void_p; z.zmq_socket.restype = C.c_void_p
    z.zmq_socket.argtypes = [C.c_void_p, C.c_int]; z.zmq_connect.argtypes = [C.c_void_p, C.c_char_p]
    z.zmq_send.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]; z.zmq_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    z.zmq_close.argtypes = [C.c_void_p]; z.zmq_ctx_term.argtypes = [C.c_void_p]
    ctx = z.zmq_ctx_new(); s = z.zmq_socket(ctx, 3)    # ZMQ_REQ
    z.zmq_connect(s, ("tcp://127.0.0.1:%d" % port).encode())
    z.zmq_send(s, None, 0, 0)
    buf = C.create_string_buffer(8); z.zmq_recv(s, buf, 8, 0)
    z.zmq_close(s); z.zmq_ctx_term(ctx)
    assert p.wait(timeout=30) == 0


def test_message_bodies(wire):
    """Request = PNG of an 8-bit (ps*n) x ps image, fp32 -> u8 by round-half-even with saturation (cv::imencode on a float
    Mat); any PNG reader (here PIL, in the reference's daemons cv2.imdecode) gets the patches back."""
    rng = np.random.default_rng(0)
    p = rng.uniform(-20, 280, (5, 32, 32)).astype(np.float32)
    p[0, 0, :6] = [0.5, 1.5, 2.5, 254.5, 255.5, -0.5]
    png, ln = C.POINTER(C.c_ubyte)(), C.c_size_t()
    assert wire.mods_zmq_encode_request(p.ctypes.data_as(C.c_void_p), 5, 32, C.byref(png), C.byref(ln)) == 0
    raw = bytes(bytearray(png[:ln.value]))
    img = np.asarray(Image.open(io.BytesIO(raw)))
    assert img.shape == (160, 32) and img.dtype == np.uint8
    want = np.clip(np.rint(p.astype(np.float64)), 0, 255).astype(np.uint8).reshape(160, 32)     # np.rint: ties to even
    assert np.array_equal(img, want)
    assert list(img[0, :6]) == [0, 2, 2, 254, 255, 0]
    pix, n, ps = C.POINTER(C.c_ubyte)(), C.c_int(), C.c_int()
    assert wire.mods_zmq_decode_request(png, ln, C.byref(pix), C.byref(n), C.byref(ps)) == 0
    assert (n.value, ps.value) == (5, 32)
    assert np.array_equal(np.ctypeslib.as_array(pix, shape=(160, 32)), want)
    wire.mods_zmq_free(png); wire.mods_zmq_free(pix)
    # not a column of square patches -> refused
    buf = io.BytesIO(); Image.fromarray(np.zeros((50, 32), np.uint8)).save(buf, format="PNG"); b = buf.getvalue()
    arr = (C.c_ubyte * len(b)).from_buffer_copy(b)
    assert wire.mods_zmq_decode_request(arr, C.c_size_t(len(b)), C.byref(pix), C.byref(n), C.byref(ps)) != 0


def test_round_trip_with_stats_model_and_request_splitting(wire):
    port = _free_port()
    d = _start("stats", port)
    try:
        rng = np.random.default_rng(1)
        for n in (1, 7, 2000, 4100):        # 4100 = three requests (2000 + 2000 + 100), answered in order
            p = rng.integers(0, 256, (n, 32, 32)).astype(np.float32)
            got = _describe(wire, "tcp://127.0.0.1:%d" % port, p)
            f = p.reshape(n, -1).astype(np.float64)
            want = np.stack([f.mean(1), f.std(1), f.min(1), f.max(1)], axis=1).astype(np.float32)
            assert got.shape == (n, 4) and np.array_equal(got, want)
    finally:
        _stop(wire, d, port)


@pytest.mark.parametrize("model,dim", [("hardnet", 128), ("affnet", 3), ("orinet", 2)])
def test_network_daemons_on_cpu(wire, model, dim):
    """The three reference daemons' architectures and post-processing behind the same wire format (seeded weights)."""
    sys.path.insert(0, os.path.join(ROOT, "mods-light-zmq_amd"))
    import importlib
    zd = importlib.import_module("zmq_daemon")
    port = _free_port()
    d = _start(model, port, ["--seed", "3"])
    try:
        rng = np.random.default_rng(2)
        p = rng.integers(0, 256, (9, 32, 32)).astype(np.float32)
        got = _describe(wire, "tcp://127.0.0.1:%d" % port, p)
        assert got.shape == (9, dim)
        want = zd.build_model(model, None, 3, "cpu")(p.reshape(9, 1, 32, 32))
        assert np.allclose(got, want, rtol=0, atol=1e-5 if model != "hardnet" else 1.0)     # hardnet: integer quantisation
        if model == "hardnet":
            assert np.array_equal(got, np.rint(got)) and got.min() >= 0 and got.max() <= 255
            assert np.mean(got == want) > 0.99
        if model == "affnet":
            assert np.all(np.abs(got[:, 0] - 1) <= 1) and np.all(np.abs(got[:, 2] - 1) <= 1)        # tanh + 1
    finally:
        _stop(wire, d, port)


def test_client_times_out_without_a_daemon(wire):
    p = np.zeros((1, 32, 32), np.float32)
    out = np.zeros(512, np.float32); dim = C.c_int()
    rc = wire.mods_zmq_describe(("tcp://127.0.0.1:%d" % _free_port()).encode(), p.ctypes.data_as(C.c_void_p), 1, 32,
                                out.ctypes.data_as(C.c_void_p), C.c_size_t(512), C.byref(dim), 300)
    assert rc != 0 and b"zmq_recv" in wire.mods_zmq_last_error()


def _nets_fixture():
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nets.npz"))
    weights = {tag: {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + ".")} for tag in ("affnet", "orinet")}
    return g, weights


def test_daemon_networks_match_the_reference_checkpoints():
    """tests/golden/nets.npz (tools/gen_nets_golden.py): outputs of the REFERENCE's AffNetFast / OriNetFast classes
    (build/affnet_server.py, build/orinet_server.py) with the reference's build/AffNet.pth / OriNet.pth on fixed patches.
    The daemon's modules take the same tensors (strict load) and give the same numbers on the CPU."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("zmq_daemon", DAEMON)
    zd = importlib.util.module_from_spec(spec); spec.loader.exec_module(zd)
    g, weights = _nets_fixture()
    patches = g["patches"].astype(np.float32)[:, None]
    for tag in ("affnet", "orinet"):
        got = zd.build_model(tag, weights=weights[tag], device="cpu")(patches)
        assert got.shape == g[tag + "_out"].shape
        assert np.max(np.abs(got - g[tag + "_out"])) < 1e-5, tag
    assert np.all(g["affnet_out"][:, 0] > 0.5) and np.all(g["affnet_out"][:, 2] > 0.5)      # the +1 of AffNetFast.forward
