"""CPU tests of the mods command line: argument handling, the .ini reader's semantics (inline ';' comments,
typed getters), and the loud failure without a GPU (no CPU path)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODS = os.path.join(ROOT, "mods-light-zmq_amd", "mods")
CFG = os.path.join(ROOT, "tests", "configs")
G1, G6 = (os.path.join(ROOT, "tests", "golden", n) for n in ("graf1.png", "graf6.png"))


def run(args, cwd):
    p = subprocess.run([MODS] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return p.returncode, p.stderr.decode()


@pytest.fixture(scope="module", autouse=True)
def built(pkg):
    assert os.path.exists(MODS), "mods CLI not built (make -C mods-light-zmq_amd)"


def test_usage_and_bad_arguments(tmp_path):
    rc, err = run([], tmp_path)
    assert rc == 1 and "Usage: mods img1 img2" in err
    base = [G1, G6, "o1", "o2", "k1", "k2", "m", "log", "0"]
    rc, err = run(base + ["3", "H", os.path.join(CFG, "classic.ini"), os.path.join(CFG, "iters_one_view.ini")], tmp_path)
    assert rc == 1 and "wrong correspondence verification type" in err
    rc, err = run(base + ["1", "no_such_H.txt", os.path.join(CFG, "classic.ini"), os.path.join(CFG, "iters_one_view.ini")], tmp_path)
    assert rc == 1 and "Cannot open ground truth file no_such_H.txt" in err      # ver_type 1 reads the homography (mods.cpp:89-104)
    rc, err = run(base + ["1"], tmp_path)
    assert rc == 1 and "Ground truth homography file is needed" in err
    rc, err = run(base + ["0", "H", "/nonexistent.ini", os.path.join(CFG, "iters_one_view.ini")], tmp_path)
    assert rc == 1 and "Can't load /nonexistent.ini" in err
    rc, err = run(["/nonexistent.png"] + base[1:] + ["0", "H", os.path.join(CFG, "classic.ini"), os.path.join(CFG, "iters_one_view.ini")], tmp_path)
    assert rc == 1 and "cannot open /nonexistent.png" in err


def test_config_parsing_and_no_cpu_path(tmp_path):
    """Without a HIP device the CLI parses everything, reports what it skips, and then refuses to run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_cli.py")
    rc, err = run([G1, G6, "o1", "o2", "k1", "k2", "m", "log", "0", "0", "H", os.path.join(CFG, "classic.ini"),
                   os.path.join(CFG, "iters_ladder.ini")], tmp_path)
    assert rc == 1
    assert "detector ORB is outside this build" in err and "HalfRootSIFT is outside" not in err
    assert "Image1: 800x640, Image2: 800x640" in err
    assert "no MI355X / HIP device available" in err and "no CPU path" in err
    assert not os.path.exists(tmp_path / "m")


def test_npz_keypoint_files_round_trip_with_numpy(tmp_path):
    """The .npz reader / writer of the command line (cnpy's role, imagerepresentation.cpp:1257-1316, 1355-1513): an archive
    written by numpy is read and written back member for member, and numpy reads the result."""
    import subprocess
    import numpy as np
    rng = np.random.default_rng(5)
    n = 37
    src = {"xy": rng.random((n, 2)) * 500, "scales": rng.random((n, 1)) * 9 + 1, "responses": rng.standard_normal((n, 1)),
           "A": rng.standard_normal((n, 4)), "descs": rng.integers(0, 256, (n, 128), dtype=np.uint8)}
    np.savez(tmp_path / "in.npz", **src)
    p = subprocess.run([MODS, "--npz-echo", str(tmp_path / "in.npz"), str(tmp_path / "out.npz")], stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 0, p.stderr.decode()
    got = np.load(tmp_path / "out.npz")
    assert sorted(got.files) == sorted(src)
    for k, v in src.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    # a one-dimensional member and an empty one
    np.savez(tmp_path / "in2.npz", angles=np.arange(5, dtype=np.float64), empty=np.zeros((0, 2)))
    p = subprocess.run([MODS, "--npz-echo", str(tmp_path / "in2.npz"), str(tmp_path / "out2.npz")], stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 0, p.stderr.decode()
    got = np.load(tmp_path / "out2.npz")
    assert np.array_equal(got["angles"], np.arange(5.0)) and got["empty"].shape == (0, 2)
    # compressed archives are refused with a message
    np.savez_compressed(tmp_path / "c.npz", **src)
    p = subprocess.run([MODS, "--npz-echo", str(tmp_path / "c.npz"), str(tmp_path / "o.npz")], stderr=subprocess.PIPE, timeout=60)
    assert p.returncode != 0 and b"compressed" in p.stderr


def test_npz_reader_refuses_malformed_archives(tmp_path):
    """A corrupt or hostile k1 / k2 .npz (pre-extracted mode) must be refused with a message: truncated central directory,
    headers without descr / shape, sizes that overflow, payloads shorter than the shape claims."""
    import numpy as np
    src = {"xy": np.arange(12, dtype=np.float64).reshape(6, 2), "descs": np.arange(6 * 128, dtype=np.uint8).reshape(6, 128)}
    np.savez(tmp_path / "ok.npz", **src)
    good = (tmp_path / "ok.npz").read_bytes()

    def echo(data):
        (tmp_path / "bad.npz").write_bytes(data)
        p = subprocess.run([MODS, "--npz-echo", str(tmp_path / "bad.npz"), str(tmp_path / "o.npz")], stderr=subprocess.PIPE, timeout=60)
        return p.returncode, p.stderr.decode()

    rc, _ = echo(good)
    assert rc == 0
    cases = {
        "descr": good.replace(b"'descr'", b"'dexcr'"),
        "shape": good.replace(b"'shape'", b"'shxpe'"),
        "paren": good.replace(b"(6, 2)", b" 6, 2 "),
        "huge": good.replace(b"(6, 2)", b"(9, 9)"),                      # payload shorter than the shape claims
        "overflow": good.replace(b"(6, 128), }", b"(99999999999, 99999999999), }"[:len(b"(6, 128), }")]),
    }
    for name, data in cases.items():
        assert len(data) == len(good), name
        rc, err = echo(data)
        assert rc not in (0, -11, -6) and rc > 0 and err.strip(), (name, rc, err)      # refused with a message, no crash
    # central directory entry whose name length runs past the end of the file
    cd = good.rfind(b"PK\x01\x02")
    data = bytearray(good)
    data[cd + 28:cd + 30] = (60000).to_bytes(2, "little")
    rc, err = echo(bytes(data))
    assert rc > 0 and "central directory" in err
    # every truncation of the archive is refused, none crashes
    for cut in range(0, len(good) - 1, 37):
        rc, _ = echo(good[:cut])
        assert rc > 0, cut


def test_jpeg_input(tmp_path):
    """JPEG files (cv::imread, mods.cpp:116-118) are decoded by libmodsjpeg.so: colour and grey files, both LoadColor settings;
    the decoder's output agrees with PIL's decoder within the +-1 a JPEG decoder is specified to."""
    import ctypes as C
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:96, 0:128]
    rgb = np.stack([127 + 100 * np.sin(xx / 9.0), 127 + 100 * np.cos(yy / 7.0), 60 + xx], -1).clip(0, 255).astype(np.uint8)
    Image.fromarray(rgb).save(tmp_path / "c.jpg", quality=92, subsampling=0)
    Image.fromarray(rgb[:, :, 1]).save(tmp_path / "g.jpg", quality=92)
    lib = C.CDLL(os.path.join(ROOT, "mods-light-zmq_amd", "libmodsjpeg.so"))
    lib.mods_jpeg_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.c_char_p]
    for name, colour, ch_want in (("c.jpg", 1, 3), ("c.jpg", 0, 1), ("g.jpg", 1, 3), ("g.jpg", 0, 1)):
        buf = C.POINTER(C.c_ubyte)()
        w, h, ch = C.c_int(), C.c_int(), C.c_int()
        err = C.create_string_buffer(256)
        assert lib.mods_jpeg_read(str(tmp_path / name).encode(), colour, C.byref(buf), C.byref(w), C.byref(h), C.byref(ch), err) == 0, err.value
        assert (w.value, h.value, ch.value) == (128, 96, ch_want)
        got = np.ctypeslib.as_array(buf, (96, 128, ch_want)).astype(np.int32).copy()
        lib.mods_jpeg_free(buf)
        ref = np.asarray(Image.open(tmp_path / name).convert("RGB" if colour else "L")).astype(np.int32).reshape(96, 128, ch_want)
        assert np.abs(got - ref).max() <= 2 and np.abs(got - ref).mean() < 0.5
    err = C.create_string_buffer(256)
    buf = C.POINTER(C.c_ubyte)()
    (tmp_path / "bad.jpg").write_bytes(b"\xff\xd8\xff\xe0 not a jpeg")
    assert lib.mods_jpeg_read(str(tmp_path / "bad.jpg").encode(), 1, C.byref(buf), C.byref(w), C.byref(h), C.byref(ch), err) == -1 and err.value
    # the command line takes the file (and then stops at the missing GPU, or runs on one)
    Image.fromarray(np.asarray(Image.open(G1).convert("RGB"))).save(tmp_path / "graf1.jpg", quality=95)
    rc, msg = run([str(tmp_path / "graf1.jpg"), G6, "o1", "o2", "k1", "k2", "m", "log", "0", "0", "H", os.path.join(CFG, "classic.ini"),
                   os.path.join(CFG, "iters_one_view.ini")], tmp_path)
    assert "Cannot read image" not in msg and "Image1: 800x640" in msg
