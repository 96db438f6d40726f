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
    rc, err = run(base + ["1", "H", os.path.join(CFG, "classic.ini"), os.path.join(CFG, "iters_one_view.ini")], tmp_path)
    assert rc == 1 and "wrong correspondence verification type" in err
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
    assert "detector MSER is outside this build" in err and "descriptor HalfRootSIFT is outside this build" in err
    assert "Image1: 800x640, Image2: 800x640" in err
    assert "no MI355X / HIP device available" in err and "no CPU path" in err
    assert not os.path.exists(tmp_path / "m")
