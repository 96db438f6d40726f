"""-m gpu parity tests of the view-synthesis path (GenerateSynthImageCorr + the per-view
detect/orient/reproject/describe chain) against the CPU oracle: bit-exact pixels, identical region lists."""
import math

import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

VIEWS = [(2.0, 0.0), (4.0, math.pi / 2), (6.0, math.pi / 3), (8.0, 3 * math.pi / 4), (-4.0, 0.0), (2.0, 2.5), (1.0, 0.0),
         (3.0, 0.1)]


@pytest.fixture(scope="module")
def img():
    return synth.texture(640, 480, seed=11)


def test_view_geometry_matches_oracle(pkg):
    for w, h in ((640, 480), (1920, 1080), (801, 333)):
        for tilt, phi in VIEWS:
            for zoom in (1.0, 0.5):
                a = pkg.view_geometry(w, h, tilt, phi, zoom, 0.2)
                b = orc.view_geometry(w, h, tilt, phi, zoom, 0.2)
                for f, _ in a._fields_:
                    va, vb = getattr(a, f), getattr(b, f)
                    if hasattr(va, "__len__"):
                        assert list(va) == list(vb), (f, tilt, phi, zoom)
                    else:
                        assert va == vb, (f, tilt, phi, zoom, va, vb)


def test_warp_affine_bit_exact(pkg, img):
    w, h = img.shape[1], img.shape[0]
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    rng = np.random.default_rng(0)
    for k in range(8):
        ang = rng.uniform(0, math.pi)
        sc = rng.uniform(0.2, 1.2)
        M = np.array([math.cos(ang) * sc, math.sin(ang), rng.uniform(-50, 300), -math.sin(ang) * sc, math.cos(ang), rng.uniform(-50, 400)])
        dw, dh = int(rng.integers(50, 900)), int(rng.integers(50, 900))
        want = orc.warp_affine(img, M, dw, dh)
        got = ctx.warp_affine(img, M, dw, dh)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), k
    ctx.close()


def test_blur_xy_bit_exact(pkg, img):
    ctx = pkg.Context(0, img.shape[1], img.shape[0], 1)
    for kx, ky, sx, sy in ((3, 3, 0.2, 0.1), (5, 3, 0.8, 0.1), (3, 5, 0.1, 0.6), (5, 5, 0.7, 0.8), (7, 3, 1.0, 0.1), (9, 9, 1.4, 1.4)):
        want = orc.gauss_blur_xy(img, kx, ky, sx, sy)
        got = ctx.gauss_blur_xy(img, kx, ky, sx, sy)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (kx, ky)
    tiny = img[:7, :5].copy()       # reflection wraps more than once on tiny images
    assert np.array_equal(ctx.gauss_blur_xy(tiny, 5, 5, 0.8, 0.8), orc.gauss_blur_xy(tiny, 5, 5, 0.8, 0.8))
    ctx.close()


@pytest.mark.parametrize("tilt,phi", VIEWS)
def test_view_detect_describe_matches_oracle(pkg, img, tilt, phi):
    import torch
    w, h = img.shape[1], img.shape[0]
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    t = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    want_px, g0 = orc.synth_view(img, tilt, phi, 1.0, 0.2, 1)
    want, want_det, nd0 = orc.detect_describe_view(want_px, np.array(g0.H), w, h)
    g, nd, nr = ctx.detect_describe_view_dev(t.data_ptr(), w, h, tilt, phi, 1.0, 0.2, 1)
    got_px = ctx.view_pixels(g)
    assert got_px.shape == want_px.shape
    assert np.array_equal(got_px.view(np.uint32), want_px.view(np.uint32))
    assert nd == nd0 and nr == len(want)
    got = ctx.regions_fetch(0)
    for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["desc"], want["desc"])
    if abs(tilt) > 1.5:
        assert nr > 20
        # regions are expressed in the original frame
        assert got["x"].min() > 0 and got["x"].max() < w and got["y"].min() > 0 and got["y"].max() < h
    ctx.close()
