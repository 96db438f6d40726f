"""CPU tests of the MSER row: the oracle (oracle/mser.cpp, a restatement of detectors/mser/extrema/ - PARITY UNPINNED, the
reference's MSER cannot be compiled without OpenCV headers and its tree holds no MSER output) against the properties that
define its result, and the product's host half (csrc/mser_host.hpp: the grey-level growth and the merge tree the kernels walk)
against the oracle - no device needed."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def graf(name):
    from PIL import Image
    return orc.grey_of_rgb(np.asarray(Image.open(os.path.join(HERE, "golden", name)).convert("RGB")))


def _u8(img):
    return np.ascontiguousarray(img.astype(np.int32).astype(np.uint8))


@pytest.mark.parametrize("name", ["graf1.png", "graf6.png"])
def test_oracle_regions_are_the_components_of_their_threshold_sets(name):
    """Every region (seed, threshold) the growth selects and the flood fill outlines is the 4-connected component of
    {pixel <= threshold} around the seed (scipy.ndimage.label as an independent labelling), its run list covers exactly that
    set in (line, column) order, the area equals the growth's pixel counter, and the margin obeys the stability bound."""
    from scipy import ndimage
    img = graf(name)
    info, runs, ell = orc.mser_regions(img, orc.HessAffParams.mser())
    assert len(info) > 300 and (info[:, 8] == 0).sum() > 50 and (info[:, 8] == 1).sum() > 50
    u8 = _u8(img)
    cache = {}
    for i in range(len(info)):
        th, margin, mn, mx, area, border, sx, sy, pol, nr = (int(v) for v in info[i])
        if (pol, th) not in cache:
            cache[(pol, th)] = ndimage.label((255 - u8 if pol else u8) <= th)[0]
        lab = cache[(pol, th)]
        comp = lab == lab[sy, sx]
        got = np.zeros_like(comp)
        r = runs[i]
        assert len(r) == nr and np.all(r[:, 2] >= r[:, 1])
        key = r[:, 0].astype(np.int64) * 100000 + r[:, 1]
        assert np.all(np.diff(key) > 0)                                   # sorted, disjoint
        for line, c1, c2 in r:
            got[line, c1:c2 + 1] = True
        assert np.array_equal(got, comp), i
        assert area == comp.sum() and 30 < area <= int(img.shape[0] * img.shape[1] * 0.05)
        assert margin > 8 and mn <= th < mx
        # moments of the pixel set, integrated over unit squares (RLE2Ellipse, libExtrema.cpp:117-160)
        ys, xs = np.nonzero(comp)
        assert abs(ell[i, 0] - (xs.mean() + 0.5)) < 1e-9 and abs(ell[i, 1] - (ys.mean() + 0.5)) < 1e-9
        assert abs(ell[i, 2] - (xs.var() + 1 / 12.0)) < 1e-6 * max(1.0, ell[i, 2])
        assert abs(ell[i, 4] - (ys.var() + 1 / 12.0)) < 1e-6 * max(1.0, ell[i, 4])
        assert abs(ell[i, 3] - np.mean((xs - xs.mean()) * (ys - ys.mean()))) < 1e-6 * max(1.0, abs(ell[i, 3]))


def test_oracle_keys_and_selection_modes():
    img = synth.texture(480, 360, seed=9)
    fixed = orc.detect_hessian_affine(img, orc.HessAffParams.mser())
    assert len(fixed) > 20 and set(fixed["sub_type"].tolist()) <= {20, 21}
    plus = np.nonzero(fixed["sub_type"] == 21)[0]
    assert len(plus) and plus.max() == len(plus) - 1                    # MSER+ first (extrema.cpp:246-290)
    assert np.all(fixed["a12"] == 0) and np.allclose(fixed["a11"] * fixed["a22"], 1.0)      # rectifyTransformation
    allk = orc.detect_hessian_affine(img, orc.HessAffParams.mser(mode=3, rel_reg_number=1.0))
    assert len(allk) > len(fixed) and np.all(np.diff(allk["response"]) <= 0) and allk["response"].min() >= 2
    n = orc.detect_hessian_affine(img, orc.HessAffParams.mser(mode=2, reg_number=40))
    assert len(n) == 40
    for f in ("x", "y", "response"):
        assert np.array_equal(n[f], allk[f][:40])
    rel = orc.detect_hessian_affine(img, orc.HessAffParams.mser(mode=1, rel_threshold=0.5))
    assert len(rel) == (allk["response"] > 0.5 * allk["response"][0]).sum()
    view = orc.detect_mser_view(img, orc.HessAffParams.mser(mode=2, reg_number=40), 3.0, 1.0)
    assert len(view) == 26                                              # floor(1.0 * 2 * 40 / 3), extrema.cpp:201-202


def _grow(pkg, u8, pol, min_size=30, max_area=0.05, min_margin=8.0, tree=False):
    h, w = u8.shape
    out = np.zeros((1 << 16, 5), np.int32)
    tr = np.zeros(((w + 2) * (h + 2), 3), np.int32) if tree else None
    fn = pkg.lib().mods_test_mser_grow
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    n = fn(u8.ctypes.data, w, h, min_size, max_area, min_margin, pol, out.ctypes.data, 1 << 16, tr.ctypes.data if tree else None)
    assert 0 <= n <= 1 << 16
    return out[:n].copy(), tr


@pytest.mark.parametrize("case", ["graf1", "flat", "small_params", "margin1"])
def test_host_growth_equals_oracle(pkg, case):
    """csrc/mser_host.hpp (the product's host half) against the oracle: the same stable (seed, threshold, margin, area) rows in
    the same order for both polarities - including plateaus of one grey level, where the order inside the level decides."""
    kw = {}
    if case == "graf1":
        img = graf("graf1.png")
    elif case == "flat":
        img = np.clip((synth.texture(640, 480, seed=3) - 128.0) * 3.0 + 128.0, 0, 255).astype(np.float32)
    elif case == "small_params":
        img = synth.texture(320, 240, seed=4)
        kw = dict(min_size=5, max_area=0.3, min_margin=3)
    else:
        img = synth.texture(320, 240, seed=6)
        kw = dict(min_margin=1)
    p = orc.HessAffParams.mser(**kw)
    info, runs, ell = orc.mser_regions(img, p)
    u8 = _u8(img)
    for pol in (0, 1):
        want = info[info[:, 8] == pol]
        got, _ = _grow(pkg, u8, pol, p.mserMinSize, p.mserMaxArea, p.mserMinMargin)
        assert len(got) == len(want) and len(want) > 5
        assert np.array_equal(got, want[:, [6, 7, 0, 1, 4]])


def test_merge_tree_reproduces_the_regions(pkg):
    """The tree handed to the kernels: walking pix_slot / tpar / tlev with merge levels <= t from every pixel of level <= t
    ends in the region's slot exactly for the pixels of the oracle's run list (the membership rule of mser_inner_kernel,
    restated in numpy)."""
    img = synth.texture(200, 150, seed=8)
    u8 = _u8(img)
    h, w = u8.shape
    cols = w + 2
    info, runs, ell = orc.mser_regions(img, orc.HessAffParams.mser(min_margin=5))
    for pol in (0, 1):
        want = np.nonzero(info[:, 8] == pol)[0]
        got, tr = _grow(pkg, u8, pol, 30, 0.05, 5.0, tree=True)
        assert len(got) == len(want) > 3
        lev = (255 - u8 if pol else u8).astype(np.int32)
        ps, tp, tl = tr[:, 0], tr[:, 1], tr[:, 2]
        for k, i in enumerate(want[:12]):
            sx, sy, th = int(got[k, 0]), int(got[k, 1]), int(got[k, 2])
            slot = (sy + 1) * cols + sx + 1
            member = np.zeros((h, w), bool)
            for y, x in zip(*np.nonzero(lev <= th)):
                s = ps[(y + 1) * cols + x + 1]
                while tp[s] != 0x7fffffff and tl[s] <= th:
                    s = tp[s]
                member[y, x] = s == slot
            exp = np.zeros((h, w), bool)
            for line, c1, c2 in runs[i]:
                exp[line, c1:c2 + 1] = True
            assert np.array_equal(member, exp), (pol, k)


def test_library_exports_the_mser_symbols(pkg):
    lib = pkg.lib()
    assert hasattr(lib, "mods_test_mser_grow")
    assert C.sizeof(pkg.HessAffParams) == 88 == C.sizeof(orc.HessAffParams)
    assert pkg.HessAffParams.mser().detectorType == 3


def _fuzz_image(rng, it):
    w, h = int(rng.integers(8, 90)), int(rng.integers(8, 70))
    kind = it % 4
    if kind == 0:
        img = rng.integers(0, 4, (h, w)) * 60.0                      # four grey levels: plateaus everywhere
    elif kind == 1:
        img = rng.integers(0, 256, (h, w)).astype(float)
    elif kind == 2:
        img = np.kron(rng.integers(0, 6, (h // 6 + 1, w // 6 + 1)) * 40.0, np.ones((6, 6)))[:h, :w] + rng.integers(0, 2, (h, w))
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        img = (128 + 100 * np.sin(xx / 5.0) * np.cos(yy / 7.0) + rng.integers(0, 3, (h, w))).clip(0, 255)
    return np.ascontiguousarray(img, np.float32)


def test_host_growth_equals_oracle_on_random_plateau_images(pkg):
    """160 small images of four kinds (a few grey levels, white noise, blocky, smooth + noise) with min_size down to 1 and
    max_area up to 0.9: the order inside a grey level, min-region labels absorbing regions, recycled region records - every
    stable (seed, threshold, margin, area) row of both polarities equal to the oracle's."""
    rng = np.random.default_rng(12345)
    regions = 0
    for it in range(160):
        img = _fuzz_image(rng, it)
        ms, mm, ma = int(rng.choice([1, 2, 5, 12, 30])), float(rng.choice([1, 2, 3, 8])), float(rng.choice([0.05, 0.3, 0.9]))
        info, runs, ell = orc.mser_regions(img, orc.HessAffParams.mser(min_margin=mm, max_area=ma, min_size=ms))
        u8 = _u8(img)
        for pol in (0, 1):
            got, _ = _grow(pkg, u8, pol, ms, ma, mm)
            assert np.array_equal(got, info[info[:, 8] == pol][:, [6, 7, 0, 1, 4]]), (it, pol)
            regions += len(got)
    assert regions > 5000
