"""-m gpu parity tests: HIP detector (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

FIELDS = ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type", "octave", "level", "r0", "c0")


def _assert_keys_equal(got, want):
    assert len(got) == len(want)
    for f in FIELDS:
        assert np.array_equal(got[f], want[f]), "field %s differs" % f


@pytest.mark.parametrize("w,h", [(64, 48), (135, 67), (257, 130), (33, 200), (640, 480)])
@pytest.mark.parametrize("sigma", [0.8, 1.2263, 1.5199, 2.4525, 4.9])
def test_gauss_blur_bitexact(gpu_ctx, w, h, sigma):
    img = synth.texture(w, h, seed=w * 7 + h)
    assert np.array_equal(gpu_ctx.gauss_blur(img, sigma), orc.gauss_blur(img, sigma))


@pytest.mark.parametrize("w,h", [(3, 3), (64, 48), (135, 67), (1000, 37)])
def test_hessian_response_bitexact(gpu_ctx, w, h):
    img = synth.texture(max(w, 8), max(h, 8), seed=3)[:h, :w].copy()
    assert np.array_equal(gpu_ctx.hessian_response(img, 2.56), orc.hessian_response(img, 2.56))


@pytest.mark.parametrize("w,h", [(64, 48), (135, 67), (17, 30), (240, 135), (31, 33), (2, 2)])
def test_resize_half_bitexact(gpu_ctx, w, h):
    img = synth.texture(max(w, 8), max(h, 8), seed=5)[:h, :w].copy()
    got, want = gpu_ctx.resize_half(img), orc.resize_half(img)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("w,h,seed", [(320, 240, 7), (515, 389, 11), (800, 640, 3)])
def test_pyramid_and_candidates(gpu_ctx, w, h, seed):
    img = synth.texture(w, h, seed=seed)
    got = gpu_ctx.detect_hessian_affine(img)
    pyr = orc.Pyramid(img)
    assert gpu_ctx.pyramid_octaves() == pyr.n_oct
    for o in range(pyr.n_oct):
        assert gpu_ctx.pyramid_dims(o) == pyr.dims(o)
        for lv in range(5):
            for kind in (0, 1):
                assert np.array_equal(gpu_ctx.pyramid_plane(0, o, lv, kind), pyr.plane(o, lv, kind)), (o, lv, kind)
    cand_want, _ = pyr.candidates()
    cand_got = gpu_ctx.pyramid_candidates(0)
    order = np.lexsort((cand_got["c0"], cand_got["r0"], cand_got["level"], cand_got["octave"]))
    cand_got = cand_got[order]
    assert len(cand_got) == len(cand_want)
    for f in cand_want.dtype.names:
        assert np.array_equal(cand_got[f], cand_want[f]), f
    _assert_keys_equal(got, orc.detect_hessian_affine(img))


def _alt_params(mod, det):
    par = mod.HessAffParams.harris() if det == "harris" else mod.HessAffParams.dog()
    if det == "iidog":
        par.iiDoGMode = 1
    if det == "dog_baumberg":
        par.doBaumberg = 1
    return par


@pytest.mark.parametrize("det", ["dog", "harris", "iidog", "dog_baumberg"])
@pytest.mark.parametrize("w,h,seed", [(320, 240, 7), (515, 389, 11)])
def test_dog_and_harris_detectors(gpu_ctx, pkg, det, w, h, seed):
    """[DoG] / [HarrisAffine]: the same scale-space detector on another response (pyramid.cpp:126-194, 256-278), final
    threshold un-squared (pyramid.h:55-56), point types 10/11 and 30/31 (pyramid.cpp:91-107): planes, candidates and
    keypoints equal the oracle's bit for bit."""
    img = synth.texture(w, h, seed=seed)
    got = gpu_ctx.detect_hessian_affine(img, params=_alt_params(pkg, det))
    po = _alt_params(orc, det)
    pyr = orc.Pyramid(img, po)
    assert gpu_ctx.pyramid_octaves() == pyr.n_oct
    for o in range(pyr.n_oct):
        for lv in range(5):
            for kind in (0, 1):
                assert np.array_equal(gpu_ctx.pyramid_plane(0, o, lv, kind), pyr.plane(o, lv, kind)), (o, lv, kind)
    cand_want, _ = pyr.candidates()
    cand_got = gpu_ctx.pyramid_candidates(0)
    order = np.lexsort((cand_got["c0"], cand_got["r0"], cand_got["level"], cand_got["octave"]))
    cand_got = cand_got[order]
    assert len(cand_got) == len(cand_want) and len(cand_want) > 20
    for f in cand_want.dtype.names:
        assert np.array_equal(cand_got[f], cand_want[f]), f
    want = orc.detect_hessian_affine(img, po)
    _assert_keys_equal(got, want)
    assert set(np.unique(want["sub_type"])) <= ({30, 31} if det == "harris" else {10, 11})


def test_iidog_is_refused_for_hessian_and_harris(gpu_ctx, pkg):
    img = synth.texture(64, 48, seed=1)
    for par in (pkg.HessAffParams.default(), pkg.HessAffParams.harris()):
        par.iiDoGMode = 1
        with pytest.raises(pkg.ModsError, match="iiDoGMode"):
            gpu_ctx.detect_hessian_affine(img, params=par)


def test_detect_graf(gpu_ctx):
    from PIL import Image
    import os
    p = os.path.join(os.path.dirname(__file__), "golden", "graf1.png")
    g = orc.grey_of_rgb(np.asarray(Image.open(p).convert("RGB")))
    _assert_keys_equal(gpu_ctx.detect_hessian_affine(g), orc.detect_hessian_affine(g))


def test_detect_1080p_and_batch(gpu_ctx, pkg):
    import torch
    a = synth.texture(1920, 1080, seed=21)
    b = synth.texture(1920, 1080, seed=22)
    wa, wb = orc.detect_hessian_affine(a), orc.detect_hessian_affine(b)
    assert len(wa) > 5000
    _assert_keys_equal(gpu_ctx.detect_hessian_affine(a), wa)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    ga, gb = gpu_ctx.detect_hessian_affine_dev(t.data_ptr(), 2, 1920, 1080)
    _assert_keys_equal(ga, wa)
    _assert_keys_equal(gb, wb)


def test_empty_image(gpu_ctx):
    img = np.full((100, 120), 77.0, np.float32)
    assert len(gpu_ctx.detect_hessian_affine(img)) == 0
    tiny = synth.texture(12, 12, seed=1)          # below 2*border+2: no octave at all
    assert len(gpu_ctx.detect_hessian_affine(tiny)) == 0


@pytest.mark.parametrize("w,h", [(13, 13), (12, 40), (14, 14), (30, 17), (31, 57), (129, 65), (801, 603)])
def test_small_and_odd_sizes_end_to_end(pkg, w, h):
    """Sizes around the octave cut-off (rows, cols > 12, pyramid.cpp:520), non-multiples of every tile size:
    detector + orientation + descriptor identical to the oracle (mostly empty lists; must not crash or overrun)."""
    import orc
    import synth
    img = synth.texture(w, h, seed=w * 1000 + h, blobs=max(4, w * h // 300))
    ctx = pkg.Context(0, max(w, 16), max(h, 16), 1)
    keys = ctx.detect_hessian_affine(img)
    want = orc.detect_hessian_affine(img)
    assert len(keys) == len(want)
    for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response"):
        assert np.array_equal(keys[f], want[f]), f
    wr, _ = orc.detect_describe(img)
    regs = ctx.orient_describe(img, keys)
    assert len(regs) == len(wr)
    if len(wr):
        assert np.array_equal(regs["desc"], wr["desc"]) and np.array_equal(regs["x"], wr["x"])
    ctx.close()


@pytest.mark.parametrize("smm", [9, 19, 23, 27, 31])
def test_baumberg_window_sizes(pkg, smm):
    """[HessianAffine] smmWindowSize other than the .ini's 19: baumberg_kernel keeps three product arrays per keypoint (the third
    over the window itself) while a keypoint's 32 lanes cover a window row (W <= 32: all of these), samples windows of up to 24
    columns with the next tile row's taps in flight and wider ones tile row by tile row (27, 31): the shapes are the oracle's."""
    import orc
    import synth
    img = synth.texture(640, 480, seed=77)
    gp, op = pkg.HessAffParams.default(), orc.HessAffParams.default()
    gp.smmWindowSize = smm; op.smmWindowSize = smm
    ctx = pkg.Context(0, 640, 480, 1)
    got, want = ctx.detect_hessian_affine(img, params=gp), orc.detect_hessian_affine(img, params=op)
    assert len(got) == len(want) and len(got) > 200
    for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response"):
        assert np.array_equal(got[f], want[f]), f
    ctx.close()


def test_baumberg_work_counters(pkg):
    """mods_baumberg_stats: keypoints that entered the iteration = the points the pyramid accepted, 1 .. max_iter iterations each."""
    import synth
    img = synth.texture(640, 480, seed=78)
    ctx = pkg.Context(0, 640, 480, 1)
    ctx.baumberg_stats_enable(True)
    keys = ctx.detect_hessian_affine(img)
    kp, it = ctx.baumberg_stats(0)
    assert kp >= len(keys) > 200 and kp <= it <= 16 * kp
    keys2 = ctx.detect_hessian_affine(img)
    kp2, it2 = ctx.baumberg_stats(0)
    assert (kp2, it2) == (2 * kp, 2 * it) and np.array_equal(keys2["a11"], keys["a11"])
    ctx.baumberg_stats_enable(False)
    with pytest.raises(pkg.ModsError):
        ctx.baumberg_stats(0)
    ctx.close()


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_keypoint_selection_modes(pkg, mode):
    """[HessianAffine] mode = RelativeTh / FixedRegNumber / RelativeRegNumber / NotLessThanRegions
    (AffineDetector::prepareKeysForExport, scale-space-detector.hpp:126-198): in these modes the detector's thresholds are 0
    (pyramid.h:58-59), every 3x3x3 extremum is localised and adapted, and the sorted list is cut."""
    import orc
    import synth
    w, h = 400, 300
    img = synth.texture(w, h, seed=11)
    ctx = pkg.Context(0, w, h, 1)
    for reg, rel_th, rel_n in ((300, 0.02, 0.25), (5000, 0.5, 1.0), (0, 0.0, 0.0)):
        po, pg = orc.HessAffParams.default(), pkg.HessAffParams.default()
        for p in (po, pg):
            p.mode, p.relativeThreshold, p.regionsNumber, p.relativeRegionsNumber = mode, rel_th, reg, rel_n
        want = orc.detect_hessian_affine(img, po)
        keys = ctx.detect_hessian_affine(img, pg)
        assert len(keys) == len(want), (mode, reg, rel_th, rel_n)
        for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type"):
            assert np.array_equal(keys[f], want[f]), f
    # the selection applies to whole-pair calls too (counts feed the describe stage on the device)
    po, pg = orc.HessAffParams.default(), pkg.HessAffParams.default()
    for p in (po, pg):
        p.mode, p.relativeThreshold, p.regionsNumber, p.relativeRegionsNumber = mode, 0.02, 300, 0.25
    wr, nd = orc.detect_describe(img, po)
    import torch
    t = torch.from_numpy(img).cuda()
    got_nd, got_nr = ctx.detect_describe_dev(t.data_ptr(), 1, w, h, pg)
    assert got_nd[0] == nd and got_nr[0] == len(wr)
    assert np.array_equal(ctx.regions_fetch(0)["desc"], wr["desc"])
    bad = pkg.HessAffParams.default()
    bad.mode = 2                      # regionsNumber = -1: the reference would resize its list to (size_t)-1
    with pytest.raises(Exception):
        ctx.detect_hessian_affine(img, bad)
    ctx.close()


@pytest.mark.parametrize("name,w,h,seed,sfi", [("HessianAffine", 640, 480, 11, 0), ("HessianAffine", 515, 389, 3, 1),
                                               ("DoG", 400, 300, 5, 0), ("HarrisAffine", 400, 300, 6, 0)])
def test_hessian_form_of_the_baumberg_iteration(pkg, name, w, h, seed, sfi):
    """affBmbrgMethod = 1 (io_mods.cpp:193, affine.cpp:92-128): 3x3 samples at s * 0.5, SVD of the finite-difference Hessian,
    Ap <- Au Ap Au - one lane per point (baumberg_hessian_kernel) against the oracle's restatement, bit for bit; with
    sampleFromImage, and behind the DoG / Harris responses with doBaumberg switched on."""
    import orc
    img = synth.texture(w, h, seed=seed)
    make = {"HessianAffine": "default", "DoG": "dog", "HarrisAffine": "harris"}[name]
    par, opar = getattr(pkg.HessAffParams, make)(), getattr(orc.HessAffParams, make)()
    for q in (par, opar):
        q.affBmbrgMethod = 1; q.doBaumberg = 1; q.sampleFromImage = sfi; q.mode = 0
    ctx = pkg.Context(0, w, h, 1)
    got = ctx.detect_hessian_affine(img, par)
    want = orc.detect_hessian_affine(img, opar)
    par.affBmbrgMethod = 0
    base = ctx.detect_hessian_affine(img, par)
    assert len(want) > 100
    _assert_keys_equal(got, want)
    assert len(base) != len(got) or not np.array_equal(base["a21"], got["a21"])
    par.affBmbrgMethod = 2
    with pytest.raises(Exception, match="affBmbrgMethod"):
        ctx.detect_hessian_affine(img, par)
    ctx.close()


def test_hessian_form_batch_and_describe(pkg):
    """The same through the batched entry point the pipeline uses (two images per launch, orientation + RootSIFT behind it)."""
    import orc
    import torch
    w, h = 480, 360
    imgs = [synth.texture(w, h, seed=21), synth.texture(w, h, seed=22)]
    par, opar = pkg.HessAffParams.default(), orc.HessAffParams.default()
    par.affBmbrgMethod = 1; opar.affBmbrgMethod = 1
    ctx = pkg.Context(0, w, h, 2)
    t = torch.from_numpy(np.stack(imgs)).cuda()
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), 2, w, h, det=par)
    for b in range(2):
        want, ndet = orc.detect_describe(imgs[b], opar)
        got = ctx.regions_fetch(b)
        assert nd[b] == ndet and nr[b] == len(want) > 100
        for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type", "desc"):
            assert np.array_equal(got[f], want[f]), (b, f)
    ctx.close()


def test_sample_from_image(pkg):
    """[HessianAffine] sampleFromImage = 1 (io_mods.cpp:184): the Baumberg iteration samples the input image at pixel distance 1
    (scale-space-detector.hpp:47-55) instead of the blur level; keypoints equal the oracle's bit for bit and differ from the
    default's."""
    import orc
    img = synth.texture(640, 480, seed=11)
    par, opar = pkg.HessAffParams.default(), orc.HessAffParams.default()
    par.sampleFromImage = 1; opar.sampleFromImage = 1
    ctx = pkg.Context(0, 640, 480, 1)
    got = ctx.detect_hessian_affine(img, par)
    want = orc.detect_hessian_affine(img, opar)
    base = ctx.detect_hessian_affine(img)
    assert len(got) == len(want) > 300
    for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type"):
        assert np.array_equal(got[f], want[f]), f
    assert len(base) != len(got) or not np.array_equal(base["a11"], got["a11"])
    ctx.close()
