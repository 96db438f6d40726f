"""-m gpu parity tests of the MSER detector (DetectMSERs behind DetectAffineRegions, detectors/mser/extrema/extrema.cpp:196-295,
imagerepresentation.cpp:780-783): keypoints bit-equal to the CPU oracle (oracle/mser.cpp) in every selection mode, through the
describe stage, on a batch and on synthesised views."""
import math
import os

import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KEY_FIELDS = ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type", "octave", "level", "r0", "c0")


def graf(name):
    from PIL import Image
    return orc.grey_of_rgb(np.asarray(Image.open(os.path.join(HERE, "golden", name)).convert("RGB")))


def same_keys(got, want):
    assert len(got) == len(want), (len(got), len(want))
    for f in KEY_FIELDS:
        assert np.array_equal(got[f], want[f]), f


@pytest.mark.parametrize("name", ["graf1.png", "graf6.png"])
def test_mser_keys_graf(pkg, name):
    img = graf(name)
    ctx = pkg.Context(0, img.shape[1], img.shape[0], 1)
    want = orc.detect_hessian_affine(img, orc.HessAffParams.mser())
    got = ctx.detect_hessian_affine(img, pkg.HessAffParams.mser())
    assert len(want) > 300 and {20, 21} == set(want["sub_type"].tolist())
    same_keys(got, want)
    ctx.close()


@pytest.mark.parametrize("mode,kw", [(1, dict(rel_threshold=0.2)), (2, dict(reg_number=300)), (3, dict(rel_reg_number=0.25)),
                                     (4, dict(reg_number=5000)), (4, dict(reg_number=50))])
def test_mser_selection_modes(pkg, mode, kw):
    """prepareKeysForExport (extrema.cpp:31-90): margin bound 1, std::sort by margin (its own order inside equal margins), cut."""
    img = synth.texture(640, 480, seed=5)
    ctx = pkg.Context(0, 640, 480, 1)
    want = orc.detect_hessian_affine(img, orc.HessAffParams.mser(mode=mode, **kw))
    got = ctx.detect_hessian_affine(img, pkg.HessAffParams.mser(mode=mode, **kw))
    assert len(want) > 20
    assert np.all(np.diff(np.abs(want["response"])) <= 0)
    same_keys(got, want)
    ctx.close()


def test_mser_batch_and_parameters(pkg):
    """two images per launch (four growth jobs), other [MSER] keys, a strided batch"""
    import torch
    a, b, _ = synth.pair(800, 600, seed=21)
    ctx = pkg.Context(0, 800, 600, 2)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    for kw in (dict(), dict(min_margin=5, max_area=0.01, min_size=60), dict(min_margin=12, max_area=0.2, min_size=10)):
        got = ctx.detect_hessian_affine_dev(t.data_ptr(), 2, 800, 600, pkg.HessAffParams.mser(**kw))
        for im, g in zip((a, b), got):
            same_keys(g, orc.detect_hessian_affine(im, orc.HessAffParams.mser(**kw)))
    ctx.close()


def test_mser_1080p_flat_areas(pkg):
    """saturated plateaus (many pixels of one grey level entering together: the within-level order decides which label slot
    survives) and a 1080p frame"""
    img = synth.texture(1920, 1080, seed=3)
    img = np.clip((img - 128.0) * 3.0 + 128.0, 0, 255).astype(np.float32)      # large areas at 0 and 255
    ctx = pkg.Context(0, 1920, 1080, 1)
    want = orc.detect_hessian_affine(img, orc.HessAffParams.mser())
    got = ctx.detect_hessian_affine(img, pkg.HessAffParams.mser())
    assert len(want) > 1000
    same_keys(got, want)
    ctx.close()


def test_mser_detect_describe(pkg):
    """MSER keys through orientation + RootSIFT (the common path of SynthDetectDescribeKeypoints)"""
    import torch
    img = graf("graf1.png")
    h, w = img.shape
    ctx = pkg.Context(0, w, h, 1)
    t = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), 1, w, h, det=pkg.HessAffParams.mser())
    want, nd0 = orc.detect_describe(img, orc.HessAffParams.mser())
    got = ctx.regions_fetch(0)
    assert nd == [nd0] and nr == [len(want)] and len(want) > 300
    for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["desc"], want["desc"])
    ctx.close()


@pytest.mark.parametrize("tilt,phi,zoom", [(1.0, 0.0, 0.25), (1.0, 0.0, 0.125), (3.0, 0.0, 1.0), (6.0, math.pi / 3, 0.25), (3.0, 2.0, 0.25)])
def test_mser_views(pkg, tilt, phi, zoom):
    """the views of [MSER0] / [MSER1] of iters_MODS.ini (ScaleSet 1, 0.25, 0.125; TiltSet 1, 3, 6): non-integer pixel values are
    truncated to 8 bits, regions reprojected to the original frame"""
    import torch
    img = graf("graf1.png")
    h, w = img.shape
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    t = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    want_px, g0 = orc.synth_view(img, tilt, phi, zoom, 0.8, 1)
    want, want_det, nd0 = orc.detect_describe_view(want_px, np.array(g0.H), w, h, params=orc.HessAffParams.mser())
    g, nd, nr = ctx.detect_describe_view_dev(t.data_ptr(), w, h, tilt, phi, zoom, 0.8, 1, det=pkg.HessAffParams.mser())
    assert np.array_equal(ctx.view_pixels(g).view(np.uint32), want_px.view(np.uint32))
    assert nd == nd0 and nr == len(want) and nd0 > 10
    got = ctx.regions_fetch(0)
    for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type"):
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["desc"], want["desc"])
    ctx.close()


def test_mser_view_scales_the_region_number(pkg):
    """regionsNumber of a view is scaled by 2 * zoom / tilt when tilt > 2 or zoom < 0.5 (extrema.cpp:201-202)"""
    import torch
    img = graf("graf6.png")
    h, w = img.shape
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    t = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    tilt, zoom = 3.0, 1.0
    px, g0 = orc.synth_view(img, tilt, 0.0, zoom, 0.8, 1)
    want = orc.detect_mser_view(px, orc.HessAffParams.mser(mode=2, reg_number=90), g0.tilt, g0.zoom)
    assert len(want) == int(math.floor(zoom * 2.0 * 90 / tilt))
    g, nd, nr = ctx.detect_describe_view_dev(t.data_ptr(), w, h, tilt, 0.0, zoom, 0.8, 1, det=pkg.HessAffParams.mser(mode=2, reg_number=90))
    assert nd == len(want)
    ctx.close()


def test_ladder_with_mser_steps(pkg):
    """The shape of build/iters_MODS.ini: [MSER0] (scales 1, 0.25, 0.125), [MSER1] (tilts 1, 3, 6 at scales 1, 0.25), then a
    HessianAffine step, Half* descriptors named (orientation modulo pi) - every step against the oracle chain: MSER and
    HessianAffine keep separate banks and tentative lists, joined in name order (HessianAffine before MSER)."""
    import torch
    import pipeline_oracle as po
    import refdeg
    from test_gpu_views import _hard_pair
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    w, h = 480, 360
    a, b, Htrue = _hard_pair(w, h, seed=29)
    mser_steps = [((1,), 360.0, (1, 0.25, 0.125), 0.8, 0.85), ((1, 3, 6), 360.0, (1, 0.25), 0.8, 0.8), None]
    dets = [dict(params=orc.HessAffParams.default(), steps=[None, None, ((1, 2, 4), 360.0)], ratio=0.8, half_orientation=True),
            dict(params=orc.HessAffParams.mser(), steps=mser_steps, half_orientation=True)]
    want = po.match_ladder(a, b, None, seed_time=31, min_matches=100000, detectors=dets)
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    reps1, reps2 = [pkg.ImgRep(ctx), pkg.ImgRep(ctx)], [pkg.ImgRep(ctx), pkg.ImgRep(ctx)]
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(31)
    L = pkg.LadderStep.make
    det_steps = [[None, None, L((1, 2, 4), 360.0, half_orientation=1)],
                 [L((1,), 360.0, scales=(1, 0.25, 0.125), init_sigma=0.8, fginn=0.85, half_orientation=1),
                  L((1, 3, 6), 360.0, scales=(1, 0.25), init_sigma=0.8, fginn=0.8, half_orientation=1), None]]
    res, m = pkg.match_ladder_dets_dev(ctx, t.data_ptr(), w, h, det_steps, [pkg.HessAffParams.default(), pkg.HessAffParams.mser()],
                                       reps1, reps2, min_matches=100000, max_matches=100000)
    pkg.ransac_pin_seed(-1)
    assert res.steps_done == want["steps_done"] == 3 and res.n_views == want["n_views"]
    assert list(res.n_described) == want["n_described"]
    assert len(reps1[1]) > 100 and len(reps2[1]) > 50 and len(reps1[0]) > 100
    assert res.n_tentatives == want["n_tentatives"] and res.n_unique == want["n_unique"]
    assert [res.ransac_samples, res.ransac_lo, res.ransac_rejects] == want["stats"]
    assert res.n_inliers == want["n_inliers"]
    assert np.array_equal(m, want["u6"][want["mask"]][:, [0, 1, 3, 4]])
    for r in reps1 + reps2:
        r.close()
    ctx.close()


def test_mser_degenerate_inputs(pkg):
    """a constant image (one component, never stable within max_area), an image smaller than min_size, a two-level image"""
    ctx = pkg.Context(0, 256, 256, 1)
    for img in (np.full((64, 96), 77.0, np.float32), np.full((5, 5), 10.0, np.float32),
                np.kron(np.indices((8, 8)).sum(0) % 2, np.ones((16, 16))).astype(np.float32) * 200.0):
        want = orc.detect_hessian_affine(img, orc.HessAffParams.mser())
        got = ctx.detect_hessian_affine(img, pkg.HessAffParams.mser())
        same_keys(got, want)
    assert len(want) == 64          # the checkerboard: 32 dark + 32 bright squares
    # parameters the reference would run into undefined behaviour with are refused
    bad = pkg.HessAffParams.mser(min_margin=0)
    with pytest.raises(pkg.ModsError):
        ctx.detect_hessian_affine(img, bad)
    ctx.close()


def test_mser_pair_and_pipeline(pkg):
    """MSER as the detector of mods_match_pair_dev and of the pair pipeline (growth on the pipeline's worker threads): counts,
    RANSAC statistics and the inlier set against the oracle chain"""
    import torch
    import pipeline_oracle as po
    import refdeg
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    w, h = 800, 600
    a, b, Htrue = synth.pair(w, h, seed=31)
    regs = tuple(orc.detect_describe(im, orc.HessAffParams.mser()) for im in (a, b))
    want = po.match_pair(a, b, seed_time=55, regions=regs)
    par = pkg.PairParams.default()
    par.det = pkg.HessAffParams.mser()
    ctx = pkg.Context(0, w, h, 2)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(55)
    res, m = pkg.match_pair_dev(ctx, t.data_ptr(), w, h, par, max_matches=100000)
    assert list(res.n_detected) == want["n_detected"] and list(res.n_described) == want["n_described"]
    assert res.n_tentatives == want["n_tentatives"] and res.n_unique == want["n_unique"]
    assert [res.ransac_samples, res.ransac_lo, res.ransac_rejects] == want["stats"]
    assert res.n_inliers == want["n_inliers"] > 30
    assert np.array_equal(m, want["u6"][want["mask"]][:, [0, 1, 3, 4]])
    ctx.close()
    pipe = pkg.Pipeline(0, w, h, par, 2, 2, 2)
    for i in range(4):
        pipe.submit(t.data_ptr(), i)
    for i in range(4):
        r, tag = pipe.next()
        assert tag == i and r.n_inliers == want["n_inliers"] and r.n_tentatives == want["n_tentatives"]
    pipe.close()
    pkg.ransac_pin_seed(-1)


def test_mser_random_plateau_images(pkg):
    """the random small images of tests/test_cpu_mser.py (few grey levels, noise, blocks) through the whole detector: nested regions
    of every depth, one-pixel runs, regions touching the frame"""
    from test_cpu_mser import _fuzz_image
    rng = np.random.default_rng(777)
    ctx = pkg.Context(0, 128, 128, 1)
    n = 0
    for it in range(60):
        img = _fuzz_image(rng, it)
        kw = dict(min_size=int(rng.choice([1, 5, 30])), min_margin=float(rng.choice([1, 3, 8])), max_area=float(rng.choice([0.05, 0.9])))
        want = orc.detect_hessian_affine(img, orc.HessAffParams.mser(**kw))
        got = ctx.detect_hessian_affine(img, pkg.HessAffParams.mser(**kw))
        same_keys(got, want)
        n += len(want)
    assert n > 2000
    ctx.close()
