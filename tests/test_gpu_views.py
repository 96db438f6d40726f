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


def _hard_pair(w, h, seed):
    """Image 2 = image 1 seen under a strong tilt (an affine map with anisotropy ~4.5 at 35 degrees)."""
    a = synth.texture(w, h, seed)
    ang = np.deg2rad(35.0)
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    A = R @ np.diag([1.0, 1.0 / 4.5]) @ R.T @ np.array([[np.cos(0.3), -np.sin(0.3)], [np.sin(0.3), np.cos(0.3)]])
    c = np.array([w / 2.0, h / 2.0])
    H = np.eye(3)
    H[:2, :2] = A
    H[:2, 2] = c - A @ c
    return a, synth.warp(a, H, seed=seed), H


def test_view_schedule_counts(pkg):
    hist = []
    steps = pkg.iters_mods_steps()
    assert len(pkg.view_schedule(steps[0], hist)) == 11      # 11 + 20 new views per image
    assert len(pkg.view_schedule(steps[1], hist)) == 20
    import pipeline_oracle as po
    h2 = []
    assert po.view_schedule((1, 2, 4, 6, 8), 360.0, h2) == hist[:11]
    assert po.view_schedule((1, 2, 4, 6, 8), 120.0, h2) == hist[11:]


@pytest.mark.parametrize("w,h,steps_spec", [
    (480, 360, [((1,), 360.0), ((1, 2, 4), 360.0), ((1, 2, 4), 120.0)]),
    # the HessianAffine steps of build/iters_MODS.ini: TiltSet 1 2 4 6 8, Phi 360 then 120 (11 + 20 new views per image)
    (640, 480, [((1, 2, 4, 6, 8), 360.0), ((1, 2, 4, 6, 8), 120.0)])])
def test_ladder_matches_oracle(pkg, w, h, steps_spec):
    """The step loop on a pair that the identity view alone cannot match: same views, same accumulated regions,
    same tentatives, same inlier set as the CPU oracle chain."""
    import torch
    import pipeline_oracle as po
    import refdeg
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    a, b, Htrue = _hard_pair(w, h, seed=21)
    want = po.match_ladder(a, b, steps_spec, seed_time=31)
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    rep1, rep2 = pkg.ImgRep(ctx), pkg.ImgRep(ctx)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(31)
    steps = [pkg.LadderStep.make(tl, ph) for tl, ph in steps_spec]
    res, m = pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, steps, rep1, rep2, max_matches=100000)
    assert res.steps_done == want["steps_done"]
    assert res.steps_done >= 2 or len(steps_spec) == 2                       # first case: the identity view alone must fail
    assert res.n_views == want["n_views"]
    assert list(res.n_described) == want["n_described"] == [len(rep1), len(rep2)]
    ra, rb = rep1.fetch(), rep2.fetch()
    for got, exp in ((ra, want["regions"][0]), (rb, want["regions"][1])):
        for f in ("x", "y", "s", "a11", "a12", "a21", "a22"):
            assert np.array_equal(got[f], exp[f]), f
        assert np.array_equal(got["desc"], exp["desc"])
    assert res.n_tentatives == want["n_tentatives"] and res.n_unique == want["n_unique"]
    assert [res.ransac_samples, res.ransac_lo, res.ransac_rejects] == want["stats"]
    assert res.n_inliers == want["n_inliers"] >= 15
    assert np.array_equal(m, want["u6"][want["mask"]][:, [0, 1, 3, 4]])
    # the verified matches follow the generating map (a handful of inliers: compare transfer, not coefficients)
    p = np.c_[m[:, 0], m[:, 1], np.ones(len(m))] @ Htrue.T
    assert np.median(np.hypot(p[:, 0] / p[:, 2] - m[:, 2], p[:, 1] / p[:, 2] - m[:, 3])) < 3.0
    # query slices reproduce the full match (what each rank of the multi-GPU path computes)
    full, _, _ = pkg.match_reps(ctx, rep1, rep2)
    n = len(rep1)
    parts = [pkg.match_reps(ctx, rep1, rep2, q0, q1)[0] for q0, q1 in ((0, n // 3), (n // 3, n // 2), (n // 2, n))]
    assert np.array_equal(np.concatenate(parts), full)
    rep1.close(); rep2.close(); ctx.close()


def test_ladder_with_half_descriptors(pkg):
    """A step whose descriptor list is RootSIFT,HalfRootSIFT with both thresholds given: orientation in doHalfSIFT mode, both
    descriptor lists matched separately, tentatives joined, then duplicate filter + RANSAC - against the oracle chain."""
    import torch
    import pipeline_oracle as po
    import refdeg
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    w, h = 480, 360
    a, b, Htrue = _hard_pair(w, h, seed=21)
    steps_spec = [((1, 2, 4), 360.0), ((1, 2, 4), 120.0)]
    want = po.match_ladder(a, b, steps_spec, seed_time=31, half_orientation=True, ratio_half=0.8)
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    rep1, rep2 = pkg.ImgRep(ctx), pkg.ImgRep(ctx)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(31)
    steps = [pkg.LadderStep.make(tl, ph, half_orientation=1, fginn_half=0.8) for tl, ph in steps_spec]
    res, m = pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, steps, rep1, rep2, max_matches=100000)
    assert res.steps_done == want["steps_done"] and res.n_views == want["n_views"]
    assert list(res.n_described) == want["n_described"]
    ra, rb = rep1.fetch(), rep2.fetch()
    for got, exp in ((ra, want["regions"][0]), (rb, want["regions"][1])):
        for f in ("x", "y", "a11", "a12", "a21", "a22"):
            assert np.array_equal(got[f], exp[f]), f
        assert np.array_equal(got["desc"], exp["desc"])
    assert res.n_tentatives == want["n_tentatives"] and res.n_unique == want["n_unique"]
    assert [res.ransac_samples, res.ransac_lo, res.ransac_rejects] == want["stats"]
    assert res.n_inliers == want["n_inliers"] >= 15
    assert np.array_equal(m, want["u6"][want["mask"]][:, [0, 1, 3, 4]])
    rep1.close(); rep2.close(); ctx.close()


def test_ladder_with_two_detectors(pkg):
    """[DoG<i>] next to [HessianAffine<i>]: every detector has its own view history, banks and tentative lists; the joint list
    is the bank's key order (DoG before HessianAffine), lists of a detector without new views in a step are kept - against the
    oracle chain (CorrespondenceBank::MatchImgReps, correspondencebank.cpp:286-340)."""
    import torch
    import orc
    import pipeline_oracle as po
    import refdeg
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    w, h = 480, 360
    a, b, Htrue = _hard_pair(w, h, seed=23)
    # step 0: both detectors on the plain images; step 1: only HessianAffine brings new (tilted) views, the DoG list stays
    dets = [dict(params=orc.HessAffParams.dog(), steps=[((1,), 360.0), None], ratio=0.8),
            dict(params=orc.HessAffParams.default(), steps=[((1,), 360.0), ((1, 2, 4), 360.0)], ratio=0.8)]
    want = po.match_ladder(a, b, None, seed_time=31, min_matches=100000, detectors=dets)
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    reps1, reps2 = [pkg.ImgRep(ctx), pkg.ImgRep(ctx)], [pkg.ImgRep(ctx), pkg.ImgRep(ctx)]
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(31)
    det_steps = [[pkg.LadderStep.make((1,), 360.0), None],
                 [pkg.LadderStep.make((1,), 360.0), pkg.LadderStep.make((1, 2, 4), 360.0)]]
    res, m = pkg.match_ladder_dets_dev(ctx, t.data_ptr(), w, h, det_steps, [pkg.HessAffParams.dog(), pkg.HessAffParams.default()],
                                       reps1, reps2, min_matches=100000, max_matches=100000)
    assert res.steps_done == want["steps_done"] == 2 and res.n_views == want["n_views"]
    assert list(res.n_described) == want["n_described"]
    assert len(reps1[0]) > 50 and len(reps1[1]) > 50
    assert res.n_tentatives == want["n_tentatives"] and res.n_unique == want["n_unique"]
    assert [res.ransac_samples, res.ransac_lo, res.ransac_rejects] == want["stats"]
    assert res.n_inliers == want["n_inliers"] >= 15
    assert np.array_equal(m, want["u6"][want["mask"]][:, [0, 1, 3, 4]])
    for r in reps1 + reps2:
        r.close()
    ctx.close()


def test_ladder_with_distance_threshold(pkg):
    """DistanceThreshold > 0 for RootSIFT: the step's tentatives come from MatchFLANNDistance (Hamming nearest neighbour within
    the threshold) instead of the FGINN search, then the same duplicate filter and RANSAC - against the oracle chain."""
    import torch
    import pipeline_oracle as po
    import refdeg
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    w, h = 480, 360
    a, b, Htrue = _hard_pair(w, h, seed=25)
    dets = [dict(params=None, steps=[((1, 2), 360.0)], ratio=0.8, dist=330.0)]
    want = po.match_ladder(a, b, None, seed_time=31, detectors=dets)
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    rep1, rep2 = pkg.ImgRep(ctx), pkg.ImgRep(ctx)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(31)
    res, m = pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, [pkg.LadderStep.make((1, 2), 360.0, dist=330.0)], rep1, rep2, max_matches=100000)
    assert res.n_tentatives == want["n_tentatives"] > 100 and res.n_unique == want["n_unique"]
    assert [res.ransac_samples, res.ransac_lo, res.ransac_rejects] == want["stats"]
    assert res.n_inliers == want["n_inliers"]
    assert np.array_equal(m, want["u6"][want["mask"]][:, [0, 1, 3, 4]])
    rep1.close(); rep2.close(); ctx.close()


def test_ladder_with_grouped_detectors(pkg):
    """[Matching<i>] GroupDetectors = HessianAffine, DoG with GroupDescriptors = RootSIFT (correspondencebank.cpp:245-285): the
    two detectors' regions are searched as one list (HessianAffine first, the order named), the result sits in the bank as
    detector "Group" - between DoG and HessianAffine in the joint list - next to the separate DoG list; step 1 names the group
    again without new DoG views - against the oracle chain."""
    import torch
    import orc
    import pipeline_oracle as po
    import refdeg
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    w, h = 480, 360
    a, b, Htrue = _hard_pair(w, h, seed=27)
    dets = [dict(params=orc.HessAffParams.dog(), steps=[((1,), 360.0), None], ratio=0.8),
            dict(params=orc.HessAffParams.default(), steps=[((1,), 360.0), ((1, 2, 4), 360.0)], ratio=0.0)]   # Hessian: only through the group
    groups = [dict(dets=[1, 0], ratio=0.8), dict(dets=[1, 0], ratio=0.8)]
    want = po.match_ladder(a, b, None, seed_time=31, min_matches=100000, detectors=dets, groups=groups, group_pos=1)
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    reps1, reps2 = [pkg.ImgRep(ctx), pkg.ImgRep(ctx)], [pkg.ImgRep(ctx), pkg.ImgRep(ctx)]
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(31)
    det_steps = [[pkg.LadderStep.make((1,), 360.0), None],
                 [pkg.LadderStep.make((1,), 360.0, fginn=0.0), pkg.LadderStep.make((1, 2, 4), 360.0, fginn=0.0)]]
    g = [pkg.LadderGroup.make((1, 0), ratio=0.8), pkg.LadderGroup.make((1, 0), ratio=0.8)]
    res, m = pkg.match_ladder_dets_dev(ctx, t.data_ptr(), w, h, det_steps, [pkg.HessAffParams.dog(), pkg.HessAffParams.default()],
                                       reps1, reps2, min_matches=100000, max_matches=100000, groups=g, group_pos=1)
    assert res.steps_done == want["steps_done"] == 2 and res.n_views == want["n_views"]
    assert res.n_tentatives == want["n_tentatives"] > 20 and res.n_unique == want["n_unique"]
    assert [res.ransac_samples, res.ransac_lo, res.ransac_rejects] == want["stats"]
    assert res.n_inliers == want["n_inliers"] >= 15
    assert np.array_equal(m, want["u6"][want["mask"]][:, [0, 1, 3, 4]])
    for r in reps1 + reps2:
        r.close()
    ctx.close()


@pytest.mark.parametrize("both", [True, False])
def test_ladder_with_ground_truth_verification(pkg, both):
    """ver_type 1 (GR_TRUTH, mods.cpp:290-320): HMatrixFiltering of the unique tentatives against a known homography and, with
    doBothRANSACgroundTruth, LORANSACFiltering + HMatrixFiltering of its inliers - counts and verified list equal to the oracle chain
    (the reference's own error functions and degensac)."""
    import torch
    import pipeline_oracle as po
    import refdeg
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    w, h = 640, 480
    a, b, Htrue = synth.pair(w, h, seed=41)
    Hgt = Htrue.copy()
    Hgt[0, 2] += 1.5                                   # a slightly wrong ground truth: both sides of the threshold occur
    want = po.match_pair_ground_truth(a, b, Hgt, both=both, seed_time=31)
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    rep1, rep2 = pkg.ImgRep(ctx), pkg.ImgRep(ctx)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    pkg.ransac_pin_seed(31)
    par = pkg.PairParams.default()
    par.ransac.groundTruth = 2 if both else 1
    par.ransac.ransacForStopping = 1
    for i, v in enumerate(Hgt.reshape(9)):
        par.ransac.gtH[i] = v
    res, m = pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, [pkg.LadderStep.make((1,), 360.0)], rep1, rep2, params=par, max_matches=100000)
    assert res.n_tentatives == want["n_tentatives"] and res.n_unique == want["n_unique"] > 500
    assert res.gt_true == want["gt_true"] and 0.1 * res.n_unique < res.gt_true < res.n_unique
    assert res.gt_ransac_inliers == want["gt_ransac_inliers"] and res.gt_true_of_ransac == want["gt_true_of_ransac"]
    assert res.n_inliers == want["n_inliers"] == len(m) and np.array_equal(m, want["matches"])
    assert np.array_equal(np.array(res.H), Hgt.reshape(9))
    rep1.close(); rep2.close(); ctx.close()


_LADDER_1080P = {}


def _ladder_1080p_oracle():
    """Both HessianAffine steps of iters_MODS.ini (11 + 20 views per image) on the benchmark's hard 1920x1080 pair through the CPU
    oracle chain, kept for the session (about a minute of host time)."""
    import pipeline_oracle as po
    if "want" not in _LADDER_1080P:
        a, b, Htrue = _hard_pair(1920, 1080, seed=3000)          # bench.py --config c3
        steps_spec = [((1, 2, 4, 6, 8), 360.0), ((1, 2, 4, 6, 8), 120.0)]
        # min_matches above anything reachable: the loop does not stop after the first step, every view of both steps is made
        _LADDER_1080P.update(a=a, b=b, H=Htrue, steps=steps_spec, want=po.match_ladder(a, b, steps_spec, seed_time=31, min_matches=10 ** 6))
    return _LADDER_1080P


def test_ladder_1080p_concurrent_views_vs_oracle(pkg):
    """BASELINE configs[2] at its own size: the step loop spreads the 62 views of the two HessianAffine steps over four contexts of
    the GPU, the two images of a view through one chain of launches (run_view_jobs, csrc/imgrep.hip) - every region of both banks,
    the tentatives, the RANSAC statistics and the inlier list must be the CPU oracle chain's.  Then the same ladder again and
    again against its first result while two more contexts run the pair chain next to it (the workers' staging arenas, the side
    streams of the scale space and the matcher's exclusive workgroups are all concurrency the parity tests at small sizes do not
    reach)."""
    import threading
    import torch
    import refdeg
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    L = _ladder_1080p_oracle()
    a, b, want = L["a"], L["b"], L["want"]
    w, h = 1920, 1080
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 2)          # two image slots: paired views
    rep1, rep2 = pkg.ImgRep(ctx, 1 << 20), pkg.ImgRep(ctx, 1 << 20)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    torch.cuda.synchronize()
    steps = [pkg.LadderStep.make(tl, ph) for tl, ph in L["steps"]]

    def run():
        pkg.ransac_pin_seed(31)
        return pkg.match_ladder_dev(ctx, t.data_ptr(), w, h, steps, rep1, rep2, min_matches=10 ** 6, max_matches=200000)

    res, m = run()
    assert res.steps_done == want["steps_done"] == 2 and res.n_views == want["n_views"] == 62
    assert list(res.n_described) == want["n_described"] == [len(rep1), len(rep2)]
    first = (rep1.fetch(), rep2.fetch())
    for got, exp in zip(first, want["regions"]):
        for f in ("x", "y", "s", "a11", "a12", "a21", "a22"):
            assert np.array_equal(got[f], exp[f]), f
        assert np.array_equal(got["desc"], exp["desc"])
    assert res.n_tentatives == want["n_tentatives"] and res.n_unique == want["n_unique"]
    assert [res.ransac_samples, res.ransac_lo, res.ransac_rejects] == want["stats"]
    assert res.n_inliers == want["n_inliers"] >= 15
    assert np.array_equal(m, want["u6"][want["mask"]][:, [0, 1, 3, 4]])

    # repeated under load
    stop = threading.Event()
    errors = []
    pa, pb, _ = synth.pair(w, h, seed=2000)
    tp = torch.from_numpy(np.stack([pa, pb]).astype(np.float32)).cuda()
    torch.cuda.synchronize()

    def neighbour():
        try:
            c2 = pkg.Context(0, w, h, 2)
            while not stop.is_set():
                pkg.match_pair_dev(c2, tp.data_ptr(), w, h, max_matches=1 << 16)
            c2.close()
        except Exception as e:      # pragma: no cover
            errors.append(e)

    ths = [threading.Thread(target=neighbour) for _ in range(2)]
    for th in ths:
        th.start()
    differing = []
    try:
        for it in range(40):
            r2, m2 = run()
            regs = (rep1.fetch(), rep2.fetch())
            same = (r2.n_tentatives, r2.n_unique, r2.n_inliers) == (res.n_tentatives, res.n_unique, res.n_inliers) and np.array_equal(m2, m)
            for g0, g1 in zip(first, regs):
                same = same and len(g0) == len(g1) and all(np.array_equal(g0[f], g1[f]) for f in g0.dtype.names if f != "pad")
            if not same:
                differing.append(it)
    finally:
        stop.set()
        for th in ths:
            th.join()
    assert not errors, errors
    assert not differing, differing
    rep1.close(); rep2.close(); ctx.close()
