"""-m gpu parity tests: orientation + RootSIFT through the C ABI vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

RFIELDS = ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type", "parent")


def _assert_regions_equal(got, want):
    assert len(got) == len(want)
    for f in RFIELDS:
        assert np.array_equal(got[f], want[f]), "field %s differs" % f
    bad = np.nonzero((got["desc"] != want["desc"]).any(axis=1))[0]
    assert len(bad) == 0, "descriptors differ for %d regions, first %s" % (len(bad), bad[:5])


def _patches(n, ps, seed):
    rng = np.random.default_rng(seed)
    img = synth.texture(640, 480, seed=seed)
    out = []
    for _ in range(n):
        x, y = rng.integers(0, 640 - ps), rng.integers(0, 480 - ps)
        out.append(img[y:y + ps, x:x + ps].copy())
    out.append(np.full((ps, ps), 100.0, np.float32))          # flat patch: no gradients at all
    g = np.tile(np.arange(ps, dtype=np.float32) * 4, (ps, 1))  # pure ramp: single orientation
    out.append(g)
    out.append(g.T.copy())
    return out


def test_dominant_angle_patches(gpu_ctx):
    for p in _patches(40, 32, 5):
        fw, aw = orc.dominant_angle(p)
        fg, ag = gpu_ctx.dominant_angle(p)
        assert fw == fg
        if fw:
            assert np.float32(aw) == np.float32(ag)


def test_fast_sqrt_is_sqrtf(gpu_ctx):
    """The gradient magnitudes of the orientation and SIFT kernels come from fast_sqrtf (csrc/device_util.hpp): hipcc's correctly
    rounded sqrtf without the rescaling of operands below 2^-96 and without the zero / infinity class test.  Exhaustive: every one of
    the 2^32 float operands against sqrtf on the device - identical for +0, every x >= 2^-96, +infinity and NaN; the form the SIFT
    kernel uses (the compiler's expansion whenever a lane of the wave holds a smaller positive operand) for every non-negative
    operand.  Negative operands are outside both domains (the callers pass sums of squares): the 2^23 - 1 negative denormals
    give -0 where sqrtf says NaN, and nothing else differs there."""
    bad_domain, bad_tiny, bad_any, seen, bad_negative = gpu_ctx.selftest_fast_sqrt()
    assert seen == 1 << 32
    assert bad_domain == 0 and bad_any == 0, (bad_domain, bad_tiny, bad_any, bad_negative)
    assert bad_negative <= (1 << 23) - 1


@pytest.mark.parametrize("ps", [34, 41, 48])
def test_dominant_angle_large_orientation_patches(gpu_ctx, ps):
    """[DominantOrientation] patchSize above 32: ps*(ps-2) > 1024 votes, and on a ramp (or a straight edge) every vote goes to ONE
    10-degree bin, so a vote's place in its bin's list passes 1023 - the list places are 32-bit words there."""
    pats = _patches(12, ps, 21)
    ramp = np.tile(np.arange(ps, dtype=np.float32) * 3, (ps, 1))
    c, s = np.cos(0.35), np.sin(0.35)
    yy, xx = np.mgrid[0:ps, 0:ps].astype(np.float32)
    pats += [ramp, ramp.T.copy(), (2.5 * (c * xx + s * yy)).astype(np.float32), np.where(xx + yy > ps, 200.0, 20.0).astype(np.float32)]
    for p in pats:
        fw, aw = orc.dominant_angle(p)
        fg, ag = gpu_ctx.dominant_angle(p)
        assert fw == fg
        if fw:
            assert np.float32(aw) == np.float32(ag)


@pytest.mark.parametrize("root", [True, False])
def test_sift_patches(gpu_ctx, root):
    for p in _patches(40, 41, 9):
        assert np.array_equal(gpu_ctx.sift_patch(p, root), orc.sift_desc(p, root))


@pytest.mark.parametrize("w,h,seed", [(320, 240, 7), (515, 389, 11), (800, 640, 3)])
def test_orient_describe(gpu_ctx, w, h, seed):
    img = synth.texture(w, h, seed=seed)
    keys = orc.detect_hessian_affine(img)
    regs = orc.regions_from_keys(keys)
    want = orc.describe_rootsift(img, orc.filter_touch_boundary(orc.detect_orientation(img, orc.filter_centres_inside(regs, w, h)), w, h))
    # oracle `parent` indexes the centre-filtered list; all centres are inside for detector output
    assert len(orc.filter_centres_inside(regs, w, h)) == len(regs)
    got = gpu_ctx.orient_describe(img, keys)
    _assert_regions_equal(got, want)


def test_graf_end_to_end_counts(gpu_ctx):
    """README.md:91-108 of the reference: 2665 -> 2331 and 3287 -> 2912 regions -> descriptors."""
    import os
    import torch
    from PIL import Image
    for name, nreg, ndesc in (("graf1", 2665, 2331), ("graf6", 3287, 2912)):
        g = orc.grey_of_rgb(np.asarray(Image.open(os.path.join(os.path.dirname(__file__), "golden", name + ".png")).convert("RGB")))
        want, nd_want = orc.detect_describe(g)
        t = torch.from_numpy(g).cuda()
        nd, nr = gpu_ctx.detect_describe_dev(t.data_ptr(), 1, g.shape[1], g.shape[0])
        assert nd[0] == nd_want == nreg
        assert nr[0] == len(want) == ndesc
        _assert_regions_equal(gpu_ctx.regions_fetch(0), want)


def test_large_regions_tier_b(gpu_ctx):
    """Measurement regions above the LDS/slab tier (P2 > 160) go through the big-scratch launch."""
    img = synth.texture(1200, 900, seed=31)
    keys = orc.detect_hessian_affine(img)[:64].copy()
    keys["x"] = 600.0; keys["y"] = 450.0
    keys["s"] = np.linspace(14.0, 60.0, len(keys))
    regs = orc.regions_from_keys(keys)
    want = orc.describe_rootsift(img, orc.filter_touch_boundary(orc.detect_orientation(img, regs), 1200, 900))
    assert len(want) > 10
    _assert_regions_equal(gpu_ctx.orient_describe(img, keys), want)


def test_regions_above_the_fused_tier(pkg):
    """Measurement regions of more than 1024 px (P2 > BIG_FUSE_P2: the sampled region lives in an HBM slab, wave-per-item sample and
    row-pass kernels) next to ones just below the limit (the fused kernel with 8 rows per item)."""
    w = h = 2400
    img = synth.texture(w, h, seed=33)
    keys = orc.detect_hessian_affine(img[:600, :600].copy())[:4].copy()
    keys["x"] = 1200.0; keys["y"] = 1200.0
    keys["s"] = np.array([90.0, 97.0, 99.0, 106.0])          # P2 = 939, 1013, 1033, 1105
    regs = orc.regions_from_keys(keys)
    want = orc.describe_rootsift(img, orc.filter_touch_boundary(orc.detect_orientation(img, regs), w, h))
    assert len(want) == 4
    ctx = pkg.Context(0, w, h, 1)
    _assert_regions_equal(ctx.orient_describe(img, keys), want)
    ctx.close()


def test_describe_1080p_batch(gpu_ctx):
    import torch
    a = synth.texture(1920, 1080, seed=21)
    b = synth.texture(1920, 1080, seed=22)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    nd, nr = gpu_ctx.detect_describe_dev(t.data_ptr(), 2, 1920, 1080)
    for i, im in enumerate((a, b)):
        want, ndw = orc.detect_describe(im)
        assert nd[i] == ndw
        _assert_regions_equal(gpu_ctx.regions_fetch(i), want)


@pytest.mark.parametrize("w,h", [(1921, 1081), (1922, 1079), (1923, 1080)])
def test_large_planes_with_unaligned_rows(pkg, w, h):
    """Rows that are not a multiple of four pixels on planes large enough for the 32-row tiles of the fused blur + response kernel
    (two images of ~1920 x 1080): its window loader takes 16-byte loads at any dword alignment, and a window float4 that reaches
    past the last column is the row's last float4 shifted down with the last pixel repeated (csrc/pyramid.hip: load_window) - a path
    that images with aligned rows, and single images of this size (16-row tiles), never take."""
    import torch
    import pipeline_oracle as po
    imgs = [synth.texture(w, h, seed=900 + w + i) for i in range(2)]
    want = po.pmap(orc.detect_describe, imgs)
    ctx = pkg.Context(0, w, h, 2)
    t = torch.from_numpy(np.stack(imgs)).cuda()
    torch.cuda.synchronize()
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), 2, w, h)
    for i, (exp, nd_exp) in enumerate(want):
        assert nd[i] == nd_exp and nr[i] == len(exp), i
        _assert_regions_equal(ctx.regions_fetch(i), exp)
    ctx.close()


def test_batch16_1080p_vs_oracle(pkg):
    """The batching of the benchmark's pipeline (pairs_per_batch = 8: 16 images of 1920 x 1080 per launch) against the
    oracle directly, image by image."""
    import torch
    import pipeline_oracle as po
    w, h = 1920, 1080
    imgs = []
    for i in range(8):
        a, b, _ = synth.pair(w, h, seed=2000 + i)
        imgs += [a, b]
    want = po.pmap(orc.detect_describe, imgs)
    ctx = pkg.Context(0, w, h, 16)
    t = torch.from_numpy(np.stack(imgs)).cuda()
    torch.cuda.synchronize()
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), 16, w, h)
    for i, (exp, nd_exp) in enumerate(want):
        assert nd[i] == nd_exp and nr[i] == len(exp), i
        _assert_regions_equal(ctx.regions_fetch(i), exp)
    ctx.close()


def test_graph_replay_equals_eager_launches(pkg):
    """mods_ctx_graphs: from the second call with the same arguments on, the launches of detect + describe are recorded and replayed
    as one hipGraph - regions identical to the eager calls', for different images in the same device buffer (what a pipeline worker
    does batch after batch).  A batch whose scale space forks onto the side stream (16 x 720p: two branches in the graph) is
    replayed; a call without the fork (2 images, or one pyramid stream) stays eager, because a linear recording faults on replay
    with this runtime (csrc/capi.hip: dd_run; tools/exp_graph.py)."""
    import torch
    w, h = 1280, 720
    for n_img, streams, replayed in ((16, 2, True), (2, 2, False), (16, 1, False)):
        imgs = [np.stack([synth.texture(w, h, seed=500 + 7 * j + i) for i in range(n_img)]) for j in range(2)]
        buf = torch.from_numpy(imgs[0]).cuda()
        torch.cuda.synchronize()
        eager = pkg.Context(0, w, h, n_img)
        eager.pyramid_streams(streams)
        want = []
        for im in imgs:
            buf.copy_(torch.from_numpy(im)); torch.cuda.synchronize()
            eager.detect_describe_dev(buf.data_ptr(), n_img, w, h)
            want.append([eager.regions_fetch(i) for i in range(n_img)])
        eager.close()
        ctx = pkg.Context(0, w, h, n_img, nonblocking=True)
        ctx.pyramid_streams(streams)
        ctx.graphs(True)
        for rep in range(3):
            for im, exp in zip(imgs, want):
                buf.copy_(torch.from_numpy(im)); torch.cuda.synchronize()
                ctx.detect_describe_dev(buf.data_ptr(), n_img, w, h)
                for s in range(n_img):
                    _assert_regions_equal(ctx.regions_fetch(s), exp[s])
        # calls 1-2 eager (the first one uploads tables: the second is the first plain repeat), call 3 records - and replays when the
        # recording has two branches -, 4.. replay
        assert (ctx.graph_replays() >= 3) if replayed else (ctx.graph_replays() == 0), (n_img, streams, ctx.graph_replays())
        ctx.close()


def test_half_orientation_and_half_rootsift(pkg):
    """DetectOrientation in doHalfSIFT mode + HalfRootSIFT (64 values) for the same regions, as a step with
    Descriptors = RootSIFT,HalfRootSIFT asks for (imagerepresentation.cpp:725-731, 909-943, 970-979; siftdesc.cpp:401-436)."""
    import torch
    w, h = 640, 480
    img = synth.texture(w, h, seed=33)
    want, want_half, nd_want = orc.detect_describe(img, half_orientation=True, half_desc=True)
    plain, _ = orc.detect_describe(img)
    assert not np.array_equal(want["a11"][:50], plain["a11"][:50])      # the half mode does pick other angles
    ctx = pkg.Context(0, w, h, 1)
    desc = pkg.DescribeParams.default()
    desc.ori_halfMode, desc.halfDesc = 1, 1
    t = torch.from_numpy(img).cuda()
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), 1, w, h, None, desc)
    assert nd[0] == nd_want and nr[0] == len(want) > 500
    _assert_regions_equal(ctx.regions_fetch(0), want)
    got_half = ctx.regions_fetch_half(0)
    _assert_regions_equal(got_half, want_half)
    assert np.all(got_half["desc"][:, 64:] == 0) and got_half["desc"][:, :64].any()
    ctx.close()


@pytest.mark.parametrize("max_angles", [1, 0])
def test_add_upright(pkg, max_angles):
    """[DominantOrientation] addUpRight (imagerepresentation.cpp:915-930, synth-detection.cpp:1140-1142): the unrotated copy of
    every region that passes the border test is described too and comes first; maxAngles = 0 leaves only those."""
    import torch
    w, h = 640, 480
    img = synth.texture(w, h, seed=34)
    want, nd_want = orc.detect_describe(img, add_upright=True, max_angles=max_angles)
    ctx = pkg.Context(0, w, h, 1)
    desc = pkg.DescribeParams.default()
    desc.addUpRight, desc.ori_maxAngles = 1, max_angles
    t = torch.from_numpy(img).cuda()
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), 1, w, h, None, desc)
    assert nd[0] == nd_want and nr[0] == len(want) > 500
    got = ctx.regions_fetch(0)
    _assert_regions_equal(got, want)
    n_up = int(np.sum(got["a12"] == 0))
    assert n_up >= len(got) // 2 and np.all(got["a12"][:n_up] == 0)
    ctx.close()


def test_fast_patch_extraction(pkg):
    """[SIFTDescriptor] FastPatchExtraction: the fast_extraction branch of DescribeRegions (synth-detection.hpp:232-253) - one
    interpolation of the image at (2*int(mrSize*s)+1)/patchSize, no smoothing - for every region, large ones included."""
    import torch
    w, h = 800, 600
    img = synth.texture(w, h, seed=35)
    want, nd_want = orc.detect_describe(img, fast_extraction=True)
    plain, _ = orc.detect_describe(img)
    assert len(want) == len(plain) and not np.array_equal(want["desc"], plain["desc"])
    ctx = pkg.Context(0, w, h, 1)
    desc = pkg.DescribeParams.default()
    desc.fastExtraction = 1
    t = torch.from_numpy(img).cuda()
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), 1, w, h, None, desc)
    assert nd[0] == nd_want and nr[0] == len(want) > 500
    _assert_regions_equal(ctx.regions_fetch(0), want)
    ctx.close()


@pytest.mark.parametrize("max_angles,half,upright", [(2, False, False), (5, False, False), (3, True, False), (2, False, True)])
def test_several_dominant_orientations(pkg, max_angles, half, upright):
    """[DominantOrientation] maxAngles > 1 (EstimateDominantAnglesFunctor, synth-detection.cpp:900-927; DetectOrientation
    :1095-1106): every keypoint yields an oriented copy per histogram peak above the threshold, the first maxAngles of them in
    bin order, each with its own border test; copies of one keypoint are neighbours in the list."""
    import torch
    w, h = 640, 480
    img = synth.texture(w, h, seed=37)
    one, _ = orc.detect_describe(img)
    want, nd_want = orc.detect_describe(img, max_angles=max_angles, half_orientation=half, add_upright=upright)
    assert len(want) > len(one) * (1.05 if not upright else 2.0)
    ctx = pkg.Context(0, w, h, 2)
    desc = pkg.DescribeParams.default()
    desc.ori_maxAngles, desc.ori_halfMode, desc.addUpRight = max_angles, int(half), int(upright)
    t = torch.from_numpy(np.stack([img, img[::-1].copy()])).cuda()
    nd, nr = ctx.detect_describe_dev(t.data_ptr(), 2, w, h, None, desc)
    assert nd[0] == nd_want and nr[0] == len(want)
    _assert_regions_equal(ctx.regions_fetch(0), want)
    want2, _ = orc.detect_describe(img[::-1].copy(), max_angles=max_angles, half_orientation=half, add_upright=upright)
    _assert_regions_equal(ctx.regions_fetch(1), want2)
    ctx.close()


def test_several_orientations_on_a_view(pkg):
    import math
    import torch
    img = synth.texture(640, 480, seed=38)
    h, w = img.shape
    d = pkg.view_ctx_dims(w, h)
    ctx = pkg.Context(0, d[0], d[1], 1)
    t = torch.from_numpy(img).cuda()
    px, g0 = orc.synth_view(img, 3.0, math.pi / 5, 1.0, 0.2, 1)
    import ctypes as C
    a, p = orc._f(px)
    Hc = np.ascontiguousarray(np.array(g0.H), np.float64).ravel()
    out = np.zeros(1 << 18, orc.REGION_DTYPE); det = np.zeros(1 << 18, orc.REGION_DTYPE); ndet = C.c_int()
    par = orc.HessAffParams.default()
    n = orc.lib().orc_detect_describe_view_ex(p, a.shape[1], a.shape[0], Hc.ctypes.data_as(C.c_void_p), w, h, C.byref(par), C.c_double(orc.ORI_MRSIZE),
                                             orc.ORI_PATCH, 2, C.c_double(orc.ORI_TH), C.c_double(orc.DESC_MRSIZE), orc.DESC_PATCH, 1, 0,
                                             out.ctypes.data_as(C.c_void_p), det.ctypes.data_as(C.c_void_p), None, 1 << 18, C.byref(ndet))
    want = out[:n]
    desc = pkg.DescribeParams.default()
    desc.ori_maxAngles = 2
    g, nd, nr = ctx.detect_describe_view_dev(t.data_ptr(), w, h, 3.0, math.pi / 5, 1.0, 0.2, 1, desc=desc)
    assert nd == ndet.value and nr == n > 100
    got = ctx.regions_fetch(0)
    for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type"):     # (parent: the oracle numbers the keypoints after the centre test)
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["desc"], want["desc"])
    ctx.close()
