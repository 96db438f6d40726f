"""-m gpu parity tests: MFMA brute-force FGINN matcher vs the CPU oracle (exact)."""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

TF = ("q", "t", "t_bad", "t_2nd", "d1", "d2", "d2nd", "ratio")


def _rand_regions(n, seed, w=800, h=600, clusters=0):
    rng = np.random.default_rng(seed)
    r = np.zeros(n, orc.REGION_DTYPE)
    r["x"] = rng.uniform(1, w - 1, n).astype(np.float32)
    r["y"] = rng.uniform(1, h - 1, n).astype(np.float32)
    if clusters:
        # many near-duplicates within contradDist of each other (same place, similar descriptors)
        c = rng.integers(0, clusters, n)
        cx = rng.uniform(20, w - 20, clusters); cy = rng.uniform(20, h - 20, clusters)
        r["x"] = (cx[c] + rng.uniform(-3, 3, n)).astype(np.float32)
        r["y"] = (cy[c] + rng.uniform(-3, 3, n)).astype(np.float32)
    r["s"] = 2.0; r["a11"] = 1.0; r["a22"] = 1.0
    v = rng.gamma(0.6, 40.0, (n, 128))
    if clusters:
        base = rng.gamma(0.6, 40.0, (clusters, 128))
        v = base[c] + rng.normal(0, 6.0, (n, 128))
    r["desc"] = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    return r


def _assert_tents_equal(got, want):
    assert len(got) == len(want)
    for f in TF:
        assert np.array_equal(got[f], want[f]), f


@pytest.mark.parametrize("nq,nt,seed", [(1, 1, 1), (1, 2, 2), (5, 3, 3), (33, 31, 4), (300, 257, 5), (1000, 1500, 6)])
def test_match_random(gpu_ctx, nq, nt, seed):
    q, t = _rand_regions(nq, seed), _rand_regions(nt, seed + 100)
    t["desc"][: min(nq, nt) // 2] = q["desc"][: min(nq, nt) // 2]      # exact matches: d0 = 0
    t["desc"][-1] = t["desc"][0]                                       # exact ties between trains
    for ratio in (0.8, 0.95):
        got, u6 = gpu_ctx.match_fginn(q, t, ratio)
        want = orc.match_fginn(q, t, ratio)
        _assert_tents_equal(got, want)
        assert np.array_equal(u6[:, 0], q["x"][got["q"]]) and np.array_equal(u6[:, 4], t["y"][got["t"]])
        assert np.all(u6[:, 2] == 1) and np.all(u6[:, 5] == 1)


def test_match_clusters_rank_cap(gpu_ctx):
    """Near-duplicate clusters: more than nn consistent neighbours below the ratio threshold."""
    q = _rand_regions(400, 11, clusters=4)
    t = _rand_regions(600, 12, clusters=4)
    for nn in (50, 8, 3):
        got, _ = gpu_ctx.match_fginn(q, t, 0.97, 10.0, nn)
        want = orc.match_fginn(q, t, 0.97, 10.0, nn)
        _assert_tents_equal(got, want)


def test_match_ties_and_parity(gpu_ctx):
    """Descriptors from a tiny alphabet: many trains at exactly the same distance (index order decides), odd and even
    norms mixed (the matcher ranks by a half-precision seed and settles parity on the exact path), shrinking lists
    (stale entries of an earlier, larger call stay behind the end of the lists)."""
    rng = np.random.default_rng(77)
    for nq, nt in ((700, 1900), (257, 95), (64, 33)):
        q, t = _rand_regions(nq, 1000 + nq), _rand_regions(nt, 2000 + nt)
        q["desc"] = rng.integers(0, 3, (nq, 128)).astype(np.uint8) * 40
        t["desc"] = rng.integers(0, 3, (nt, 128)).astype(np.uint8) * 40
        t["desc"][::7, :3] += 1                      # odd norms on a subset
        t["desc"][nt // 2:] = t["desc"][: nt - nt // 2]   # every train of the first half has an exact twin later in the list
        for ratio in (0.8, 0.999):
            got, _ = gpu_ctx.match_fginn(q, t, ratio, 1e9)     # no contradiction cut: deep neighbour walks
            want = orc.match_fginn(q, t, ratio, 1e9)
            _assert_tents_equal(got, want)


def test_match_repeated_trains(gpu_ctx):
    """Hundreds of identical train descriptors spread over many tiles: more than three half tiles of one train split tie at the
    level of a query's runner-up, the case in which pass 1's three keys per split cannot name every candidate and the exact
    finish rescans the query over all trains (match_fix_kernel's fallback).  Equal distances are ordered by the lower index."""
    rng = np.random.default_rng(404)
    nq, nt = 600, 6000
    q, t = _rand_regions(nq, 31), _rand_regions(nt, 32)
    base = q["desc"][0].copy()
    rep = np.arange(7, nt, 9)                       # 666 copies of one descriptor, every ninth train
    t["desc"][rep] = base
    q["desc"][:40] = base                            # distance 0 to all of them
    noisy = np.clip(base.astype(np.int16) + rng.integers(-2, 3, (60, 128)), 0, 255).astype(np.uint8)
    q["desc"][40:100] = noisy                        # the same small distance to all of them
    t["x"][rep] = t["x"][rep[0]]; t["y"][rep] = t["y"][rep[0]]          # ... at one place: consistent neighbours, deep walks
    for ratio, nn in ((0.8, 50), (0.999, 50), (0.999, 1000)):
        got, _ = gpu_ctx.match_fginn(q, t, ratio, 10.0, nn)
        want = orc.match_fginn(q, t, ratio, 10.0, nn)
        _assert_tents_equal(got, want)


def test_match_large_lists(gpu_ctx, pkg):
    """Lists of the size of a view-synthesis bank (more than one query block per wave, many train splits)."""
    ctx = pkg.Context(0, 2048, 2048, 1)      # capacity 524288 regions
    q, t = _rand_regions(9000, 5), _rand_regions(23000, 6)
    t["desc"][:4000] = np.clip(q["desc"][:4000].astype(np.int16) + np.random.default_rng(1).integers(-3, 4, (4000, 128)), 0, 255).astype(np.uint8)
    got, _ = ctx.match_fginn(q, t, 0.8)
    want = orc.match_fginn(q, t, 0.8)
    _assert_tents_equal(got, want)
    assert len(got) > 3000
    ctx.close()


def test_match_empty(gpu_ctx):
    q, t = _rand_regions(10, 1), _rand_regions(0, 2)
    assert len(gpu_ctx.match_fginn(q, t)[0]) == 0
    assert len(gpu_ctx.match_fginn(t, q)[0]) == 0


def test_match_pair_and_dupfilter(gpu_ctx, pkg):
    import torch
    a, b, H = synth.pair(1280, 960, seed=5)
    t = torch.from_numpy(np.stack([a, b])).cuda()
    ctx2 = pkg.Context(0, 1280, 960, 2)
    nd, nr = ctx2.detect_describe_dev(t.data_ptr(), 2, 1280, 960)
    ra, rb = ctx2.regions_fetch(0), ctx2.regions_fetch(1)
    got, u6 = ctx2.match_dev(0, 1)
    want = orc.match_fginn(ra, rb)
    assert len(want) > 100
    _assert_tents_equal(got, want)
    fw = orc.duplicate_filter(want, ra, rb, 2.0, 1)
    fg, fu = pkg.duplicate_filter(got, u6, 2.0, 1)
    _assert_tents_equal(fg, fw)
    assert np.array_equal(fu[:, 0], ra["x"][fg["q"]]) and np.array_equal(fu[:, 3], rb["x"][fg["t"]])
    # most tentatives agree with the generating homography
    p = np.c_[fu[:, 0], fu[:, 1], np.ones(len(fu))] @ H.T
    err = np.hypot(p[:, 0] / p[:, 2] - fu[:, 3], p[:, 1] / p[:, 2] - fu[:, 4])
    assert (err < 3).mean() > 0.5
    ctx2.close()


@pytest.mark.parametrize("nq,nt,seed", [(1, 1, 1), (1, 2, 2), (5, 3, 3), (65, 64, 4), (300, 257, 5), (1000, 1500, 6)])
@pytest.mark.parametrize("threshold", [0.9, 260.0, 1024.0])
def test_distance_matcher_hamming(gpu_ctx, nq, nt, seed, threshold):
    """MatchFLANNDistance (matching.cpp:572-633): nearest by Hamming distance within (int)(float)threshold, ties by train
    index, d2 = the second smallest distance - equal to the oracle on random lists with many ties (few distinct bytes)."""
    q = _rand_regions(nq, seed)
    t = _rand_regions(nt, seed + 100)
    rng = np.random.default_rng(seed)
    q["desc"] = rng.integers(0, 4, (nq, 128)).astype(np.uint8) * 85      # distances cluster: ties between trains are common
    t["desc"] = rng.integers(0, 4, (nt, 128)).astype(np.uint8) * 85
    if nt > 2:
        t["desc"][nt // 2] = t["desc"][0]                                   # an exact duplicate train: index order decides
    got, u6 = gpu_ctx.match_distance(q, t, threshold)
    want = orc.match_distance(q, t, threshold)
    _assert_tents_equal(got, want)
    if len(want):
        assert np.array_equal(u6[:, 0], q["x"][want["q"]]) and np.array_equal(u6[:, 3], t["x"][want["t"]])
    if threshold >= 1024:
        assert len(want) == nq


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_duplicate_filter_on_the_device_equals_the_host_filter(gpu_ctx, pkg, mode):
    """DuplicateFiltering as three launches (csrc/dedup.hip) against the sequential host form (mods_duplicate_filter, itself checked
    against the oracle above): clusters of near-duplicates with chains (a drops b, so c - near b only - stays), equal sort keys,
    points exactly at the distance, a list in reverse key order; every mode of whichCorrespondenceRemains."""
    rng = np.random.default_rng(100 + mode)
    for n, spread in ((1, 50.0), (700, 60.0), (6000, 900.0), (3000, 40.0)):
        base = rng.uniform(0, spread, (n, 4))
        k = n // 3                                          # a third of the list: copies of other entries, jittered around the radius
        src = rng.integers(0, n, k)
        base[n - k:] = base[src] + rng.choice([0.0, 0.7, 1.4, 1.99, 2.01], (k, 4)) * rng.choice([-1.0, 1.0], (k, 4)) * 0.7071
        if n > 10:                                          # exactly on the circle in image 1 (3-4-5 triangle, r = 2)
            base[5] = base[4] + np.array([1.2, 1.6, 0.0, 0.0])
            base[7] = base[6] + np.array([1.2, 1.6, 1.2, 1.6])
        tent = np.zeros(n, pkg.TENT_DTYPE)
        tent["q"] = np.arange(n); tent["t"] = rng.permutation(n)
        tent["ratio"] = np.round(rng.uniform(0.1, 0.8, n), 2)      # many equal keys: list order decides among them
        tent["d1"] = np.round(rng.uniform(1000, 90000, n), -2).astype(np.float32)
        u6 = np.ones((n, 6))
        u6[:, 0:2] = base[:, 0:2]; u6[:, 3:5] = base[:, 2:4]
        laf = rng.uniform(0.5, 3.0, (n, 14))
        laf[:, 6] = np.round(laf[:, 6], 1)
        if n == 3000:                                       # already sorted the wrong way round
            o = np.argsort(-tent["ratio"], kind="stable")
            tent, u6, laf = tent[o], u6[o], laf[o]
        ht, hu, hl = pkg.duplicate_filter(tent, u6, 2.0, mode, laf=laf)
        gt, gu, gl, on_dev = pkg.duplicate_filter_gpu(gpu_ctx, tent, u6, laf, 2.0, mode)
        assert on_dev or n == 3000, (n, mode)               # (the 40 x 40 square packs more than 12 near predecessors: host hand-over)
        assert len(gt) == len(ht) and 0 < len(ht) <= n
        assert gt.tobytes() == ht.tobytes() and np.array_equal(gu, hu) and np.array_equal(gl, hl), (n, mode)


@pytest.mark.gpu
def test_pair_with_device_and_host_duplicate_filter(pkg, monkeypatch):
    """mods_match_pair_dev with the filter behind the search on the device (default) and with doBeforeRANSAC = 0 (the filter runs on
    the verified list, on the host): unique counts against the host filter applied to the same tentatives."""
    import torch
    w, h = 640, 480
    a, b, _ = synth.pair(w, h, seed=21)
    t = torch.from_numpy(np.stack([a, b])).cuda(); torch.cuda.synchronize()
    ctx = pkg.Context(0, w, h, 2)
    par = pkg.PairParams.default()
    pkg.ransac_pin_seed(5)
    res, _ = pkg.match_pair_dev(ctx, t.data_ptr(), w, h, par)
    tent, u6 = ctx.match_dev(0, 1)                       # the regions of the pair are still in the context
    ht, _ = pkg.duplicate_filter(tent, u6, par.dup_dist, par.dup_mode)
    assert res.n_tentatives == len(tent) and res.n_unique == len(ht) and 0 < len(ht) < len(tent)
    par.dup_before_ransac = 0
    res2, _ = pkg.match_pair_dev(ctx, t.data_ptr(), w, h, par)
    assert res2.n_tentatives == res.n_tentatives and res2.n_unique == res.n_tentatives and res2.n_inliers > 15
    ctx.close()
