"""-m gpu parity tests: GPU-scored DEGENSAC fundamental-matrix verification (exp_ransacFcustom of
libmodsgpu) vs the reference's own degensac compiled from /root/reference (oracle/_ref), same
pinned seed: identical sample / LO counts, per-sample inlier histogram, plane consensus, inlier
mask; F within 1e-7 after normalisation."""
import numpy as np
import pytest

import fsynth
import refdeg

pytestmark = pytest.mark.gpu


def _normed(F):
    F = np.asarray(F, float).ravel()
    F = F / np.linalg.norm(F)
    return F * np.sign(F[np.argmax(np.abs(F))])


def _compare(got, want, tag=""):
    assert got["samples"] == want["samples"], (tag, got["samples"], want["samples"])
    assert got["lo"] == want["lo"], (tag, got["lo"], want["lo"])
    assert np.array_equal(got["hist"], want["hist"]), tag
    assert got["Ih"] == want["Ih"], (tag, got["Ih"], want["Ih"])
    assert got["I"] == want["I"], (tag, got["I"], want["I"])
    assert np.array_equal(got["inl"], want["inl"]), tag
    assert np.max(np.abs(_normed(got["F"]) - _normed(want["F"]))) < 1e-7, tag


CASES = [  # n, inlier ratio, fraction of inliers on one plane, noise
    (60, 0.9, 0.0, 0.3), (200, 0.7, 0.0, 0.5), (200, 0.7, 0.6, 0.5), (500, 0.5, 0.0, 0.5), (500, 0.5, 0.8, 0.5),
    (1000, 0.35, 0.0, 0.7), (1000, 0.4, 0.5, 0.7), (2000, 0.3, 0.9, 0.5), (400, 0.6, 1.0, 0.5)]


@pytest.mark.skipif(not refdeg.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("n,ratio,plane,noise", CASES)
@pytest.mark.parametrize("seed", [12345, 7])
def test_ransac_f_matches_reference(pkg, n, ratio, plane, noise, seed):
    u, _, _ = fsynth.two_view(n, ratio, plane, noise, seed=seed + n)
    for err, sym in (("sampson", 0), ("sampson", 1), ("symm", 1)):
        want = refdeg.ransac_f(u, 16.0, max_sam=20000, err=err, sym_check=sym, seed_time=seed)
        got = pkg.ransac_f(u, 16.0, max_sam=20000, err=err, sym_check=sym, seed_time=seed)
        _compare(got, want, "%s sym=%d" % (err, sym))


@pytest.mark.skipif(not refdeg.available(), reason="oracle/_ref not built")
def test_ransac_f_no_lo_and_limits(pkg):
    u, _, _ = fsynth.two_view(600, 0.5, 0.3, 0.5, seed=3)
    for do_lo, lim in ((0, 0), (1, 49), (1, 1000000)):
        want = refdeg.ransac_f(u, 9.0, max_sam=5000, do_lo=do_lo, inl_limit=lim, seed_time=99)
        got = pkg.ransac_f(u, 9.0, max_sam=5000, do_lo=do_lo, inl_limit=lim, seed_time=99)
        _compare(got, want, "do_lo=%d lim=%d" % (do_lo, lim))


@pytest.mark.skipif(not refdeg.available(), reason="oracle/_ref not built")
def test_ransac_f_large(pkg):
    u, _, _ = fsynth.two_view(15000, 0.25, 0.3, 0.7, seed=11, size=(4096, 4096))
    want = refdeg.ransac_f(u, 16.0, max_sam=100000, seed_time=4242)
    got = pkg.ransac_f(u, 16.0, max_sam=100000, seed_time=4242)
    _compare(got, want)


def test_ransac_f_properties(pkg):
    """Size-independent properties: determinism under a pinned seed; the returned mask is exactly the set
    of correspondences whose Sampson error under the returned F is within the threshold; epipolar
    geometry recovers the true inliers."""
    u, true_in, _ = fsynth.two_view(4000, 0.4, 0.2, 0.5, seed=21)
    a = pkg.ransac_f(u, 16.0, seed_time=5)
    b = pkg.ransac_f(u, 16.0, seed_time=5)
    assert np.array_equal(a["inl"], b["inl"]) and a["samples"] == b["samples"] and np.array_equal(a["F"], b["F"])
    F = a["F"]
    rxc = F[0] * u[:, 3] + F[3] * u[:, 4] + F[6]
    ryc = F[1] * u[:, 3] + F[4] * u[:, 4] + F[7]
    rwc = F[2] * u[:, 3] + F[5] * u[:, 4] + F[8]
    r = u[:, 0] * rxc + u[:, 1] * ryc + rwc
    rx = F[0] * u[:, 0] + F[1] * u[:, 1] + F[2]
    ry = F[3] * u[:, 0] + F[4] * u[:, 1] + F[5]
    d = r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry)
    border = np.abs(d - 16.0) < 1e-6
    assert np.array_equal((d <= 16.0)[~border], a["inl"].astype(bool)[~border])
    assert a["I"] == int(a["inl"].sum())
    assert (a["inl"].astype(bool) & true_in).sum() > 0.95 * true_in.sum()
    assert abs(np.linalg.det(F.reshape(3, 3))) < 1e-9 * np.linalg.norm(F) ** 3


def test_ransac_f_matches_golden_fixtures(pkg):
    """The committed outputs of the reference's exp_ransacFcustom (tests/golden/ransac_f.npz, tools/gen_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ransac_f.npz"))
    for key in g["keys"]:
        parts = str(key).split("_")
        ci, seed, sym, err = int(parts[0]), int(parts[-1]), int(parts[-2]), "_".join(parts[1:-2])
        got = pkg.ransac_f(g["u_%d" % ci], 16.0, max_sam=20000, err=err, sym_check=sym, seed_time=seed)
        assert [got["I"], got["samples"], got["lo"], got["Ih"]] == list(g["stat_" + str(key)]), key
        assert np.array_equal(got["inl"], g["inl_" + str(key)]), key
        nz = np.flatnonzero(got["hist"])
        assert np.array_equal(nz, g["hist_" + str(key)]) and np.array_equal(got["hist"][nz], g["histv_" + str(key)]), key
        assert np.max(np.abs(_normed(got["F"]) - _normed(g["F_" + str(key)]))) < 1e-7, key
