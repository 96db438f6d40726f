"""CPU tests of the fast host forms of the homography LO step: the SIMD error functions, the gain pass of inlidxs and the
moment matrix folded into 30 ordered sums must give the bits of the scalar code they replace (libmodsgpu's own scalar
functions) and of the reference's own degensac build where it is available (oracle/_ref; Htools.c, utools.c, rtools.c)."""
import ctypes as C

import numpy as np
import pytest

import refdeg


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def _points(n, seed, noise=0.8, outliers=0.3):
    g = np.random.default_rng(seed)
    H = np.array([[1.02, -0.04, 14.0], [0.03, 0.97, -9.0], [2e-5, 1e-5, 1.0]])
    x = g.uniform(0, 1900, (n, 2))
    y = np.c_[x, np.ones(n)] @ H.T
    y = y[:, :2] / y[:, 2:] + g.normal(0, noise, (n, 2))
    out = g.random(n) < outliers
    y[out] = g.uniform(0, 1900, (int(out.sum()), 2))
    u = np.ones((n, 6))
    u[:, 0:2], u[:, 3:5] = x, y
    # the model in the layout exp_ransacHcustom carries (column-major h, applied as in Htools.c)
    h = np.ascontiguousarray(np.linalg.inv(H).T.ravel() / np.linalg.inv(H)[2, 2])
    return np.ascontiguousarray(u), h


LANES = (1, 4, 8)


@pytest.mark.parametrize("n", [1, 7, 8, 9, 63, 1000, 6001])
@pytest.mark.parametrize("etype", [0, 1, 2])
def test_simd_error_functions_bitexact(pkg, n, etype):
    M = pkg.lib()
    u, h = _points(n, 3 + n)
    ref = np.zeros(n)
    assert M.mods_test_host_errfn(etype, P(u), n, P(h), 0, P(ref)) == 0
    assert np.isfinite(ref).all() and ref.max() > 0
    ran = 0
    for lanes in LANES:
        out = np.full(n, -1.0)
        rc = M.mods_test_host_errfn(etype, P(u), n, P(h), lanes, P(out))
        if rc != 0:
            continue   # this CPU lacks the width
        ran += 1
        assert np.array_equal(out.view(np.uint64), ref.view(np.uint64)), (lanes, etype)
    assert ran >= 1
    if refdeg.available():
        R = refdeg.lib()
        pool = np.arange(n, dtype=np.int32)
        Z = np.zeros(18 * n)
        R.lin_hg(P(u), P(Z), P(pool), n)
        r = np.zeros(n)
        getattr(R, ("HDs", "HDsSym", "HDsSymMax")[etype])(P(Z), P(u), P(h), P(r), n)
        assert np.array_equal(r.view(np.uint64), ref.view(np.uint64))


@pytest.mark.parametrize("n", [5, 12, 13, 200, 5000])
def test_moment_matrix_folded_bitexact(pkg, n):
    M = pkg.lib()
    u, _ = _points(6000, 11)
    g = np.random.default_rng(n)
    inl = np.ascontiguousarray(g.permutation(6000)[:n].astype(np.int32))
    a, b = np.zeros(81), np.zeros(81)
    assert M.mods_test_host_cov(P(u), P(inl), n, 1, P(a)) == 0
    assert M.mods_test_host_cov(P(u), P(inl), n, 0, P(b)) == 0
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert np.array_equal(a.reshape(9, 9), a.reshape(9, 9).T)
    Ha, Hb = np.zeros(9), np.zeros(9)
    assert M.mods_test_host_u2h(P(u), P(inl), n, 1, P(Ha)) == 0
    assert M.mods_test_host_u2h(P(u), P(inl), n, 0, P(Hb)) == 0
    assert np.array_equal(Ha.view(np.uint64), Hb.view(np.uint64)) and np.abs(Ha).max() > 0
    if refdeg.available():
        R = refdeg.lib()
        A1, A2 = np.zeros(3), np.zeros(3)
        R.normu(P(u), P(inl), n, P(A1), P(A2))
        Z = np.zeros(18 * n)
        R.lin_hgN(P(u), P(Z), P(inl), n, P(A1), P(A2))
        c = np.zeros(81)
        R.cov_mat(P(c), P(Z), 2 * n, 9)
        assert np.array_equal(c.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("th", [0.0, 4.0, 16.0, 128.0])
def test_inlidxs_gain_pass_bitexact(pkg, th):
    M = pkg.lib()
    n = 4099
    u, h = _points(n, 5)
    err = np.zeros(n)
    assert M.mods_test_host_errfn(0, P(u), n, P(h), 0, P(err)) == 0
    err[::97] = th * 9 / 4           # the boundary of the truncation
    err[1::97] = th
    want_inl, I0, J0 = np.zeros(n, np.int32), C.c_uint(), C.c_double()
    assert M.mods_test_host_inlidxs(P(err), n, C.c_double(th), 0, P(want_inl), C.byref(I0), C.byref(J0)) == 0
    for lanes in LANES:
        inl, I, J = np.zeros(n, np.int32), C.c_uint(), C.c_double()
        if M.mods_test_host_inlidxs(P(err), n, C.c_double(th), lanes, P(inl), C.byref(I), C.byref(J)) != 0:
            continue
        assert I.value == I0.value and np.float64(J.value).view(np.uint64) == np.float64(J0.value).view(np.uint64)
        assert np.array_equal(inl[:I.value], want_inl[:I0.value])
    # the list-only form the wide-threshold sets use: the same count and list, no sum
    inl, I, J = np.zeros(n, np.int32), C.c_uint(), C.c_double(-1.0)
    assert M.mods_test_host_inlidxs(P(err), n, C.c_double(th), -1, P(inl), C.byref(I), C.byref(J)) == 0
    assert I.value == I0.value and J.value == 0.0 and np.array_equal(inl[:I.value], want_inl[:I0.value])
    if refdeg.available():
        R = refdeg.lib()
        R.inlidxs.restype = refdeg.Score
        inl = np.zeros(n, np.int32)
        S = R.inlidxs(P(err), n, C.c_double(th), P(inl))
        assert S.I == I0.value and np.float64(S.J).view(np.uint64) == np.float64(J0.value).view(np.uint64)
        assert np.array_equal(inl[:S.I], want_inl[:S.I])


@pytest.mark.parametrize("n", [1, 9, 1000, 6001])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_simd_epipolar_error_functions_bitexact(pkg, n, mode):
    """FDs / FDsSym / exFDs / exFDsSym over all correspondences (Ftools.c:94-209): every SIMD width gives the scalar bits,
    and the scalar loops give the reference's (oracle/_ref)."""
    import fsynth
    M = pkg.lib()
    u, _, _ = fsynth.two_view(max(n, 16), 0.6, 0.0, 0.5, seed=4 + n)
    u = np.ascontiguousarray(u[:n])
    F = np.ascontiguousarray(np.random.default_rng(n + mode).normal(0, 1, 9) * [1e-6, 1e-6, 1e-3, 1e-6, 1e-6, 1e-3, 1e-3, 1e-3, 1.0])
    p0, w0 = np.zeros(n), np.zeros(n)
    assert M.mods_test_host_fds(mode, P(u), n, P(F), 0, P(p0), P(w0)) == 0
    assert np.isfinite(p0).all()
    ran = 0
    for lanes in LANES:
        p1, w1 = np.full(n, -1.0), np.full(n, -1.0)
        if M.mods_test_host_fds(mode, P(u), n, P(F), lanes, P(p1), P(w1)) != 0:
            continue
        ran += 1
        assert np.array_equal(p1.view(np.uint64), p0.view(np.uint64)), (lanes, mode)
        if mode >= 2:
            assert np.array_equal(w1.view(np.uint64), w0.view(np.uint64)), (lanes, mode)
    assert ran >= 1
    if refdeg.available():
        R = refdeg.lib()
        r, rw = np.zeros(n), np.zeros(n)
        if mode == 0:
            R.FDs(P(u), P(F), P(r), n)
        elif mode == 1:
            R.FDsSym(P(u), P(F), P(r), n)
        elif mode == 2:
            R.exFDs(P(u), P(F), P(r), P(rw), n)
        else:
            R.exFDsSym(P(u), P(F), P(r), P(rw), n)
        assert np.array_equal(r.view(np.uint64), p0.view(np.uint64))
        if mode >= 2:
            assert np.array_equal(rw.view(np.uint64), w0.view(np.uint64))


@pytest.mark.parametrize("n", [9, 14, 200, 5000])
@pytest.mark.parametrize("weighted", [False, True])
def test_least_squares_f_without_the_design_matrix_bitexact(pkg, n, weighted):
    """u2f / u2fw (Ftools.c:302-405): the moment matrix accumulated straight from the rows (cov_fmN) gives the bits of
    lin_fmN + weights + cov_mat."""
    import fsynth
    M = pkg.lib()
    u, _, _ = fsynth.two_view(6000, 0.6, 0.0, 0.5, seed=9)
    u = np.ascontiguousarray(u)
    g = np.random.default_rng(n)
    idx = np.ascontiguousarray(g.permutation(6000)[:n].astype(np.int32))
    w = np.ascontiguousarray(g.uniform(0.5, 1.5, 6000))
    Fa, Fb = np.zeros(9), np.zeros(9)
    M.mods_test_u2f_form(P(u), P(idx), n, P(w) if weighted else None, 1, P(Fa))
    M.mods_test_u2f_form(P(u), P(idx), n, P(w) if weighted else None, 0, P(Fb))
    assert np.array_equal(Fa.view(np.uint64), Fb.view(np.uint64)) and np.abs(Fa).max() > 0


@pytest.mark.parametrize("n", [9, 200, 21000])
@pytest.mark.parametrize("weighted", [False, True])
def test_moment_matrix_across_vector_lanes_bitexact(pkg, n, weighted):
    """The 45 ordered sums of the least-squares F's moment matrix side by side in vector lanes (SimdOps::cov_fm_all: what the
    degenerate branch of DEGENSAC spends its time in on a large planar pair) give the scalar loop's bits at 1, 4 and 8 lanes."""
    import fsynth
    M = pkg.lib()
    u, _, _ = fsynth.two_view(30000, 0.6, 0.0, 0.5, seed=11)
    u = np.ascontiguousarray(u)
    g = np.random.default_rng(n)
    idx = np.ascontiguousarray(g.permutation(30000)[:n].astype(np.int32))
    w = np.ascontiguousarray(g.uniform(0.5, 1.5, 30000))
    ref = np.zeros(81)
    assert M.mods_test_cov_fm(P(u), P(idx), n, P(w) if weighted else None, 0, P(ref)) == 0
    assert np.abs(ref).max() > 0 and np.array_equal(ref.reshape(9, 9), ref.reshape(9, 9).T)
    ran = 0
    for lanes in LANES:
        out = np.zeros(81)
        if M.mods_test_cov_fm(P(u), P(idx), n, P(w) if weighted else None, lanes, P(out)) != 0:
            continue            # this CPU lacks the instruction set of that lane count
        ran += 1
        assert np.array_equal(out.view(np.uint64), ref.view(np.uint64)), lanes
    assert ran >= 1


@pytest.mark.parametrize("n", [5, 12, 13, 200, 5000])
def test_moment_sums_at_every_lane_count(pkg, n):
    """the 30 folded sums through the SIMD table (8 lanes: four vectors of running sums) = the scalar folded sums = lin_hgN + cov_mat"""
    M = pkg.lib()
    u, _ = _points(6000, 17)
    g = np.random.default_rng(100 + n)
    inl = np.ascontiguousarray(g.permutation(6000)[:n].astype(np.int32))
    want = np.zeros(81)
    assert M.mods_test_host_cov(P(u), P(inl), n, 1, P(want)) == 0
    ran = 0
    for form in (0, 2, 101, 104, 108):
        got = np.full(81, -1.0)
        if M.mods_test_host_cov(P(u), P(inl), n, form, P(got)) != 0:
            assert form >= 100      # this CPU lacks the width
            continue
        ran += 1
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), form
    assert ran >= 3


def _frames(u, seed):
    """14 doubles per correspondence as H_LAF_check reads them (matching.cpp:250-308): x, y, a11, a12, a21, a22, s per image"""
    g = np.random.default_rng(seed)
    n = len(u)
    laf = np.zeros((n, 14))
    for base, cols in ((0, (0, 1)), (7, (3, 4))):
        laf[:, base:base + 2] = u[:, cols]
        A = np.eye(2) + g.normal(0, 0.25, (n, 2, 2))
        laf[:, base + 2:base + 6] = A.reshape(n, 4)
        laf[:, base + 6] = g.uniform(1.5, 30.0, n)
    return np.ascontiguousarray(laf)


@pytest.mark.parametrize("n,outliers,coef", [(9, 0.0, 12.0), (40, 0.3, 12.0), (1003, 0.3, 12.0), (6001, 0.5, 3.0), (300, 0.3, 0.0), (64, 0.95, 12.0)])
def test_checks_behind_the_homography_lanes_wide(pkg, n, outliers, coef):
    """NaiveHCheck + H_LAF_check over structure-of-arrays copies of the inliers keep exactly the scalar statement's survivors"""
    M = pkg.lib()
    M.mods_test_host_hchecks.restype = C.c_int
    u, h = _points(n, 31 + n, outliers=outliers)
    laf = _frames(u, n)
    err = np.zeros(n)
    assert M.mods_test_host_errfn(0, P(u), n, P(h), 0, P(err)) == 0
    inl = np.ascontiguousarray((err <= 16.0).astype(np.uint8))
    par = pkg.RansacParams.default()
    par.HLAFCoef = coef
    want_mask, want_H = np.zeros(n, np.uint8), np.zeros(9)
    want_n = M.mods_test_host_hchecks(P(u), P(laf), n, P(inl), P(h), C.byref(par), 0, P(want_mask), P(want_H))
    assert want_n >= 0 and want_n == int(want_mask.sum())
    if outliers < 0.9 and n >= 40:
        assert 8 <= want_n <= int(inl.sum())
        if 0 < coef < 12:
            assert want_n < int(inl.sum())       # the frame check does reject something here
    ran = 0
    for lanes in LANES:
        mask, H = np.full(n, 7, np.uint8), np.zeros(9)
        got = M.mods_test_host_hchecks(P(u), P(laf), n, P(inl), P(h), C.byref(par), lanes, P(mask), P(H))
        if got < 0:
            continue
        ran += 1
        assert got == want_n and np.array_equal(mask, want_mask) and np.array_equal(H.view(np.uint64), want_H.view(np.uint64)), lanes
    assert ran >= 1
