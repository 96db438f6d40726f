"""CPU tests of the host-side pieces of the F-matrix path (7-point solver, least-squares F, plane-
degeneracy test, homography LO, plane-and-parallax search) against the reference's own degensac
functions (oracle/_ref), through libmodsgpu's host-only self-test hooks (no device needed)."""
import ctypes as C

import numpy as np
import pytest

import fsynth
import refdeg

pytestmark = pytest.mark.skipif(not refdeg.available(), reason="oracle/_ref not built")


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def _normed(F):
    F = np.asarray(F, float).ravel()
    F = F / np.linalg.norm(F)
    return F * np.sign(F[np.argmax(np.abs(F))])


@pytest.fixture(scope="module")
def libs(pkg):
    R = refdeg.lib()
    M = pkg.lib()
    R.rroots3.restype = C.c_int
    R.innerH.restype = C.c_uint
    R.rFtH.restype = C.c_uint
    M.mods_test_rfth2.restype = C.c_uint
    M.mods_test_inner_h2.restype = C.c_uint
    M.mods_test_inner_h.restype = C.c_uint
    M.mods_test_rfth.restype = C.c_uint
    return R, M


def _ref_seven(R, u7):
    idx = np.arange(7, dtype=np.int32)
    Z = np.zeros(63)
    R.lin_fm(P(u7), P(Z), P(idx), 7)
    A = np.zeros(81)
    A[:63] = Z.reshape(9, 7).T.ravel()
    sol = np.zeros(81)
    nb = np.zeros(18, np.int32)
    if R.nullspace(P(A), P(sol), 9, P(nb)) != 2:
        return -1, []
    poly, roots = np.zeros(4), np.zeros(3)
    f1, f2 = sol[:9].copy(), sol[9:18].copy()
    R.slcm(P(f1), P(f2), P(poly))
    n = R.rroots3(P(poly), P(roots))
    return n, [f1 * roots[i] + f2 * (1 - roots[i]) for i in range(n)]


def test_least_squares_f(libs):
    R, M = libs
    u, tin, _ = fsynth.two_view(400, 0.6, 0.0, 0.5, seed=1)
    idx_all = np.where(tin)[0].astype(np.int32)
    w = np.random.default_rng(0).uniform(0.5, 1.5, len(u))
    buf = np.zeros(18 * len(u) + 100)
    for n in (8, 9, 14, 60, 200):      # 8 = null vector of the raw design matrix, > 8 = normalised moment matrix
        idx = np.ascontiguousarray(idx_all[:n])
        for weighted in (False, True):
            Fr, Fm = np.zeros(9), np.zeros(9)
            if weighted:
                R.u2fw(P(u), P(idx), P(w), n, P(Fr), P(buf))
            else:
                R.u2f(P(u), P(idx), n, P(Fr), P(buf))
            M.mods_test_u2f(P(u), P(idx), n, P(w) if weighted else None, P(Fm))
            assert np.max(np.abs(_normed(Fr) - _normed(Fm))) < 1e-10, (n, weighted)


def test_seven_point_and_checksample(libs):
    R, M = libs
    n_deg = 0
    for seed in range(200):
        u, tin, pl = fsynth.two_view(200, 0.7, 0.6 if seed % 2 else 0.0, 0.5, seed=seed)
        rng = np.random.default_rng(seed)
        pool = np.where(pl)[0] if (seed % 2 and seed % 3) else np.where(tin)[0]
        u7 = np.ascontiguousarray(u[rng.choice(pool, 7, replace=False)])
        n, Fr = _ref_seven(R, u7)
        F27 = np.zeros(27)
        assert M.mods_test_seven_point(P(u7), P(F27)) == n
        for i in range(max(n, 0)):
            assert np.max(np.abs(_normed(Fr[i]) - _normed(F27[9 * i:9 * i + 9]))) < 1e-7
            Hr, Hm = np.zeros(9), np.zeros(9)
            Fi = Fr[i].copy()
            a = R.checksample(P(Fi), P(u7), C.c_double(48.0), P(Hr))
            b = M.mods_test_checksample(P(Fr[i]), P(u7), C.c_double(48.0), P(Hm))
            assert a == b, (seed, i)
            if a:
                n_deg += 1
                assert np.max(np.abs(_normed(Hr) - _normed(Hm))) < 1e-7, (seed, i)
    assert n_deg > 50      # CCMATH's unsorted singular values are exercised (see svd_v_unsorted)


def test_homography_lo_and_plane_parallax(libs):
    R, M = libs
    libc = C.CDLL(None)
    th = 4.0
    # seeds 0-5: the mixed scene; 6-11: larger lists whose plane holds 50 % ... 97 % of the inliers (few off-plane points: the
    # search spends its whole budget; many: the first estimations cut the budget and later triggers never happen)
    paths, dropped, rounds = set(), 0, 0
    cases = [(seed, 300 + 60 * seed, 0.7) for seed in range(6)] + [(6 + k, 1500 + 200 * k, pr) for k, pr in enumerate((0.5, 0.5, 0.9, 0.9, 0.97, 0.97))]
    for seed, n, plane_ratio in cases:
        u, _, pl = fsynth.two_view(n, 0.7, plane_ratio, 0.5, seed=seed)
        idx = np.where(pl)[0][:12]
        A = []
        for i in idx:
            x, y, _, X, Y, _ = u[i]
            A.append([x, y, 1, 0, 0, 0, -X * x, -X * y, -X])
            A.append([0, 0, 0, x, y, 1, -Y * x, -Y * y, -Y])
        H12 = np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)
        Hd = np.zeros(n)
        for H0 in (H12.T.ravel().copy(), H12.ravel().copy(), np.linalg.inv(H12).T.ravel().copy(), np.linalg.inv(H12).ravel().copy()):
            R.dHDs(P(H0), P(u), n, P(Hd), None, None)
            if (Hd < 3 * th).sum() > 20:
                break
        Hr, Hm = H0.copy(), H0.copy()
        inl_r, inl_m = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        pool, buf, bufP = np.zeros(n, np.int32), np.zeros(18 * n + 100), np.zeros(n, np.int32)
        libc.srand(1000 + seed)
        Ir = R.innerH(P(Hr), P(u), n, C.c_double(16 * th), 10, P(inl_r), P(pool), P(buf))
        next_r = libc.rand()
        Im = M.mods_test_inner_h(1000 + seed, P(Hm), P(u), n, C.c_double(16 * th), 10, P(inl_m))
        assert Ir == Im and np.array_equal(inl_r, inl_m)
        assert np.max(np.abs(_normed(Hr) - _normed(Hm))) < 1e-9
        # production form (host SIMD evaluation, the ten repetitions side by side when the pool has threads): same bits as the
        # scalar form, and the generator where the reference leaves it
        Hs, inl_s, next_m, path = H0.copy(), np.zeros(n, np.uint8), C.c_int(0), C.c_int(-1)
        Is = M.mods_test_inner_h2(1000 + seed, P(Hs), P(u), n, C.c_double(16 * th), 10, P(inl_s), 1, C.byref(next_m), C.byref(path))
        assert Is == Ir and np.array_equal(inl_s, inl_r) and np.array_equal(Hs.view(np.uint64), Hm.view(np.uint64))
        assert next_m.value == next_r
        paths.add(path.value)
        Fr, Fm = np.zeros(9), np.zeros(9)
        libc.srand(2000 + seed)
        Jr = R.rFtH(P(u), P(inl_r), C.c_double(th), P(Hr), n, P(Fr), P(bufP), P(buf))
        next_r = libc.rand()       # where the reference's generator stands after the search
        Jm = M.mods_test_rfth(2000 + seed, P(u), P(inl_r), C.c_double(th), P(Hr), n, P(Fm))
        assert Jr == Jm and Jr > 50
        assert np.max(np.abs(_normed(Fr) - _normed(Fm))) < 1e-9
        # the production form: host SIMD evaluation, inner estimations of several triggers side by side on the pool's threads
        # (ransac_pool.hpp) - the same F to the bit as the scalar one-evaluation-at-a-time form, and the same generator state
        Fs, next_m, prof = np.zeros(9), C.c_int(0), np.zeros(10)
        Js = M.mods_test_rfth2(2000 + seed, P(u), P(inl_r), C.c_double(th), P(Hr), n, P(Fs), 1, C.byref(next_m), P(prof))
        assert Js == Jr and np.array_equal(Fs.view(np.uint64), Fm.view(np.uint64))
        assert next_m.value == next_r
        dropped += int(prof[8])
        rounds = max(rounds, int(prof[9]))
    # with more than one pool thread the side-by-side form is what ran (and, on these scenes, was never redone)
    assert paths == ({1} if M.mods_ransac_host_threads() > 1 else {0}), paths
    # the search ran ahead of inner estimations whose triggers then never happened, and needed more than one round of triggers
    assert dropped > 0 and rounds > 1, (dropped, rounds)


def test_inner_h_weak_plane_takes_both_paths(libs):
    """innerH on a plane with 15-20 supporting points: iterH often leaves early (fewer than 4 points within the shrinking
    threshold), so the generator does not advance by the usual number of draws - the side-by-side form notices and the call is
    redone by the one-thread loop (path 2); either way count, mask, H and the generator agree with the reference's build."""
    R, M = libs
    libc = C.CDLL(None)
    paths = set()
    for seed in range(24):
        g = np.random.default_rng(seed)
        n, k = 400, 30
        u = np.ones((n, 6))
        u[:, 0:2] = g.uniform(0, 1000, (n, 2))
        u[:, 3:5] = g.uniform(0, 1000, (n, 2))
        u[:k, 3:5] = u[:k, 0:2] + g.normal(0, 6.0, (k, 2))
        u = np.ascontiguousarray(u)
        H0 = np.eye(3).ravel().copy()
        Hd = np.zeros(n)
        R.dHDs(P(H0), P(u), n, P(Hd), None, None)
        th = float(np.sort(Hd)[14]) * 1.0001
        Hr, Hs = H0.copy(), H0.copy()
        inl_r, inl_s = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        pool, buf = np.zeros(n, np.int32), np.zeros(18 * n + 100)
        libc.srand(77 + seed)
        Ir = R.innerH(P(Hr), P(u), n, C.c_double(th), 10, P(inl_r), P(pool), P(buf))
        next_r = libc.rand()
        next_m, path = C.c_int(0), C.c_int(-1)
        Is = M.mods_test_inner_h2(77 + seed, P(Hs), P(u), n, C.c_double(th), 10, P(inl_s), 1, C.byref(next_m), C.byref(path))
        assert Ir == Is and np.array_equal(inl_r, inl_s) and next_m.value == next_r, seed
        assert np.max(np.abs(_normed(Hr) - _normed(Hs))) < 1e-9, seed
        paths.add(path.value)
    assert paths == ({1, 2} if M.mods_ransac_host_threads() > 1 else {0}), paths
