"""CPU tests (no GPU): the C ABI library loads and exports every declared symbol; host-side logic of
libmodsgpu (restated glibc generator, duplicate filter, argument / no-device error paths)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+;", "", src, flags=re.S)
    src = re.sub(r"typedef[^;{}]*;", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    names = re.findall(r"\b([A-Za-z_]\w*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",) and not n.endswith("Ptr")))


def test_every_declared_symbol_is_exported(pkg):
    lib = pkg.lib()
    names = _declared_functions("mods_hip.h") + _declared_functions("mods_degensac.h")
    assert "mods_match_pair_dev" in names and "exp_ransacHcustom" in names and len(names) > 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_device_fails_loudly(pkg):
    lib = pkg.lib()
    if lib.mods_device_count() > 0:
        return   # GPU box: covered by the -m gpu tests
    h = C.c_void_p()
    rc = lib.mods_ctx_create(0, 640, 480, 1, C.byref(h))
    assert rc == -1 and b"no CPU path" in lib.mods_last_error()


def test_glibc_generator_is_bit_exact(pkg):
    lib = pkg.lib()
    libc = C.CDLL(None)
    g = np.load(os.path.join(ROOT, "tests", "golden", "glibc_rand.npz"))
    for seed in (1, 42, 12345, 0, 2 ** 31 + 5, 4294967295):
        out = (C.c_int * 1000)()
        lib.mods_test_glibc_rand(C.c_uint(seed), 1000, out)
        libc.srand(C.c_uint(seed))
        want = [libc.rand() for _ in range(1000)]
        assert list(out) == want
        if "s%d" % seed in g:
            assert list(out) == list(g["s%d" % seed])


def test_duplicate_filter_matches_oracle(pkg):
    rng = np.random.default_rng(3)
    n = 1500
    q = np.zeros(n, orc.REGION_DTYPE); t = np.zeros(n, orc.REGION_DTYPE)
    cl = rng.integers(0, 300, n)                                # 300 clusters of near-identical correspondences
    cx, cy = rng.uniform(0, 800, 300), rng.uniform(0, 600, 300)
    q["x"] = cx[cl] + rng.uniform(-1.2, 1.2, n); q["y"] = cy[cl] + rng.uniform(-1.2, 1.2, n)
    t["x"] = 0.9 * cx[cl] + 30 + rng.uniform(-1.2, 1.2, n); t["y"] = 1.1 * cy[cl] - 10 + rng.uniform(-1.2, 1.2, n)
    tc = np.zeros(n, orc.TENT_DTYPE)
    tc["q"] = np.arange(n); tc["t"] = np.arange(n)
    perm = rng.permutation(n)
    tc = tc[perm]
    tc["ratio"] = np.round(rng.uniform(0.2, 0.8, n), 2)        # many equal keys: stable order matters
    tc["d1"] = rng.integers(100, 9000, n)
    u6 = np.c_[q["x"][tc["q"]], q["y"][tc["q"]], np.ones(n), t["x"][tc["t"]], t["y"][tc["t"]], np.ones(n)]
    q["s"] = np.round(rng.uniform(1.5, 40.0, n), 1) * rng.choice([-1.0, 1.0], n)   # biggerRegion sorts by |s|; equal keys exist
    laf = np.zeros((n, 14)); laf[:, 6] = q["s"][tc["q"]]
    for mode in (0, 1, 2, 3):
        for r in (2.0, 0.5, 7.0):
            want = orc.duplicate_filter(tc, q, t, r, mode)
            got, gu, gl = pkg.duplicate_filter(tc, u6, r, mode, laf)
            assert np.array_equal(gl[:, 6], q["s"][got["q"]])
            assert len(got) == len(want) < n
            for f in ("q", "t", "ratio", "d1"):
                assert np.array_equal(got[f], want[f])
            assert np.array_equal(gu[:, 0], q["x"][got["q"]])
    got, _ = pkg.duplicate_filter(tc, u6, 0.0, 1)
    assert len(got) == n                                        # r <= 0: no filtering (matching.cpp:2617)


def test_loransac_rejects_small_sets_without_a_gpu(pkg):
    u = np.random.default_rng(0).uniform(0, 100, (7, 6))
    mask, H, ninl, stats = pkg.loransac_h(u, None)
    assert ninl == 0 and not mask.any() and np.all(H == -1)     # tent_size < MIN_POINTS (matching.cpp:682-688)


def test_verification_without_a_device_is_an_error_not_an_abort(pkg):
    """The degensac entry points have no error channel (exp_ranH.c / exp_ranF.c signatures): a device failure inside them
    comes back from mods_loransac_h / mods_loransac_f as an error code, the process lives on, and nothing is computed on
    the CPU instead."""
    if pkg.lib().mods_device_count() > 0:
        return
    u = np.random.default_rng(0).uniform(0, 100, (40, 6))
    u[:, 2] = u[:, 5] = 1
    for fn in (pkg.loransac_h, pkg.loransac_f):
        with pytest.raises(pkg.ModsError, match="no CPU path"):
            fn(u, None)


def test_struct_layouts(pkg):
    assert pkg.REGION_DTYPE.itemsize == 208 and pkg.AFFKEY_DTYPE.itemsize == 88 and pkg.TENT_DTYPE.itemsize == 40
    assert orc.REGION_DTYPE == pkg.REGION_DTYPE and orc.TENT_DTYPE == pkg.TENT_DTYPE


def test_view_geometry_and_schedule_host_side(pkg):
    """mods_view_geometry / mods_view_schedule are host-only: checked here against the committed fixture
    (generated with the oracle) and against the view counts of the reference's MODS ladder (11 + 20)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "views.npz"))
    for row in g["geometry"]:
        w, h, tilt, phi, zoom = int(row[0]), int(row[1]), row[2], row[3], row[4]
        v = pkg.view_geometry(w, h, tilt, phi, zoom, 0.2)
        got = [v.identity, v.w_rot, v.h_rot, v.w_new, v.h_new, v.ksize_x, v.ksize_y, v.sigma_x, v.sigma_y] + list(v.H) + list(v.warpRot) + list(v.warpTilt)
        assert got == list(row[5:]), (w, h, tilt, phi, zoom)
    hist = []
    steps = pkg.iters_mods_steps()
    first = pkg.view_schedule(steps[0], hist)
    second = pkg.view_schedule(steps[1], hist)
    assert len(first) == 11 and len(second) == 20 and len(hist) == 31
    assert first[0] == (1.0, 1.0, 0.0) and all(t in (1, 2, 4, 6, 8) for _, t, _ in hist)
    assert len({(z, t, round(p, 9)) for z, t, p in hist}) == 31        # no view is scheduled twice
    neg = pkg.view_schedule(pkg.LadderStep.make((3,), -360.0), [])     # negative density: vertical + horizontal tilt, no rotation
    assert neg == [(1.0, -3.0, 0.0), (1.0, 3.0, 0.0)]


@pytest.mark.parametrize("err_type,err_name", [(0, "sampson"), (1, "symm_max"), (2, "symm_sum")])
def test_hmatrix_filter_matches_the_reference_error_functions(pkg, err_type, err_name):
    """mods_hmatrix_filter (HMatrixFiltering, matching.cpp:917-1012, the verification of ver_type 1) against the reference's own
    HDs / HDsSymMax / HDsSym (oracle/_ref): the same mask, points on both sides of the threshold."""
    import pipeline_oracle as po
    import refdeg
    if not refdeg.available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(17 + err_type)
    n = 4000
    H = np.array([[0.9, 0.12, 30.0], [-0.08, 1.05, -12.0], [1e-4, -6e-5, 1.0]])
    x1 = np.c_[rng.uniform(0, 800, n), rng.uniform(0, 600, n), np.ones(n)]
    p = x1 @ H.T
    x2 = p[:, :2] / p[:, 2:3] + rng.normal(0, 1.0, (n, 2)) * rng.choice([0.3, 3.0, 30.0], (n, 1))
    u6 = np.ascontiguousarray(np.c_[x1[:, 0], x1[:, 1], np.ones(n), x2[:, 0], x2[:, 1], np.ones(n)])
    par = pkg.RansacParams.default()
    par.errorType = err_type
    got, n_true = pkg.hmatrix_filter(u6, H, par)
    want = po.hmatrix_filter(u6, H, 4.0, err_name)
    assert n_true == int(got.sum()) and 0.2 * n < n_true < 0.9 * n
    assert np.array_equal(got, want)
    assert pkg.hmatrix_filter(u6[:0], H, par)[1] == 0


def test_matrix_core_kernels_own_their_simds(pkg):
    """A wave that issues independent MFMA chains makes double-precision VALU results of OTHER waves on the same SIMD go wrong
    (DESIGN.md "The matcher and its neighbours"; tools/ubench/mfma_aggr.hip next to tools/ubench/spin_victim.hip reproduces it
    with nothing but MFMAs, and not at all when the MFMA waves allocate the whole register file of their SIMD).  So every kernel
    of the built libmodsgpu.so that contains an MFMA instruction must leave no room for foreign waves: the waves of ONE workgroup
    fill the 512-entry file of each SIMD (register allocation x waves of the workgroup per SIMD = 512).  Read from the code
    objects the library carries (tools/kernel_resources.py): a new matrix-core kernel, an edit or a compiler update that breaks
    the rule fails here, on the CPU."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    if not os.path.exists(os.path.join(kr.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    ks = kr.kernels(os.path.join(root, "mods-light-zmq_amd", "libmodsgpu.so"))
    names = {k["name"] for k in ks}
    for needed in ("match_nn1_kernel", "match_fginn_kernel", "gauss_blur_fast_kernel", "sift_wave2_kernel", "ransac_score_kernel"):
        assert any(needed in n for n in names), (needed, len(names))
    mfma = [k for k in ks if k["mfma"]]
    assert sorted(n for k in mfma for n in ("match_nn1_kernel", "match_fginn_kernel") if n in k["name"]) == ["match_fginn_kernel", "match_nn1_kernel"]
    for k in mfma:
        waves_of_wg_per_simd = k["wg_size"] // 64 // 4
        assert k["wg_size"] % 256 == 0 and waves_of_wg_per_simd >= 1, (k["name"], k["wg_size"])
        assert k["alloc"] * waves_of_wg_per_simd == 512, (k["name"], k["vgpr"], k["alloc"], k["wg_size"])
        assert k["scratch"] == 0, (k["name"], k["scratch"])
