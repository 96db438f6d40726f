#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run in the build container, where
/root/reference and oracle/_ref exist).

  ransac_h.npz   inputs + outputs of the REFERENCE's exp_ransacHcustom (degensac compiled from
                 /root/reference by oracle/ref.mk, seed pinned through oracle/ref_shim.c) on seeded
                 synthetic correspondence sets
  glibc_rand.npz first 1000 outputs of srand(seed); rand() for seeds {1, 42, 12345} from this libc
  ransac_f.npz   the same for the REFERENCE's exp_ransacFcustom (DEGENSAC): general and plane-dominated two-view
                 scenes, Sampson / symmetric error, with and without the symmetric check
  stages.npz     small inputs/outputs of the CPU oracle (regression pins for every stage; the
                 reference itself cannot produce them: it needs OpenCV)
  views.npz      view-synthesis geometry (sizes, H, warp matrices, sigmas; plain double + libm) for a table of
                 (w, h, tilt, phi, zoom), one synthesised view, and the per-view region list of the oracle
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc      # noqa: E402
import refdeg   # noqa: E402
import synth    # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def make_corr(n, inlier_ratio, noise, seed, w=1000.0):
    rng = np.random.default_rng(seed)
    H = np.array([[1.05, 0.08, 12.0], [-0.06, 0.97, -7.0], [8e-5, -4e-5, 1.0]])
    x1 = rng.uniform(0, w, (n, 2))
    p = np.c_[x1, np.ones(n)] @ H.T
    x2 = p[:, :2] / p[:, 2:] + rng.normal(0, noise, (n, 2))
    n_out = n - int(round(n * inlier_ratio))
    idx = rng.permutation(n)[:n_out]
    x2[idx] = rng.uniform(0, w, (n_out, 2))
    return np.c_[x1, np.ones(n), x2, np.ones(n)]


def gen_ransac():
    out = {}
    cases = [(12, 1.0, 0.1), (20, 0.8, 0.5), (21, 0.6, 0.5), (100, 0.5, 0.5), (500, 0.5, 0.5), (500, 0.25, 1.0),
             (1000, 0.1, 0.5), (2000, 0.3, 0.8)]
    meta = []
    for ci, (n, ratio, noise) in enumerate(cases):
        u = make_corr(n, ratio, noise, 1000 + ci)
        out["u_%d" % ci] = u
        for err, sym in (("sampson", 1), ("symm_sum", 1), ("symm_max", 0)):
            for seed in (12345, 7):
                ms = 1000 if n <= 20 else 20000
                r = refdeg.ransac_h(u, 16.0, max_sam=ms, err=err, sym_check=sym, seed_time=seed)
                key = "%d_%s_%d_%d" % (ci, err, sym, seed)
                out["inl_" + key] = r["inl"]
                out["H_" + key] = r["H"]
                out["stat_" + key] = np.array([r["I"], r["samples"], r["lo"], r["rej"]], np.int64)
                out["J_" + key] = np.array([r["J"]])
                meta.append(key)
    out["keys"] = np.array(meta)
    np.savez_compressed(os.path.join(GOLD, "ransac_h.npz"), **out)


def gen_rand():
    libc = C.CDLL(None)
    out = {}
    for seed in (1, 42, 12345):
        libc.srand(seed)
        out["s%d" % seed] = np.array([libc.rand() for _ in range(1000)], np.int32)
    np.savez_compressed(os.path.join(GOLD, "glibc_rand.npz"), **out)


def gen_stages():
    out = {}
    img = synth.texture(96, 80, seed=3)
    out["img"] = img.astype(np.uint8)
    out["blur_1p2263"] = orc.gauss_blur(img, 1.2263)
    out["resp"] = orc.hessian_response(img, 2.56)
    out["half"] = orc.resize_half(img[:67, :95])
    big = synth.texture(400, 300, seed=11)
    out["big"] = big.astype(np.uint8)
    keys = orc.detect_hessian_affine(big)
    out["keys"] = keys
    regs, nd = orc.detect_describe(big)
    out["regions"] = regs
    a, b, _ = synth.pair(400, 300, seed=4)
    out["pair_a"] = a.astype(np.uint8); out["pair_b"] = b.astype(np.uint8)
    ra, _ = orc.detect_describe(a); rb, _ = orc.detect_describe(b)
    tc = orc.match_fginn(ra, rb)
    out["tent"] = tc
    out["tent_unique"] = orc.duplicate_filter(tc, ra, rb)
    np.savez_compressed(os.path.join(GOLD, "stages.npz"), **out)


def gen_ransac_f():
    import fsynth
    out, meta = {}, []
    cases = [(60, 0.9, 0.0, 0.3), (200, 0.7, 0.6, 0.5), (500, 0.5, 0.0, 0.5), (500, 0.5, 0.8, 0.5), (1000, 0.35, 0.5, 0.7), (400, 0.6, 1.0, 0.5)]
    for ci, (n, ratio, plane, noise) in enumerate(cases):
        u, _, _ = fsynth.two_view(n, ratio, plane, noise, seed=3000 + ci)
        out["u_%d" % ci] = u
        for err, sym in (("sampson", 0), ("sampson", 1), ("symm", 1)):
            for seed in (12345, 7):
                r = refdeg.ransac_f(u, 16.0, max_sam=20000, err=err, sym_check=sym, seed_time=seed)
                key = "%d_%s_%d_%d" % (ci, err, sym, seed)
                out["inl_" + key] = r["inl"]
                out["F_" + key] = r["F"]
                out["stat_" + key] = np.array([r["I"], r["samples"], r["lo"], r["Ih"]], np.int64)
                out["hist_" + key] = np.flatnonzero(r["hist"]).astype(np.int32)
                out["histv_" + key] = r["hist"][np.flatnonzero(r["hist"])].astype(np.int32)
                meta.append(key)
    out["keys"] = np.array(meta)
    np.savez_compressed(os.path.join(GOLD, "ransac_f.npz"), **out)


def gen_views():
    import math
    out = {}
    table = []
    for (w, h) in ((800, 640), (1920, 1080), (333, 517)):
        for tilt, phi in ((1.0, 0.0), (2.0, 0.0), (4.0, math.pi / 2), (6.0, math.pi / 3), (8.0, 3 * math.pi / 4), (-4.0, 0.0), (2.0, 2.5), (3.0, 0.1)):
            for zoom in (1.0, 0.5, 0.25):
                g = orc.view_geometry(w, h, tilt, phi, zoom, 0.2)
                table.append([w, h, tilt, phi, zoom, g.identity, g.w_rot, g.h_rot, g.w_new, g.h_new, g.ksize_x, g.ksize_y, g.sigma_x, g.sigma_y]
                             + list(g.H) + list(g.warpRot) + list(g.warpTilt))
    out["geometry"] = np.array(table, np.float64)
    img = synth.texture(240, 180, seed=9)
    out["img"] = img.astype(np.uint8)
    px, g = orc.synth_view(img, 4.0, math.pi / 3, 1.0, 0.2, 1)
    out["view_4_60"] = px
    reg, det, nd = orc.detect_describe_view(px, np.array(g.H), 240, 180)
    out["view_regions"] = reg
    np.savez_compressed(os.path.join(GOLD, "views.npz"), **out)


if __name__ == "__main__":
    gen_ransac(); gen_rand(); gen_stages(); gen_ransac_f(); gen_views()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))
