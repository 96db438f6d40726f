"""CPU oracle for the whole pair path (test infrastructure): restated detect/describe/match/duplicate
filter + RANSAC from the reference's own degensac (oracle/_ref) when it is built."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import orc
import refdeg


def pmap(fn, items, threads=None):
    """Independent oracle calls side by side (ctypes releases the GIL; the oracle keeps no state between calls)."""
    items = list(items)
    threads = threads or min(len(items), os.cpu_count() or 1, 16)
    if threads <= 1:
        return [fn(x) for x in items]
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(fn, items))


def match_fginn_par(ra, rb, ratio=0.8, contrad=10.0, nn=50, chunks=None):
    """orc.match_fginn with the queries split over threads (every query is searched on its own); same list, same order."""
    chunks = chunks or max(1, min(os.cpu_count() or 1, 16, len(ra) // 512))
    cuts = [len(ra) * i // chunks for i in range(chunks + 1)]

    def part(i):
        tc = orc.match_fginn(ra[cuts[i]:cuts[i + 1]], rb, ratio, contrad, nn)
        tc["q"] += cuts[i]
        return tc
    return np.concatenate(pmap(part, range(chunks)))


def laf_of(ra, rb, tc):
    a, b = ra[tc["q"]], rb[tc["t"]]
    cols = []
    for r in (a, b):
        cols += [r["x"], r["y"], r["a11"], r["a12"], r["a21"], r["a22"], r["s"]]
    return np.stack(cols, axis=1) if len(tc) else np.zeros((0, 14))


def u6_of(ra, rb, tc):
    n = len(tc)
    return np.c_[ra["x"][tc["q"]], ra["y"][tc["q"]], np.ones(n), rb["x"][tc["t"]], rb["y"][tc["t"]], np.ones(n)]


def _invert3(S):
    S = S.ravel()
    d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6])
    if d == 0:
        return np.zeros(9)
    d = 1. / d
    return np.array([(S[4] * S[8] - S[5] * S[7]) * d, (S[2] * S[7] - S[1] * S[8]) * d, (S[1] * S[5] - S[2] * S[4]) * d,
                     (S[5] * S[6] - S[3] * S[8]) * d, (S[0] * S[8] - S[2] * S[6]) * d, (S[2] * S[3] - S[0] * S[5]) * d,
                     (S[3] * S[7] - S[4] * S[6]) * d, (S[1] * S[6] - S[0] * S[7]) * d, (S[0] * S[4] - S[1] * S[3]) * d])


def hds_sym_max(u, Hcol):
    """HDsSymMax (Htools.c:243-283) on rows of u[n,6]; Hcol = degensac's column-major h."""
    Hinv = np.array([Hcol[0], Hcol[3], Hcol[6], Hcol[1], Hcol[4], Hcol[7], Hcol[2], Hcol[5], Hcol[8]])
    H1 = np.linalg.inv(Hinv.reshape(3, 3)).ravel()
    out = []
    for p in u:
        a = H1[6] * p[0] + H1[7] * p[1] + H1[8]
        b = Hinv[6] * p[3] + Hinv[7] * p[4] + Hinv[8]
        xa = (H1[0] * p[0] + H1[1] * p[1] + H1[2]) / a
        ya = (H1[3] * p[0] + H1[4] * p[1] + H1[5]) / a
        d1 = (p[3] - xa) ** 2 + (p[4] - ya) ** 2
        xa = (Hinv[0] * p[3] + Hinv[1] * p[4] + Hinv[2]) / b
        ya = (Hinv[3] * p[3] + Hinv[4] * p[4] + Hinv[5]) / b
        d2 = (p[0] - xa) ** 2 + (p[1] - ya) ** 2
        out.append(max(d1, d2))
    return np.array(out)


def loransac_h(u6, laf, err_threshold=4.0, conf=0.99, max_samples=1000000, hlaf=12.0, sym=1, seed_time=12345):
    """LORANSACFiltering, matching.cpp:637-805 (useF = 0), on top of the reference's exp_ransacHcustom."""
    n = len(u6)
    H = -np.ones(9)
    mask = np.zeros(n, bool)
    if n < 8:
        return mask, H.reshape(3, 3), 0, [0, 0, 0]
    ms = 1000 if n <= 20 else max_samples
    r = refdeg.ransac_h(u6, err_threshold ** 2, conf, ms, "sampson", sym, seed_time)
    stats = [r["samples"], r["lo"], r["rej"]]
    Hl = r["H"]
    Ht = np.array([Hl[0], Hl[3], Hl[6], Hl[1], Hl[4], Hl[7], Hl[2], Hl[5], Hl[8]])
    Hinv = _invert3(Ht)
    if not np.any(Hinv != 0):
        return mask, H.reshape(3, 3), 0, stats
    H = Hinv
    cur = np.nonzero(r["inl"])[0]
    Hi = _invert3(Hinv)
    ok = 0
    for i in cur:
        p = u6[i]
        xa = (Hinv[0] * p[0] + Hinv[1] * p[1] + Hinv[2]) / (Hinv[6] * p[0] + Hinv[7] * p[1] + Hinv[8])
        ya = (Hinv[3] * p[0] + Hinv[4] * p[1] + Hinv[5]) / (Hinv[6] * p[0] + Hinv[7] * p[1] + Hinv[8])
        d1 = (p[3] - xa) ** 2 + (p[4] - ya) ** 2
        xa = (Hi[0] * p[3] + Hi[1] * p[4] + Hi[2]) / (Hi[6] * p[3] + Hi[7] * p[4] + Hi[8])
        ya = (Hi[3] * p[3] + Hi[4] * p[4] + Hi[5]) / (Hi[6] * p[3] + Hi[7] * p[4] + Hi[8])
        d2 = (p[0] - xa) ** 2 + (p[1] - ya) ** 2
        ok += (d1 <= 100.0) and (d2 <= 100.0)
    if ok < 8:
        cur = cur[:0]
    thr = 3.0 * hlaf * err_threshold
    good = []
    for i in cur:
        f = laf[i]
        u = np.zeros((3, 6))
        u[0] = [f[0], f[1], 1, f[7], f[8], 1]
        u[1] = [f[0] + 3.0 * f[3] * f[6], f[1] + 3.0 * f[5] * f[6], 1, f[7] + 3.0 * f[10] * f[13], f[8] + 3.0 * f[12] * f[13], 1]
        u[2] = [f[0] + 3.0 * f[2] * f[6], f[1] + 3.0 * f[4] * f[6], 1, f[7] + 3.0 * f[9] * f[13], f[8] + 3.0 * f[11] * f[13], 1]
        e = hds_sym_max(u, Hl)
        if not (np.sqrt(e[0] + e[1] + e[2]) > thr):
            good.append(i)
    if len(good) < 8:
        good = []
    mask[good] = True
    return mask, H.reshape(3, 3), len(good), stats


def fds(u, F):
    """FDs (Ftools.c:94-112) on rows of u[n,6]; F as degensac stores it."""
    rxc = F[0] * u[:, 3] + F[3] * u[:, 4] + F[6]
    ryc = F[1] * u[:, 3] + F[4] * u[:, 4] + F[7]
    rwc = F[2] * u[:, 3] + F[5] * u[:, 4] + F[8]
    r = u[:, 0] * rxc + u[:, 1] * ryc + rwc
    rx = F[0] * u[:, 0] + F[1] * u[:, 1] + F[2]
    ry = F[3] * u[:, 0] + F[4] * u[:, 1] + F[5]
    return r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry)


def loransac_f(u6, laf, err_threshold=4.0, conf=0.99, max_samples=1000000, laf_coef=2.0, sym=1, do_lo=1, seed_time=12345):
    """LORANSACFiltering, matching.cpp:711-726 + 804-816 (useF = 1), on top of the reference's exp_ransacFcustom."""
    n = len(u6)
    mask = np.zeros(n, bool)
    if n < 8:
        return mask, -np.ones(9), 0, [0, 0, 0]
    r = refdeg.ransac_f(u6, err_threshold ** 2, conf, max_samples, "sampson", sym, do_lo, 0, seed_time)
    stats = [r["samples"], r["lo"], r["Ih"]]
    F = r["F"]
    cur = np.nonzero(r["inl"])[0]
    thr = laf_coef * err_threshold
    good = []
    for i in cur:
        f = laf[i]
        u = np.zeros((3, 6))
        u[0] = [f[0], f[1], 1, f[7], f[8], 1]
        u[1] = [f[0] + 3.0 * f[3] * f[6], f[1] + 3.0 * f[5] * f[6], 1, f[7] + 3.0 * f[10] * f[13], f[8] + 3.0 * f[12] * f[13], 1]
        u[2] = [f[0] + 3.0 * f[2] * f[6], f[1] + 3.0 * f[4] * f[6], 1, f[7] + 3.0 * f[9] * f[13], f[8] + 3.0 * f[11] * f[13], 1]
        e = fds(u, F)
        if not (np.sqrt(e[0]) + np.sqrt(e[1]) + np.sqrt(e[2]) > thr):
            good.append(i)
    if len(good) < 8:
        good = []
    mask[good] = True
    return mask, F, len(good), stats


def hmatrix_filter(u6, H, err_threshold=4.0, err="sampson"):
    """HMatrixFiltering, matching.cpp:917-1012, on top of the reference's own HDs / HDsSymMax / HDsSym (oracle/_ref): the second
    image's point first, the column-major H, th = (float)(err_threshold^2).  H: row-major img1 -> img2."""
    import ctypes as C
    n = len(u6)
    if n == 0:
        return np.zeros(0, bool)
    u2 = np.ascontiguousarray(np.c_[u6[:, 3], u6[:, 4], np.ones(n), u6[:, 0], u6[:, 1], np.ones(n)], np.float64)
    M = np.asarray(H, np.float64).reshape(9)
    Hc = np.ascontiguousarray([M[0], M[3], M[6], M[1], M[4], M[7], M[2], M[5], M[8]], np.float64)
    d = np.zeros(n)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    Z = np.zeros(18 * n)
    p = np.ascontiguousarray(np.arange(n, dtype=np.int32))
    L = refdeg.lib()
    L.lin_hg(P(u2), P(Z), P(p), n)
    getattr(L, refdeg.ERR[err][0])(P(Z), P(u2), P(Hc), P(d), n)
    return d <= float(np.float32(err_threshold * err_threshold))


def match_pair_ground_truth(img1, img2, H_gt, both=True, seed_time=12345, ratio=0.8, regions=None):
    """One step of mods.cpp:202-383 with ver_type = GR_TRUTH (mods.cpp:290-320): HMatrixFiltering of the unique tentatives, and with
    doBothRANSACgroundTruth LORANSACFiltering + HMatrixFiltering of its inliers (the verified list)."""
    if regions is None:
        (ra, nd1), (rb, nd2) = pmap(orc.detect_describe, (img1, img2))
    else:
        (ra, nd1), (rb, nd2) = regions
    tc = match_fginn_par(ra, rb, ratio)
    un = orc.duplicate_filter(tc, ra, rb, 2.0, 1)
    u6, laf = u6_of(ra, rb, un), laf_of(ra, rb, un)
    true_all = hmatrix_filter(u6, H_gt)
    out = dict(n_tentatives=len(tc), n_unique=len(un), gt_true=int(true_all.sum()), gt_ransac_inliers=0, gt_true_of_ransac=0)
    if both:
        mask, _, ninl, _ = loransac_h(u6, laf, seed_time=seed_time)
        tr = hmatrix_filter(u6[mask], H_gt)
        out.update(gt_ransac_inliers=int(ninl), gt_true_of_ransac=int(tr.sum()), matches=u6[mask][tr][:, [0, 1, 3, 4]])
    else:
        out.update(matches=u6[true_all][:, [0, 1, 3, 4]])
    out["n_inliers"] = len(out["matches"])
    return out


def match_pair(img1, img2, seed_time=12345, ratio=0.8, use_f=False, dup_before_ransac=True, regions=None):
    """One step of mods.cpp:202-383 on identity views.  dup_before_ransac = False: [DuplicateFiltering] doBeforeRANSAC = 0,
    the verified list is de-duplicated after RANSAC (mods.cpp:357-368)."""
    if regions is None:
        (ra, nd1), (rb, nd2) = pmap(orc.detect_describe, (img1, img2))
    else:
        (ra, nd1), (rb, nd2) = regions
    tc = match_fginn_par(ra, rb, ratio)
    un = orc.duplicate_filter(tc, ra, rb, 2.0, 1) if dup_before_ransac else tc
    u6, laf = u6_of(ra, rb, un), laf_of(ra, rb, un)
    if use_f:
        mask, H, ninl, stats = loransac_f(u6, laf, seed_time=seed_time)
    else:
        mask, H, ninl, stats = loransac_h(u6, laf, seed_time=seed_time)
    matches = u6[mask][:, [0, 1, 3, 4]]
    if not dup_before_ransac:
        ver = orc.duplicate_filter(un[mask], ra, rb, 2.0, 1)
        matches = u6_of(ra, rb, ver)[:, [0, 1, 3, 4]]
        ninl = len(ver)
    return dict(n_detected=[nd1, nd2], n_described=[len(ra), len(rb)], n_tentatives=len(tc), n_unique=len(un),
                n_inliers=ninl, stats=stats, H=H, mask=mask, u6=u6, matches=matches, regions=(ra, rb))


def view_schedule(tilts, phi_base, history, scales=(1.0,)):
    """SetVSPars (synth-detection.cpp:191-322) for one step; history is extended in place."""
    import math
    eps1 = 0.01
    tmp = []
    for sc in scales:
        for t in tilts:
            if abs(t - 1) > eps1:
                n_rot = int(math.floor(180.0 * t / phi_base))
                dphi = math.pi / n_rot
                if n_rot < 0:
                    n_rot, dphi = 1, 0.0
                    tmp.append((sc, -t, 0.0))
                for r in range(n_rot):
                    tmp.append((sc, t, dphi * r))
            else:
                tmp.append((sc, t, 0.0))
    new = [v for v in tmp if not any(abs(v[0] - p[0]) <= eps1 and abs(v[1] - p[1]) <= eps1 and abs(v[2] - p[2]) <= eps1
                                     for p in history)]
    history.extend(new)
    return new


def match_ladder(img1, img2, steps, seed_time=12345, min_matches=15, init_sigma=0.2, ratio=0.8, half_orientation=False,
                 ratio_half=0.0, detectors=None, groups=None, group_pos=0):
    """mods.cpp:202-383: steps = [(tilts, phi_base), ...] for one HessianAffine detector, or `detectors` = a list (sorted by
    detector name, the bank's key order) of dicts {params, steps: [(tilts, phi_base) | None per step], ratio, ratio_half,
    half_orientation}.  half_orientation: the steps' descriptor lists name a Half* descriptor (DetectOrientation in doHalfSIFT
    mode for every descriptor); ratio_half > 0: HalfRootSIFT lists are built and matched as a second separate descriptor.
    Every (descriptor, detector) pair keeps its own tentative list, re-made in the steps that bring new views of the detector
    (MatchImgReps, correspondencebank.cpp:286-340) and joined in the bank's key order - HalfRootSIFT before RootSIFT, then by
    detector (GetCorresponcesVector, :114-148).  A detector dict may carry dist > 0 = DistanceThreshold of RootSIFT.
    groups: per step None or dict(dets=[detector indices in the order named], ratio, ratio_half): grouped matching
    (correspondencebank.cpp:245-285) - the detectors' regions joined into one query / train list per descriptor, the result
    kept as the bank's "Group" detector at position group_pos of the detector order."""
    h, w = img1.shape
    if detectors is None:
        detectors = [dict(params=None, steps=list(steps), ratio=ratio, ratio_half=ratio_half, half_orientation=half_orientation)]
    n_steps = max(len(d["steps"]) for d in detectors)
    st = [dict(banks=[[], []], banks_h=[[], []], history=[], tc=None, tch=None) for _ in detectors]
    out = None
    n_views = 0
    G = dict(tc=None, tch=None)
    for si in range(n_steps):
        for d, S in zip(detectors, st):
            step = d["steps"][si] if si < len(d["steps"]) else None
            if step is None:
                continue
            # (tilts, phi_base[, scales[, initSigma[, FGINNThreshold]]]): the keys of one [<Detector><i>] section
            tilts, phi_base = step[0], step[1]
            scales = step[2] if len(step) > 2 else (1.0,)
            sigma = step[3] if len(step) > 3 else init_sigma
            step_ratio = step[4] if len(step) > 4 else d.get("ratio", 0.8)
            views = view_schedule(tilts, phi_base, S["history"], scales)
            want_half = d.get("ratio_half", 0.0) > 0
            half_ori = d.get("half_orientation", False) or want_half

            def one_view(job, d=d, want_half=want_half, half_ori=half_ori, sigma=sigma):
                img, (zoom, tilt, phi) = job
                px, g = orc.synth_view(img, tilt, phi, zoom, sigma, 1)
                if g.w_new < 16 or g.h_new < 16:
                    return None
                r = orc.detect_describe_view(px, np.array(g.H), w, h, params=d.get("params"), half_orientation=half_ori, half_desc=want_half)
                return (r[0], r[3]) if want_half else (r[0], None)
            for im, img in enumerate((img1, img2)):
                regs = pmap(one_view, [(img, v) for v in views])      # views are independent; banks keep the view order
                n_views += len(views)
                S["banks"][im] += [r[0] for r in regs if r is not None]
                S["banks_h"][im] += [r[1] for r in regs if r is not None and want_half]
            if not views:
                continue
            if d.get("dist", 0.0) > 0:      # MatchFLANNDistance replaces what MatchFlannFGINN found (it clears the list, matching.cpp:585)
                ra, rb = np.concatenate(S["banks"][0]), np.concatenate(S["banks"][1])
                S["tc"] = (orc.match_distance(ra, rb, d["dist"]), ra, rb)
            elif step_ratio > 0:
                ra, rb = np.concatenate(S["banks"][0]), np.concatenate(S["banks"][1])
                S["tc"] = (match_fginn_par(ra, rb, step_ratio), ra, rb)
            if want_half:
                ha, hb = np.concatenate(S["banks_h"][0]), np.concatenate(S["banks_h"][1])
                S["tch"] = (match_fginn_par(ha, hb, d["ratio_half"]), ha, hb)
        if groups is not None and si < len(groups) and groups[si]:
            gspec = groups[si]
            for key, bk, r in (("tc", "banks", gspec.get("ratio", 0.0)), ("tch", "banks_h", gspec.get("ratio_half", -1.0))):
                if r < 0:
                    continue
                G[key] = None
                qa = [x for d in gspec["dets"] for x in st[d][bk][0]]
                ta = [x for d in gspec["dets"] for x in st[d][bk][1]]
                if r > 0 and qa and ta:
                    ra, rb = np.concatenate(qa), np.concatenate(ta)
                    G[key] = (match_fginn_par(ra, rb, r), ra, rb)
        # the duplicate filter and what follows only need coordinates, frames and the ratio / distance keys: one joint list
        # whose q / t index a joint region array
        tcs, ras, rbs = [], [], []
        nq = nt = 0
        for key in ("tch", "tc"):
            slots = list(st)
            if groups is not None:
                slots.insert(group_pos, G)
            for S in slots:
                if S[key] is None:
                    continue
                tc, ra, rb = S[key]
                tc = tc.copy(); tc["q"] += nq; tc["t"] += nt
                nq += len(ra); nt += len(rb)
                tcs.append(tc); ras.append(ra); rbs.append(rb)
        tc, ra, rb = np.concatenate(tcs), np.concatenate(ras), np.concatenate(rbs)
        un = orc.duplicate_filter(tc, ra, rb, 2.0, 1)
        u6, laf = u6_of(ra, rb, un), laf_of(ra, rb, un)
        mask, H, ninl, stats = loransac_h(u6, laf, seed_time=seed_time)
        n_desc = [sum(len(x) for S in st for x in S["banks"][0]), sum(len(x) for S in st for x in S["banks"][1])]
        first = st[0]
        out = dict(steps_done=si + 1, n_views=n_views, n_described=n_desc, n_tentatives=len(tc), n_unique=len(un),
                   n_inliers=ninl, stats=stats, H=H, mask=mask, u6=u6,
                   regions=(np.concatenate(first["banks"][0]) if first["banks"][0] else None,
                            np.concatenate(first["banks"][1]) if first["banks"][1] else None))
        if ninl >= min_matches:
            break
    return out


_PAIR_CACHE = {}


def cached_pair(w, h, seed, seed_time, **kw):
    """match_pair of synth.pair(w, h, seed), kept for the session: several tests check different entry points against the
    same oracle result (a 1080p pair takes seconds on the CPU).  Returns (img1, img2, H_true, oracle result)."""
    import synth
    key = (w, h, seed, seed_time, tuple(sorted(kw.items())))
    if key not in _PAIR_CACHE:
        ik = (w, h, seed)
        if ik not in _PAIR_CACHE:
            a, b, Ht = synth.pair(w, h, seed=seed)
            _PAIR_CACHE[ik] = (a, b, Ht, tuple(pmap(orc.detect_describe, (a, b))))
        a, b, Ht, regs = _PAIR_CACHE[ik]
        _PAIR_CACHE[key] = (a, b, Ht, match_pair(a, b, seed_time=seed_time, regions=regs, **kw))
    return _PAIR_CACHE[key]
