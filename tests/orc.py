"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None


class HessAffParams(C.Structure):
    _fields_ = [("numberOfScales", C.c_int), ("initialSigma", C.c_float), ("threshold", C.c_float),
                ("edgeEigenValueRatio", C.c_float), ("border", C.c_int), ("maxIterations", C.c_int),
                ("convergenceThreshold", C.c_float), ("smmWindowSize", C.c_int), ("doBaumberg", C.c_int),
                ("mode", C.c_int), ("relativeThreshold", C.c_float), ("regionsNumber", C.c_int),
                ("relativeRegionsNumber", C.c_float), ("detectorType", C.c_int), ("iiDoGMode", C.c_int),
                ("sampleFromImage", C.c_int),
                ("mserMaxArea", C.c_double), ("mserMinMargin", C.c_double), ("mserMinSize", C.c_int), ("affBmbrgMethod", C.c_int)]

    @staticmethod
    def mser(mode=0, min_margin=8, max_area=0.05, min_size=30, reg_number=500, rel_threshold=-1.0, rel_reg_number=-1.0):
        # build/config_affori_classic.ini [MSER] (DetectorType = DET_MSER = 3, detectors/structures.hpp:19)
        p = HessAffParams(3, 1.6, 5.33, 10.0, 5, 16, 0.05, 19, 1, mode, rel_threshold, reg_number, rel_reg_number, 3, 0, 0)
        p.mserMaxArea, p.mserMinMargin, p.mserMinSize = max_area, min_margin, min_size
        return p

    @staticmethod
    def default():
        # build/config_affori_classic.ini [HessianAffine]
        return HessAffParams(3, 1.6, 5.33, 10.0, 5, 16, 0.05, 19, 1, 0, -1.0, -1, -1.0, 0, 0, 0)

    @staticmethod
    def dog():
        # build/config_affori_classic.ini [DoG] (DetectorType = DET_DOG, io_mods.cpp:260-262)
        return HessAffParams(3, 1.6, 8.0, 10.0, 5, 32, 0.05, 19, 0, 0, 0.01, 3000, 0.5, 1, 0, 0)

    @staticmethod
    def harris():
        # build/config_affori_classic.ini [HarrisAffine] (DetectorType = DET_HARRIS, io_mods.cpp:208-210)
        return HessAffParams(3, 1.6, 15.0, 10.0, 5, 16, 0.1, 19, 0, 0, 0.1, 1000, 0.5, 2, 0, 0)


class Candidate(C.Structure):
    _fields_ = [("octave", C.c_int), ("level", C.c_int), ("r0", C.c_int), ("c0", C.c_int), ("r", C.c_int),
                ("c", C.c_int), ("x", C.c_float), ("y", C.c_float), ("s", C.c_float),
                ("pixelDistance", C.c_float), ("response", C.c_float), ("type", C.c_int)]


class AffKey(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("s", C.c_double), ("a11", C.c_double),
                ("a12", C.c_double), ("a21", C.c_double), ("a22", C.c_double), ("response", C.c_double),
                ("sub_type", C.c_int), ("octave", C.c_int), ("level", C.c_int), ("r0", C.c_int),
                ("c0", C.c_int), ("pad", C.c_int)]


CAND_DTYPE = np.dtype([("octave", "i4"), ("level", "i4"), ("r0", "i4"), ("c0", "i4"), ("r", "i4"), ("c", "i4"),
                       ("x", "f4"), ("y", "f4"), ("s", "f4"), ("pixelDistance", "f4"), ("response", "f4"),
                       ("type", "i4")])
AFFKEY_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("s", "f8"), ("a11", "f8"), ("a12", "f8"), ("a21", "f8"),
                         ("a22", "f8"), ("response", "f8"), ("sub_type", "i4"), ("octave", "i4"),
                         ("level", "i4"), ("r0", "i4"), ("c0", "i4"), ("pad", "i4")])


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.orc_pyramid_build.restype = C.c_void_p
        _lib.orc_det_pow2f.restype = C.c_float
        _lib.orc_det_expf.restype = C.c_float
        _lib.orc_atan2_lut.restype = C.c_float
        _lib.orc_det_pow2f.argtypes = [C.c_float]
        _lib.orc_det_expf.argtypes = [C.c_float]
        _lib.orc_atan2_lut.argtypes = [C.c_float, C.c_float]
        _lib.orc_det_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def grey_of_rgb(rgb):
    """(B + G + R) / 3.0 of GenerateSynthImageCorr on an 8-bit RGB array [h, w, 3] the way OpenCV evaluates the expression
    (fma(B + G, a, R * a), a = (float)(1/3.); see oracle/capi.cpp)."""
    a = np.ascontiguousarray(rgb[..., :3], dtype=np.uint8)
    out = np.empty(a.shape[:2], np.float32)
    lib().orc_grey_of_rgb(a.ctypes.data_as(C.c_void_p), C.c_long(a.shape[0] * a.shape[1]), out.ctypes.data_as(C.c_void_p))
    return out


def gauss_ksize(sigma):
    return lib().orc_gauss_ksize(C.c_float(sigma))


def gauss_kernel(n, sigma):
    out = np.zeros(n, np.float32)
    lib().orc_gauss_kernel(n, C.c_double(sigma), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def gauss_blur(img, sigma):
    a, p = _f(img)
    out = np.empty_like(a)
    lib().orc_gauss_blur(p, a.shape[1], a.shape[0], C.c_float(sigma), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def hessian_response(img, norm):
    a, p = _f(img)
    out = np.empty_like(a)
    lib().orc_hessian_response(p, a.shape[1], a.shape[0], C.c_float(norm), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def response(img, norm, kind):
    """kind 0 Hessian, 1 DoG, 2 Harris, 3 iiDoG (ScaleSpaceDetector::Response, pyramid.cpp:126-163)."""
    a, p = _f(img)
    out = np.empty_like(a)
    lib().orc_response(p, a.shape[1], a.shape[0], C.c_float(norm), kind, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def resize_half(img):
    a, p = _f(img)
    dw, dh = C.c_int(), C.c_int()
    lib().orc_resize_half_dims(a.shape[1], a.shape[0], C.byref(dw), C.byref(dh))
    out = np.empty((dh.value, dw.value), np.float32)
    lib().orc_resize_half(p, a.shape[1], a.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def interpolate(img, ofsx, ofsy, a11, a12, a21, a22, rw, rh):
    a, p = _f(img)
    out = np.empty((rh, rw), np.float32)
    t = lib().orc_interpolate(p, a.shape[1], a.shape[0], C.c_float(ofsx), C.c_float(ofsy), C.c_float(a11),
                              C.c_float(a12), C.c_float(a21), C.c_float(a22), rw, rh,
                              out.ctypes.data_as(C.POINTER(C.c_float)))
    return out, bool(t)


def gauss_mask(size):
    out = np.empty((size, size), np.float32)
    lib().orc_gauss_mask(size, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def circular_gauss_mask(size, sigma):
    out = np.empty((size, size), np.float32)
    lib().orc_circular_gauss_mask(size, C.c_float(sigma), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


class Pyramid:
    def __init__(self, img, params=None):
        self.params = params or HessAffParams.default()
        a, p = _f(img)
        self.h = C.c_void_p(lib().orc_pyramid_build(p, a.shape[1], a.shape[0], C.byref(self.params)))
        self.n_oct = lib().orc_pyramid_octaves(self.h)

    def dims(self, o):
        w, h = C.c_int(), C.c_int()
        lib().orc_pyramid_dims(self.h, o, C.byref(w), C.byref(h))
        return w.value, h.value

    def plane(self, o, level, kind):
        w, h = self.dims(o)
        out = np.empty((h, w), np.float32)
        lib().orc_pyramid_plane(self.h, o, level, kind, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def candidates(self, max_out=1 << 20):
        out = np.zeros(max_out, CAND_DTYPE)
        raw = np.zeros((max_out, 4), np.int32)
        nraw = C.c_int()
        n = lib().orc_pyramid_candidates(self.h, C.byref(self.params), out.ctypes.data_as(C.POINTER(Candidate)),
                                         max_out, raw.ctypes.data_as(C.POINTER(C.c_int)), max_out, C.byref(nraw))
        return out[:n].copy(), raw[:nraw.value].copy()

    def affine_shape(self, o, level, x, y, s, pd):
        a4 = (C.c_float * 4)()
        it = C.c_int()
        ok = lib().orc_affine_shape(self.h, o, level, C.c_float(x), C.c_float(y), C.c_float(s), C.c_float(pd),
                                    C.byref(self.params), a4, C.byref(it))
        return bool(ok), np.array(list(a4), np.float32), it.value

    def __del__(self):
        try:
            lib().orc_pyramid_free(self.h)
        except Exception:
            pass


def detect_hessian_affine(img, params=None, max_out=1 << 20):
    params = params or HessAffParams.default()
    a, p = _f(img)
    out = np.zeros(max_out, AFFKEY_DTYPE)
    n = lib().orc_detect_hessian_affine(p, a.shape[1], a.shape[0], C.byref(params),
                                        out.ctypes.data_as(C.POINTER(AffKey)), max_out)
    return out[:n].copy()


def svd2x2(a):
    """(d[2], U[2, 2], Vt[2, 2], degenerate) of a 2x2 fp32 matrix, the restated cv::SVD::compute."""
    a = np.ascontiguousarray(a, np.float32).reshape(4)
    out = np.zeros(10, np.float32)
    deg = lib().orc_svd2x2(a.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:2].copy(), out[2:6].reshape(2, 2).copy(), out[6:10].reshape(2, 2).copy(), bool(deg)


def mser_regions(img, params=None, max_regions=1 << 16, max_runs=1 << 24):
    """MSER+ then MSER- regions of one image: (info[n, 10], runs list per region [k, 3], ell[n, 5])."""
    params = params or HessAffParams.mser()
    a, p = _f(img)
    info = np.zeros((max_regions, 10), np.int32)
    runs = np.zeros((max_runs, 3), np.int32)
    ell = np.zeros((max_regions, 5), np.float64)
    tot = C.c_int(0)
    n = lib().orc_mser_regions(p, a.shape[1], a.shape[0], C.byref(params), info.ctypes.data_as(C.POINTER(C.c_int)), max_regions,
                               runs.ctypes.data_as(C.POINTER(C.c_int)), max_runs, ell.ctypes.data_as(C.POINTER(C.c_double)), C.byref(tot))
    assert n <= max_regions and tot.value <= max_runs
    info = info[:n].copy()
    ends = np.cumsum(info[:, 9])
    per = [runs[e - k:e].copy() for e, k in zip(ends, info[:, 9])]
    return info, per, ell[:n].copy()


def detect_mser_view(img, params, tilt, zoom, max_out=1 << 18):
    a, p = _f(img)
    out = np.zeros(max_out, AFFKEY_DTYPE)
    n = lib().orc_detect_mser_view(p, a.shape[1], a.shape[0], C.byref(params), C.c_double(tilt), C.c_double(zoom),
                                   out.ctypes.data_as(C.POINTER(AffKey)), max_out)
    return out[:n].copy()


# ---- orientation + description ---------------------------------------------------------------
REGION_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("s", "f8"), ("a11", "f8"), ("a12", "f8"), ("a21", "f8"),
                         ("a22", "f8"), ("response", "f8"), ("sub_type", "i4"), ("id", "i4"), ("parent", "i4"),
                         ("pad", "i4"), ("desc", "u1", (128,))])

# build/config_affori_classic.ini: [DominantOrientation] / [SIFTDescriptor]
ORI_MRSIZE, ORI_PATCH, ORI_MAXANG, ORI_TH = 5.1962, 32, 1, float(np.float32(0.8))
DESC_MRSIZE, DESC_PATCH = 5.1962, 41


def regions_from_keys(keys):
    r = np.zeros(len(keys), REGION_DTYPE)
    for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response", "sub_type"):
        r[f] = keys[f]
    r["id"] = np.arange(len(keys))
    r["parent"] = r["id"]
    return r


def dominant_angle(patch, th=ORI_TH):
    a, p = _f(patch)
    ang = C.c_float()
    ok = lib().orc_dominant_angle(p, a.shape[0], C.c_double(th), C.byref(ang))
    return bool(ok), ang.value


def sift_desc(patch, rootsift=True, max_bin=0.2):
    a, p = _f(patch)
    out = np.zeros(128, np.uint8)
    lib().orc_sift_desc(p, a.shape[0], int(rootsift), C.c_double(max_bin), out.ctypes.data_as(C.c_void_p))
    return out


def extract_desc_patch(img, region, mrsize=DESC_MRSIZE, ps=DESC_PATCH, photonorm=True):
    a, p = _f(img)
    r = np.ascontiguousarray(np.atleast_1d(region)[:1])
    out = np.empty((ps, ps), np.float32)
    lib().orc_extract_desc_patch(p, a.shape[1], a.shape[0], r.ctypes.data_as(C.c_void_p), C.c_double(mrsize), ps,
                                 int(photonorm), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def detect_orientation(img, regions, mrsize=ORI_MRSIZE, ps=ORI_PATCH, max_angles=ORI_MAXANG, th=ORI_TH):
    a, p = _f(img)
    r = np.ascontiguousarray(regions)
    out = np.zeros(len(r) + 1, REGION_DTYPE)
    n = lib().orc_detect_orientation(p, a.shape[1], a.shape[0], r.ctypes.data_as(C.c_void_p), len(r),
                                     C.c_double(mrsize), ps, max_angles, C.c_double(th),
                                     out.ctypes.data_as(C.c_void_p), len(out))
    return out[:n].copy()


def filter_centres_inside(regions, w, h):
    r = np.ascontiguousarray(regions).copy()
    n = lib().orc_filter_centres_inside(r.ctypes.data_as(C.c_void_p), len(r), w, h)
    return r[:n].copy()


def affnet_apply(regions, a3, w, h, mrsize):
    r = np.ascontiguousarray(regions).copy()
    a = np.ascontiguousarray(a3, np.float32)
    n = lib().orc_affnet_apply(r.ctypes.data_as(C.c_void_p), len(r), a.ctypes.data_as(C.c_void_p), w, h, C.c_double(mrsize))
    return r[:n].copy()


def orinet_apply(regions, yx):
    r = np.ascontiguousarray(regions).copy()
    a = np.ascontiguousarray(yx, np.float32)
    lib().orc_orinet_apply(r.ctypes.data_as(C.c_void_p), len(r), a.ctypes.data_as(C.c_void_p))
    return r


def filter_touch_boundary(regions, w, h):
    r = np.ascontiguousarray(regions).copy()
    n = lib().orc_filter_touch_boundary(r.ctypes.data_as(C.c_void_p), len(r), w, h)
    return r[:n].copy()


def describe_rootsift(img, regions, mrsize=DESC_MRSIZE, ps=DESC_PATCH, photonorm=True):
    a, p = _f(img)
    r = np.ascontiguousarray(regions).copy()
    lib().orc_describe_rootsift(p, a.shape[1], a.shape[0], r.ctypes.data_as(C.c_void_p), len(r), C.c_double(mrsize), ps,
                                int(photonorm))
    return r


def detect_describe(img, params=None, max_out=1 << 18, half_orientation=False, half_desc=False, add_upright=False, max_angles=None,
                    fast_extraction=False):
    """HessianAffine + RootSIFT for one identity view; returns (regions, n_detected).  half_orientation: DetectOrientation in
    doHalfSIFT mode (what the reference does for a step whose descriptor list names a Half* descriptor); half_desc: also the
    HalfRootSIFT descriptors of the same regions -> (regions, half regions, n_detected)."""
    params = params or HessAffParams.default()
    a, p = _f(img)
    out = np.zeros(max_out, REGION_DTYPE)
    outh = np.zeros(max_out if half_desc else 1, REGION_DTYPE)
    ndet = C.c_int()
    flags = (1 if half_orientation else 0) | (2 if half_desc else 0) | (4 if add_upright else 0) | (8 if fast_extraction else 0)
    n = lib().orc_detect_describe_ex(p, a.shape[1], a.shape[0], C.byref(params), C.c_double(ORI_MRSIZE), ORI_PATCH,
                                     ORI_MAXANG if max_angles is None else max_angles, C.c_double(ORI_TH), C.c_double(DESC_MRSIZE), DESC_PATCH, 1, flags,
                                     out.ctypes.data_as(C.c_void_p), outh.ctypes.data_as(C.c_void_p) if half_desc else None, max_out,
                                     C.byref(ndet))
    if half_desc:
        return out[:n].copy(), outh[:n].copy(), ndet.value
    return out[:n].copy(), ndet.value


def extract_patches_column(img, regions, mr_size=ORI_MRSIZE, patch_size=32):
    """ExtractPatchesColumn (what DescribeWithZmq sends to a daemon), fp32 [n][ps][ps]."""
    a, p = _f(img)
    r = np.ascontiguousarray(regions)
    out = np.zeros((len(r), patch_size, patch_size), np.float32)
    lib().orc_extract_patches_column(p, a.shape[1], a.shape[0], r.ctypes.data_as(C.c_void_p), len(r), C.c_double(mr_size), patch_size,
                                     out.ctypes.data_as(C.c_void_p))
    return out


# ---- view synthesis ------------------------------------------------------------------------------
class ViewGeom(C.Structure):
    _fields_ = [("identity", C.c_int), ("w_rot", C.c_int), ("h_rot", C.c_int), ("w_new", C.c_int), ("h_new", C.c_int),
                ("ksize_x", C.c_int), ("ksize_y", C.c_int), ("pad", C.c_int),
                ("rotation", C.c_double), ("tilt", C.c_double), ("zoom", C.c_double), ("sigma_x", C.c_double),
                ("sigma_y", C.c_double), ("H", C.c_double * 9), ("warpRot", C.c_double * 6), ("warpTilt", C.c_double * 6)]


def view_geometry(w, h, tilt, phi, zoom=1.0, init_sigma=0.2):
    g = ViewGeom()
    lib().orc_view_geometry(w, h, C.c_double(tilt), C.c_double(phi), C.c_double(zoom), C.c_double(init_sigma), C.byref(g))
    return g


def warp_affine(img, M, dw, dh, cval=128.0):
    a, p = _f(img)
    M = np.ascontiguousarray(M, np.float64)
    out = np.zeros((dh, dw), np.float32)
    lib().orc_warp_affine(p, a.shape[1], a.shape[0], M.ctypes.data_as(C.c_void_p), dw, dh, C.c_float(cval),
                          out.ctypes.data_as(C.c_void_p))
    return out


def gauss_blur_xy(img, kx, ky, sx, sy):
    a, p = _f(img)
    out = np.zeros_like(a)
    lib().orc_gauss_blur_xy(p, a.shape[1], a.shape[0], kx, ky, C.c_double(sx), C.c_double(sy), out.ctypes.data_as(C.c_void_p))
    return out


def synth_view(img, tilt, phi, zoom=1.0, init_sigma=0.2, do_blur=1):
    """GenerateSynthImageCorr on a grey float image; returns (view, ViewGeom)."""
    a, p = _f(img)
    g = view_geometry(a.shape[1], a.shape[0], tilt, phi, zoom, init_sigma)
    out = np.zeros((g.h_new, g.w_new), np.float32)
    lib().orc_synth_view(p, a.shape[1], a.shape[0], C.c_double(tilt), C.c_double(phi), C.c_double(zoom),
                         C.c_double(init_sigma), do_blur, out.ctypes.data_as(C.c_void_p), C.byref(g))
    return out, g


def detect_describe_view(view, H, orig_w, orig_h, params=None, max_out=1 << 18, half_orientation=False, half_desc=False):
    """One synthesised view through detect/orient/reproject/describe; returns (regions in the original frame,
    the same regions in the view frame, n_detected) [+ the HalfRootSIFT regions (original frame) with half_desc]."""
    params = params or HessAffParams.default()
    a, p = _f(view)
    Hc = np.ascontiguousarray(H, np.float64).ravel()
    out = np.zeros(max_out, REGION_DTYPE)
    det = np.zeros(max_out, REGION_DTYPE)
    outh = np.zeros(max_out if half_desc else 1, REGION_DTYPE)
    ndet = C.c_int()
    flags = (1 if half_orientation else 0) | (2 if half_desc else 0)
    n = lib().orc_detect_describe_view_ex(p, a.shape[1], a.shape[0], Hc.ctypes.data_as(C.c_void_p), orig_w, orig_h,
                                          C.byref(params), C.c_double(ORI_MRSIZE), ORI_PATCH, ORI_MAXANG, C.c_double(ORI_TH),
                                          C.c_double(DESC_MRSIZE), DESC_PATCH, 1, flags, out.ctypes.data_as(C.c_void_p),
                                          det.ctypes.data_as(C.c_void_p), outh.ctypes.data_as(C.c_void_p) if half_desc else None,
                                          max_out, C.byref(ndet))
    if half_desc:
        return out[:n].copy(), det[:n].copy(), ndet.value, outh[:n].copy()
    return out[:n].copy(), det[:n].copy(), ndet.value


# ---- matching ------------------------------------------------------------------------------------
TENT_DTYPE = np.dtype([("q", "i4"), ("t", "i4"), ("t_bad", "i4"), ("t_2nd", "i4"), ("d1", "f4"), ("d2", "f4"),
                       ("d2nd", "f4"), ("pad", "f4"), ("ratio", "f8")])


def match_distance(q, t, threshold):
    """MatchFLANNDistance (matching.cpp:572-633), Hamming distance, exact search."""
    q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
    out = np.zeros(len(q) + 1, TENT_DTYPE)
    n = lib().orc_match_distance(q.ctypes.data_as(C.c_void_p), len(q), t.ctypes.data_as(C.c_void_p), len(t),
                                 C.c_double(threshold), out.ctypes.data_as(C.c_void_p), len(out))
    return out[:n].copy()


def match_fginn(q, t, ratio=0.8, contrad=10.0, nn=50):
    q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
    out = np.zeros(len(q) + 1, TENT_DTYPE)
    n = lib().orc_match_fginn(q.ctypes.data_as(C.c_void_p), len(q), t.ctypes.data_as(C.c_void_p), len(t),
                              C.c_double(ratio), C.c_double(contrad), nn, out.ctypes.data_as(C.c_void_p), len(out))
    return out[:n].copy()


def duplicate_filter(tc, q, t, r=2.0, mode=1):
    q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
    tc = np.ascontiguousarray(tc).copy()
    n = lib().orc_duplicate_filter(tc.ctypes.data_as(C.c_void_p), len(tc), q.ctypes.data_as(C.c_void_p), len(q),
                                   t.ctypes.data_as(C.c_void_p), len(t), C.c_double(r), mode)
    return tc[:n].copy()
