"""ctypes binding of libmodsgpu.so (include/mods_hip.h): the MI355X implementation of the MODS
detect -> describe -> match -> verify hot path.

The directory name carries a hyphen (repository contract), so load it with
`import importlib.util` (see `__graft_entry__.load_package()`), not with `import`.

There is no CPU path: every call fails loudly when the shared library or a HIP device is
missing."""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MODS_LIB") or os.path.join(PKG_DIR, "libmodsgpu.so")

MODS_OK = 0
STAGES = ["blur", "response", "resize", "nms", "localize", "baumberg", "sort", "orient", "describe", "match",
          "ransac_score", "synth", "blur_small", "pyramid", "match_nn1", "extract", "sift"]


class ModsError(RuntimeError):
    pass


class HessAffParams(C.Structure):
    """[HessianAffine] section of the reference .ini (io_mods.cpp:167-207)."""
    _fields_ = [("numberOfScales", C.c_int), ("initialSigma", C.c_float), ("threshold", C.c_float),
                ("edgeEigenValueRatio", C.c_float), ("border", C.c_int), ("maxIterations", C.c_int),
                ("convergenceThreshold", C.c_float), ("smmWindowSize", C.c_int), ("doBaumberg", C.c_int),
                ("mode", C.c_int), ("relativeThreshold", C.c_float), ("regionsNumber", C.c_int),
                ("relativeRegionsNumber", C.c_float), ("detectorType", C.c_int), ("iiDoGMode", C.c_int),
                ("sampleFromImage", C.c_int),
                ("mserMaxArea", C.c_double), ("mserMinMargin", C.c_double), ("mserMinSize", C.c_int), ("affBmbrgMethod", C.c_int)]

    @staticmethod
    def mser(mode=0, min_margin=8, max_area=0.05, min_size=30, reg_number=500, rel_threshold=-1.0, rel_reg_number=-1.0):
        """[MSER] of build/config_affori_classic.ini (DetectorType = DET_MSER, detectors/structures.hpp:19; io_mods.cpp:101-123)."""
        p = HessAffParams(3, 1.6, 5.33, 10.0, 5, 16, 0.05, 19, 1, mode, rel_threshold, reg_number, rel_reg_number, 3, 0, 0)
        p.mserMaxArea, p.mserMinMargin, p.mserMinSize = max_area, min_margin, min_size
        return p

    @staticmethod
    def default():
        # build/config_affori_classic.ini; mode FixedTh, the other selection keys at PyramidParams' defaults (-1)
        return HessAffParams(3, 1.6, 5.33, 10.0, 5, 16, 0.05, 19, 1, 0, -1.0, -1, -1.0, 0, 0, 0)

    @staticmethod
    def dog():
        """[DoG] of build/config_affori_classic.ini: the same detector searching the difference-of-Gaussians response
        (DetectorType = DET_DOG, io_mods.cpp:260-262)."""
        return HessAffParams(3, 1.6, 8.0, 10.0, 5, 32, 0.05, 19, 0, 0, 0.01, 3000, 0.5, 1, 0, 0)

    @staticmethod
    def harris():
        """[HarrisAffine] of build/config_affori_classic.ini (DetectorType = DET_HARRIS, io_mods.cpp:208-210)."""
        return HessAffParams(3, 1.6, 15.0, 10.0, 5, 16, 0.1, 19, 0, 0, 0.1, 1000, 0.5, 2, 0, 0)


AFFKEY_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("s", "f8"), ("a11", "f8"), ("a12", "f8"), ("a21", "f8"),
                         ("a22", "f8"), ("response", "f8"), ("sub_type", "i4"), ("octave", "i4"),
                         ("level", "i4"), ("r0", "i4"), ("c0", "i4"), ("pad", "i4")])
CAND_DTYPE = np.dtype([("octave", "i4"), ("level", "i4"), ("r0", "i4"), ("c0", "i4"), ("r", "i4"), ("c", "i4"),
                       ("x", "f4"), ("y", "f4"), ("s", "f4"), ("pixelDistance", "f4"), ("response", "f4"),
                       ("type", "i4")])

REGION_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("s", "f8"), ("a11", "f8"), ("a12", "f8"), ("a21", "f8"),
                         ("a22", "f8"), ("response", "f8"), ("sub_type", "i4"), ("id", "i4"), ("parent", "i4"),
                         ("pad", "i4"), ("desc", "u1", (128,))])


class DescribeParams(C.Structure):
    """[DominantOrientation] + [SIFTDescriptor] (io_mods.cpp:731-740, 423-436)."""
    _fields_ = [("ori_mrSize", C.c_double), ("ori_patchSize", C.c_int), ("ori_maxAngles", C.c_int),
                ("ori_threshold", C.c_double), ("desc_mrSize", C.c_double), ("desc_patchSize", C.c_int),
                ("photoNorm", C.c_int), ("rootSift", C.c_int), ("maxBinValue", C.c_double),
                ("ori_halfMode", C.c_int), ("addUpRight", C.c_int), ("halfDesc", C.c_int), ("fastExtraction", C.c_int)]

    @staticmethod
    def default():
        # config_affori_classic.ini; threshold is parsed into a float member (descriptors_parameters.hpp:10)
        return DescribeParams(5.1962, 32, 1, float(np.float32(0.8)), 5.1962, 41, 1, 1, 0.2, 0, 0, 0, 0)


_lib = None


def build():
    """Compile every HIP source for gfx950 into libmodsgpu.so (in-tree)."""
    subprocess.check_call(["make", "-s", "-j4", "-C", PKG_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ModsError("libmodsgpu.so is not built (run __graft_entry__.build()); there is no CPU path")
        # PyTorch-ROCm bundles its own libamdhip64.so.7.  Two HIP runtimes in one process do not
        # share devices, so when torch is used for device memory / torch.distributed it must be
        # loaded first: libmodsgpu's NEEDED libamdhip64.so.7 then binds to the copy already mapped.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.mods_last_error.restype = C.c_char_p
        _lib.mods_ctx_stream.restype = C.c_void_p
    return _lib


def _check(rc):
    if rc != MODS_OK:
        raise ModsError("libmodsgpu error %d: %s" % (rc, lib().mods_last_error().decode()))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class ViewGeom(C.Structure):
    """mods_view_geom: what GenerateSynthImageCorr derives before touching pixels."""
    _fields_ = [("identity", C.c_int), ("w_rot", C.c_int), ("h_rot", C.c_int), ("w_new", C.c_int), ("h_new", C.c_int),
                ("ksize_x", C.c_int), ("ksize_y", C.c_int), ("pad", C.c_int),
                ("rotation", C.c_double), ("tilt", C.c_double), ("zoom", C.c_double), ("sigma_x", C.c_double),
                ("sigma_y", C.c_double), ("H", C.c_double * 9), ("warpRot", C.c_double * 6), ("warpTilt", C.c_double * 6)]


def view_geometry(w, h, tilt, phi, zoom=1.0, init_sigma=0.2):
    g = ViewGeom()
    _check(lib().mods_view_geometry(w, h, C.c_double(tilt), C.c_double(phi), C.c_double(zoom), C.c_double(init_sigma), C.byref(g)))
    return g


def view_ctx_dims(w, h):
    """Context size that holds every rotated intermediate of a w x h image."""
    d = int(np.ceil(np.hypot(w, h))) + 2
    return d, d


class Context:
    def __init__(self, device=0, max_w=1920, max_h=1080, batch=1, nonblocking=False):
        self.h = C.c_void_p()
        if nonblocking:     # a stream without implicit ordering after the default stream (what the pipeline's workers use)
            _check(lib().mods_ctx_create_ex(device, max_w, max_h, batch, 1, C.byref(self.h)))
        else:
            _check(lib().mods_ctx_create(device, max_w, max_h, batch, C.byref(self.h)))
        self.batch = batch
        self.device = device

    def close(self):
        if self.h:
            lib().mods_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(lib().mods_ctx_sync(self.h))

    @property
    def stream(self):
        return lib().mods_ctx_stream(self.h)

    # ---- timing
    def timing_enable(self, stages):
        mask = 0
        for s in stages:
            mask |= 1 << STAGES.index(s)
        _check(lib().mods_ctx_timing_enable(self.h, mask))

    def timing_read(self, stage):
        ms, n, by = C.c_double(), C.c_int(), C.c_double()
        _check(lib().mods_ctx_timing_read(self.h, STAGES.index(stage), C.byref(ms), C.byref(n), C.byref(by)))
        return ms.value, n.value, by.value

    def baumberg_stats_enable(self, on=True):
        """count the keypoints that enter the Baumberg iteration and the iterations they run (mods_baumberg_stats)"""
        _check(lib().mods_baumberg_stats_enable(self.h, 1 if on else 0))

    def baumberg_stats(self, img):
        kp, it = C.c_ulonglong(), C.c_ulonglong()
        _check(lib().mods_baumberg_stats(self.h, img, C.byref(kp), C.byref(it)))
        return kp.value, it.value

    def graphs(self, on=True):
        """replay the launches of detect + describe as a hipGraph from the second call with the same arguments on"""
        _check(lib().mods_ctx_graphs(self.h, 1 if on else 0))

    def graph_replays(self):
        lib().mods_ctx_graph_replays.restype = C.c_long
        return int(lib().mods_ctx_graph_replays(self.h))

    def pyramid_streams(self, n):
        _check(lib().mods_ctx_pyramid_streams(self.h, int(n)))

    def timing_reset(self):
        _check(lib().mods_ctx_timing_reset(self.h))

    # ---- primitives (host arrays)
    def gauss_blur(self, img, sigma):
        a = np.ascontiguousarray(img, np.float32)
        out = np.empty_like(a)
        _check(lib().mods_gauss_blur(self.h, _fp(a), a.shape[1], a.shape[0], C.c_float(sigma), _fp(out)))
        return out

    def hessian_response(self, img, norm):
        a = np.ascontiguousarray(img, np.float32)
        out = np.empty_like(a)
        _check(lib().mods_hessian_response(self.h, _fp(a), a.shape[1], a.shape[0], C.c_float(norm), _fp(out)))
        return out

    def resize_half(self, img):
        a = np.ascontiguousarray(img, np.float32)
        out = np.empty(((a.shape[0] + 1) // 2 + 1, (a.shape[1] + 1) // 2 + 1), np.float32).ravel()
        dw, dh = C.c_int(), C.c_int()
        _check(lib().mods_resize_half(self.h, _fp(a), a.shape[1], a.shape[0], _fp(out), C.byref(dw), C.byref(dh)))
        return out[:dw.value * dh.value].reshape(dh.value, dw.value).copy()

    # ---- detector
    def detect_hessian_affine(self, img, params=None, max_out=1 << 18):
        params = params or HessAffParams.default()
        a = np.ascontiguousarray(img, np.float32)
        out = np.zeros(max_out, AFFKEY_DTYPE)
        n = C.c_int()
        _check(lib().mods_detect_hessian_affine(self.h, _fp(a), a.shape[1], a.shape[0], a.shape[1], C.byref(params),
                                                out.ctypes.data_as(C.c_void_p), max_out, C.byref(n)))
        return out[:n.value].copy()

    def detect_hessian_affine_dev(self, dev_ptr, n_img, w, h, params=None, max_out=1 << 18, fetch=True):
        """dev_ptr: device address of [n_img][h][w] fp32 (e.g. torch tensor .data_ptr())."""
        params = params or HessAffParams.default()
        counts = (C.c_int * n_img)()
        out = np.zeros((n_img, max_out), AFFKEY_DTYPE) if fetch else None
        _check(lib().mods_detect_hessian_affine_dev(self.h, C.c_void_p(dev_ptr), n_img, w, h, w, C.byref(params),
                                                    out.ctypes.data_as(C.c_void_p) if fetch else None, max_out, counts))
        if fetch:
            return [out[i, :counts[i]].copy() for i in range(n_img)]
        return list(counts)

    # ---- introspection for parity tests
    def pyramid_octaves(self):
        return lib().mods_pyramid_octaves(self.h)

    def pyramid_dims(self, o):
        w, h = C.c_int(), C.c_int()
        _check(lib().mods_pyramid_dims(self.h, o, C.byref(w), C.byref(h)))
        return w.value, h.value

    def pyramid_plane(self, img, o, level, kind):
        w, h = self.pyramid_dims(o)
        out = np.empty((h, w), np.float32)
        _check(lib().mods_pyramid_plane(self.h, img, o, level, kind, _fp(out)))
        return out

    def pyramid_candidates(self, img=0, max_out=1 << 20):
        out = np.zeros(max_out, CAND_DTYPE)
        n = C.c_int()
        _check(lib().mods_pyramid_candidates(self.h, img, out.ctypes.data_as(C.c_void_p), max_out, C.byref(n)))
        return out[:n.value].copy()

    # ---- orientation + description
    def orient_describe(self, img, keys, params=None, max_out=None):
        params = params or DescribeParams.default()
        a = np.ascontiguousarray(img, np.float32)
        k = np.ascontiguousarray(keys)
        max_out = max_out or len(k) + 1
        out = np.zeros(max_out, REGION_DTYPE)
        n = C.c_int()
        _check(lib().mods_orient_describe(self.h, _fp(a), a.shape[1], a.shape[0], a.shape[1], k.ctypes.data_as(C.c_void_p),
                                          len(k), C.byref(params), out.ctypes.data_as(C.c_void_p), max_out, C.byref(n)))
        return out[:n.value].copy()

    def regions_fetch_half(self, img=0, max_out=1 << 18):
        """HalfRootSIFT twins of the regions of image slot img (desc[:64]; describe with DescribeParams.halfDesc = 1)."""
        out = np.zeros(max_out, REGION_DTYPE)
        n = C.c_int()
        _check(lib().mods_regions_fetch_half(self.h, img, out.ctypes.data_as(C.c_void_p), max_out, C.byref(n)))
        return out[:n.value].copy()

    def detect_describe_dev(self, dev_ptr, n_img, w, h, det=None, desc=None):
        det = det or HessAffParams.default()
        desc = desc or DescribeParams.default()
        nd, nr = (C.c_int * n_img)(), (C.c_int * n_img)()
        _check(lib().mods_detect_describe_dev(self.h, C.c_void_p(dev_ptr), n_img, w, h, w, C.byref(det), C.byref(desc), nd, nr))
        return list(nd), list(nr)

    def regions_fetch(self, img, max_out=1 << 18):
        n = C.c_int()
        _check(lib().mods_regions_fetch(self.h, img, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), REGION_DTYPE)
        _check(lib().mods_regions_fetch(self.h, img, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
        return out[:n.value].copy()

    def set_external_descriptor(self, fn_ptr, user, mr_size=3.0 * np.sqrt(3.0), patch_size=32):
        """fn_ptr: address of a mods_descriptor_fn (e.g. mods_zmq_descriptor_hook of libmodszmq.so), user: its void* argument
        (keep the object it points to alive).  fn_ptr = None restores RootSIFT."""
        _check(lib().mods_ctx_set_external_descriptor(self.h, C.c_void_p(fn_ptr), C.c_void_p(user), C.c_double(mr_size), patch_size))

    def set_external_shape(self, fn_ptr, user, mr_size=3.0 * np.sqrt(3.0), patch_size=32):
        """AffNet in the place of Baumberg (detect with doBaumberg = 0): fn returns (a11, a21, a22) per patch.  None: off."""
        _check(lib().mods_ctx_set_external_shape(self.h, C.c_void_p(fn_ptr), C.c_void_p(user), C.c_double(mr_size), patch_size))

    def set_external_orientation(self, fn_ptr, user, mr_size=3.0 * np.sqrt(3.0), patch_size=32):
        """OriNet in the place of the dominant gradient orientation: fn returns (y, x) per patch.  None: off."""
        _check(lib().mods_ctx_set_external_orientation(self.h, C.c_void_p(fn_ptr), C.c_void_p(user), C.c_double(mr_size), patch_size))

    def patches_fetch(self, img, ps, max_regions=1 << 17):
        n = C.c_int()
        out = np.zeros((max_regions, ps, ps), np.float32) if max_regions <= 4096 else None
        if out is None:
            cnt = C.c_int()
            _check(lib().mods_regions_fetch(self.h, img, None, 0, C.byref(cnt)))
            out = np.zeros((max(cnt.value, 1), ps, ps), np.float32)
        _check(lib().mods_patches_fetch(self.h, img, ps, _fp(out), out.shape[0], C.byref(n)))
        return out[:n.value].copy()

    def regions_copy_dev(self, img, dst_ptr, n):
        """Device-to-device copy of the first n regions of image slot img (e.g. into a torch tensor)."""
        _check(lib().mods_regions_copy_dev(self.h, img, C.c_void_p(dst_ptr), n))

    def dominant_angle(self, patch, th=float(np.float32(0.8))):
        a = np.ascontiguousarray(patch, np.float32)
        ang, found = C.c_float(), C.c_int()
        _check(lib().mods_dominant_angle(self.h, _fp(a), a.shape[0], C.c_double(th), C.byref(ang), C.byref(found)))
        return bool(found.value), ang.value

    def selftest_fast_sqrt(self):
        """(differs inside the stated domain, differs for 0 < x < 2^-96, everywhere-exact form differs on a non-negative operand,
        operands visited, negative operands where either form differs)"""
        out = (C.c_ulonglong * 5)()
        _check(lib().mods_selftest_fast_sqrt(self.h, out))
        return tuple(int(v) for v in out)

    def sift_patch(self, patch, rootsift=True, max_bin=0.2):
        a = np.ascontiguousarray(patch, np.float32)
        out = np.zeros(128, np.uint8)
        _check(lib().mods_sift_patch(self.h, _fp(a), a.shape[0], int(rootsift), C.c_double(max_bin),
                                     out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- view synthesis
    def warp_affine(self, img, M, dw, dh, cval=128.0):
        a = np.ascontiguousarray(img, np.float32)
        Mc = np.ascontiguousarray(M, np.float64).ravel()
        out = np.empty((dh, dw), np.float32)
        _check(lib().mods_warp_affine(self.h, _fp(a), a.shape[1], a.shape[0], Mc.ctypes.data_as(C.c_void_p), dw, dh,
                                      C.c_float(cval), _fp(out)))
        return out

    def gauss_blur_xy(self, img, kx, ky, sx, sy):
        a = np.ascontiguousarray(img, np.float32)
        out = np.empty_like(a)
        _check(lib().mods_gauss_blur_xy(self.h, _fp(a), a.shape[1], a.shape[0], kx, ky, C.c_double(sx), C.c_double(sy), _fp(out)))
        return out

    def synth_view_dev(self, src_ptr, w, h, geom, dst_ptr, do_blur=1):
        _check(lib().mods_synth_view_dev(self.h, C.c_void_p(src_ptr), w, h, w, C.byref(geom), do_blur, C.c_void_p(dst_ptr)))

    def detect_describe_view_dev(self, src_ptr, w, h, tilt, phi, zoom=1.0, init_sigma=0.2, do_blur=1, det=None, desc=None):
        """One synthesised view of the w x h image at src_ptr (HBM): returns (geom, n_detected, n_regions); the
        regions (original frame) stay in the context (regions_fetch(0))."""
        det = det or HessAffParams.default()
        desc = desc or DescribeParams.default()
        g = ViewGeom()
        nd, nr = C.c_int(), C.c_int()
        _check(lib().mods_detect_describe_view_dev(self.h, C.c_void_p(src_ptr), w, h, w, C.c_double(tilt), C.c_double(phi),
                                                   C.c_double(zoom), C.c_double(init_sigma), do_blur, C.byref(det),
                                                   C.byref(desc), C.byref(g), C.byref(nd), C.byref(nr)))
        return g, nd.value, nr.value

    def view_pixels(self, geom):
        """Host copy of the last synthesised view."""
        out = np.empty((geom.h_new, geom.w_new), np.float32)
        _check(lib().mods_view_fetch(self.h, C.byref(geom), _fp(out)))
        return out

    # ---- matching
    def match_fginn(self, q, t, ratio=0.8, contrad=10.0, nn=50):
        q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
        cap = max(len(q), 1)
        out = np.zeros(cap, TENT_DTYPE)
        u6 = np.zeros((cap, 6), np.float64)
        laf = np.zeros((cap, 14), np.float64)
        n = C.c_int()
        _check(lib().mods_match_fginn(self.h, q.ctypes.data_as(C.c_void_p), len(q), t.ctypes.data_as(C.c_void_p), len(t),
                                      C.c_double(ratio), C.c_double(contrad), nn, out.ctypes.data_as(C.c_void_p),
                                      u6.ctypes.data_as(C.c_void_p), laf.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        self.last_laf = laf[:n.value].copy()
        return out[:n.value].copy(), u6[:n.value].copy()

    def match_distance(self, q, t, threshold):
        """MatchFLANNDistance (matching.cpp:572-633): nearest train by Hamming distance, kept within the threshold."""
        q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
        cap = max(len(q), 1)
        out = np.zeros(cap, TENT_DTYPE)
        u6 = np.zeros((cap, 6), np.float64)
        laf = np.zeros((cap, 14), np.float64)
        n = C.c_int()
        _check(lib().mods_match_distance(self.h, q.ctypes.data_as(C.c_void_p), len(q), t.ctypes.data_as(C.c_void_p), len(t),
                                         C.c_double(threshold), out.ctypes.data_as(C.c_void_p),
                                         u6.ctypes.data_as(C.c_void_p), laf.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value].copy(), u6[:n.value].copy()

    def match_dev(self, img_q, img_t, ratio=0.8, contrad=10.0, nn=50, cap=1 << 18):
        out = np.zeros(cap, TENT_DTYPE)
        u6 = np.zeros((cap, 6), np.float64)
        laf = np.zeros((cap, 14), np.float64)
        n = C.c_int()
        _check(lib().mods_match_dev(self.h, img_q, img_t, C.c_double(ratio), C.c_double(contrad), nn,
                                    out.ctypes.data_as(C.c_void_p), u6.ctypes.data_as(C.c_void_p),
                                    laf.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        self.last_laf = laf[:n.value].copy()
        return out[:n.value].copy(), u6[:n.value].copy()


TENT_DTYPE = np.dtype([("q", "i4"), ("t", "i4"), ("t_bad", "i4"), ("t_2nd", "i4"), ("d1", "f4"), ("d2", "f4"),
                       ("d2nd", "f4"), ("pad", "f4"), ("ratio", "f8")])


def duplicate_filter(tent, u6, r=2.0, mode=1, laf=None):
    """Host-side DuplicateFiltering (matching.cpp:2615-2679) on (tentatives, correspondences[, frames])."""
    tent = np.ascontiguousarray(tent).copy()
    u6 = np.ascontiguousarray(u6, np.float64).copy()
    lf = np.ascontiguousarray(laf, np.float64).copy() if laf is not None else None
    n = C.c_int()
    _check(lib().mods_duplicate_filter(tent.ctypes.data_as(C.c_void_p), u6.ctypes.data_as(C.c_void_p),
                                       lf.ctypes.data_as(C.c_void_p) if lf is not None else None, len(tent),
                                       C.c_double(r), mode, C.byref(n)))
    if lf is not None:
        return tent[:n.value].copy(), u6[:n.value].copy(), lf[:n.value].copy()
    return tent[:n.value].copy(), u6[:n.value].copy()


def duplicate_filter_gpu(ctx, tent, u6, laf, r=2.0, mode=1):
    """DuplicateFiltering on the context's GPU (csrc/dedup.hip); returns (tent, u6, laf, on_device)."""
    tent = np.ascontiguousarray(tent).copy()
    u6 = np.ascontiguousarray(u6, np.float64).copy()
    lf = np.ascontiguousarray(laf, np.float64).copy()
    n, dev = C.c_int(), C.c_int()
    _check(lib().mods_duplicate_filter_gpu(ctx.h, tent.ctypes.data_as(C.c_void_p), u6.ctypes.data_as(C.c_void_p),
                                           lf.ctypes.data_as(C.c_void_p), len(tent), C.c_double(r), mode, C.byref(n), C.byref(dev)))
    return tent[:n.value].copy(), u6[:n.value].copy(), lf[:n.value].copy(), bool(dev.value)


# ---- verification (degensac C ABI + LORANSACFiltering) ---------------------------------------------
class Score(C.Structure):
    _fields_ = [("I", C.c_uint), ("J", C.c_double)]


class RansacParams(C.Structure):
    """[RANSAC] section (io_mods.cpp:437-455).  errorType 0 Sampson, 1 SymmMax, 2 SymmSum."""
    _fields_ = [("err_threshold", C.c_double), ("confidence", C.c_double), ("max_samples", C.c_int),
                ("localOptimization", C.c_int), ("LAFCoef", C.c_double), ("HLAFCoef", C.c_double),
                ("errorType", C.c_int), ("doSymmCheck", C.c_int), ("useF", C.c_int),
                ("groundTruth", C.c_int), ("ransacForStopping", C.c_int), ("gtH", C.c_double * 9)]

    @staticmethod
    def default(useF=0):
        return RansacParams(4.0, 0.99, 1000000, 1, 2.0, 12.0, 0, 1, useF, 0, 0)   # config_affori_classic.ini


_ERR = {"sampson": ("HDs", "HDsi", "HDsidx"), "symm_max": ("HDsSymMax", "HDsiSymMax", "HDsSymidxMax"),
        "symm_sum": ("HDsSym", "HDsiSym", "HDsSymidx")}


def ransac_pin_seed(seed):
    lib().mods_ransac_pin_seed.argtypes = [C.c_long]
    lib().mods_ransac_pin_seed(seed)


def ransac_h(u6, th_sq, conf=0.99, max_sam=1000000, err="sampson", sym_check=1, seed_time=12345):
    """exp_ransacHcustom exactly as LORANSACFiltering calls it (matching.cpp:731)."""
    L = lib()
    L.exp_ransacHcustom.restype = Score
    ransac_pin_seed(seed_time)
    u = np.ascontiguousarray(u6, np.float64).copy()
    n = len(u)
    H = np.zeros(9, np.float64)
    inl = np.zeros(max(n, 1), np.uint8)
    data_out = np.zeros(max(18 * n, 8), np.int32)
    resids = C.POINTER(C.c_double)()
    f = [C.cast(getattr(L, name), C.c_void_p) for name in _ERR[err]]
    S = L.exp_ransacHcustom(u.ctypes.data_as(C.c_void_p), n, C.c_double(th_sq), C.c_double(conf), max_sam,
                            H.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p), 4,
                            data_out.ctypes.data_as(C.c_void_p), 1, C.c_uint(0), C.byref(resids), f[0], f[1], f[2],
                            sym_check)
    C.CDLL(None).free(resids)
    return dict(I=S.I, J=S.J, H=H, inl=inl[:n], samples=int(data_out[0]), lo=int(data_out[1]), rej=int(data_out[2]))


_FERR = {"sampson": ("FDs", "exFDs"), "symm": ("FDsSym", "exFDsSym")}


def ransac_f(u6, th_sq, conf=0.99, max_sam=100000, err="sampson", sym_check=0, do_lo=1, inl_limit=0, seed_time=12345):
    """exp_ransacFcustom exactly as LORANSACFiltering calls it (matching.cpp:722)."""
    L = lib()
    L.exp_ransacFcustom.restype = C.c_int
    ransac_pin_seed(seed_time)
    u = np.ascontiguousarray(u6, np.float64).copy()
    n = len(u)
    F = np.zeros(9, np.float64)
    Hb = np.zeros(9, np.float64)
    inl = np.zeros(max(n, 1), np.uint8)
    data_out = np.zeros(max(18 * n, 16), np.int32)
    resids = C.POINTER(C.c_double)()
    Ih = C.c_int(0)
    fds, exfds = (C.cast(getattr(L, name), C.c_void_p) for name in _FERR[err])
    I = L.exp_ransacFcustom(u.ctypes.data_as(C.c_void_p), n, C.c_double(th_sq), C.c_double(conf), max_sam,
                            F.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p), data_out.ctypes.data_as(C.c_void_p),
                            do_lo, C.c_uint(inl_limit), C.byref(resids), Hb.ctypes.data_as(C.c_void_p), C.byref(Ih), exfds, fds,
                            sym_check)
    C.CDLL(None).free(resids)
    return dict(I=I, F=F, inl=inl[:n], samples=int(data_out[0]), lo=int(data_out[1]), Ih=Ih.value, hist=data_out[2:n + 3].copy())


def loransac_h(u6, laf, params=None, seed_time=12345):
    params = params or RansacParams.default()
    ransac_pin_seed(seed_time)
    u = np.ascontiguousarray(u6, np.float64)
    n = len(u)
    lf = np.ascontiguousarray(laf, np.float64) if laf is not None else None
    mask = np.zeros(max(n, 1), np.uint8)
    H = np.zeros(9, np.float64)
    ninl = C.c_int()
    stats = (C.c_int * 3)()
    _check(lib().mods_loransac_h(u.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p) if lf is not None else None,
                                 n, C.byref(params), mask.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p),
                                 C.byref(ninl), stats))
    return mask[:n].astype(bool), H.reshape(3, 3), ninl.value, list(stats)


def loransac_f(u6, laf, params=None, seed_time=12345):
    """useF branch of LORANSACFiltering: DEGENSAC + F_LAF_check.  Returns (mask, F as stored, n, stats)."""
    params = params or RansacParams.default(useF=1)
    ransac_pin_seed(seed_time)
    u = np.ascontiguousarray(u6, np.float64)
    n = len(u)
    lf = np.ascontiguousarray(laf, np.float64) if laf is not None else None
    mask = np.zeros(max(n, 1), np.uint8)
    F = np.zeros(9, np.float64)
    ninl = C.c_int()
    stats = (C.c_int * 3)()
    _check(lib().mods_loransac_f(u.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p) if lf is not None else None,
                                 n, C.byref(params), mask.ctypes.data_as(C.c_void_p), F.ctypes.data_as(C.c_void_p),
                                 C.byref(ninl), stats))
    return mask[:n].astype(bool), F, ninl.value, list(stats)


def verify_tentatives(tent, u6, laf, params, device=0, seed_time=None):
    """mods_verify_tentatives: duplicate filtering + LORANSACFiltering in the order [DuplicateFiltering] doBeforeRANSAC asks for
    (mods.cpp:278-368).  Returns (verified tentatives, their u6, their laf, n_unique, H, stats)."""
    if seed_time is not None:
        ransac_pin_seed(seed_time)
    tent = np.ascontiguousarray(tent).copy()
    u = np.ascontiguousarray(u6, np.float64).copy()
    lf = np.ascontiguousarray(laf, np.float64).copy()
    nu, nv = C.c_int(), C.c_int()
    H = np.zeros(9, np.float64)
    stats = (C.c_int * 3)()
    _check(lib().mods_verify_tentatives(device, C.byref(params), tent.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p),
                                        lf.ctypes.data_as(C.c_void_p), len(tent), C.byref(nu), C.byref(nv),
                                        H.ctypes.data_as(C.c_void_p), stats, None, None))
    m = nv.value
    return tent[:m].copy(), u[:m].copy(), lf[:m].copy(), nu.value, H, list(stats)


# ---- multi-view representation and the MODS step loop --------------------------------------------------------
class ViewPar(C.Structure):
    _fields_ = [("zoom", C.c_double), ("tilt", C.c_double), ("phi", C.c_double)]


class LadderStep(C.Structure):
    """One [HessianAffine<i>] section of an iterations .ini (io_mods.cpp:457-492)."""
    _fields_ = [("scale_set", C.c_double * 8), ("n_scales", C.c_int), ("tilt_set", C.c_double * 8), ("n_tilts", C.c_int),
                ("phi", C.c_double), ("initSigma", C.c_double), ("doBlur", C.c_int), ("fginn_ratio", C.c_double),
                ("half_orientation", C.c_int), ("fginn_ratio_half", C.c_double), ("dist_threshold", C.c_double),
                ("dist_threshold_half", C.c_double)]

    @staticmethod
    def make(tilts, phi, scales=(1.0,), init_sigma=0.2, do_blur=1, fginn=0.8, half_orientation=0, fginn_half=0.0, dist=0.0, dist_half=0.0):
        """half_orientation: the step's descriptor list names a Half* descriptor; fginn_half > 0: HalfRootSIFT lists are built and
        matched too (SeparateDescriptors = RootSIFT,HalfRootSIFT)."""
        s = LadderStep()
        for i, v in enumerate(scales):
            s.scale_set[i] = v
        for i, v in enumerate(tilts):
            s.tilt_set[i] = v
        s.n_scales, s.n_tilts, s.phi, s.initSigma, s.doBlur, s.fginn_ratio = len(scales), len(tilts), phi, init_sigma, do_blur, fginn
        s.half_orientation, s.fginn_ratio_half = half_orientation, fginn_half
        s.dist_threshold, s.dist_threshold_half = dist, dist_half     # DistanceThreshold: MatchFLANNDistance replaces the FGINN list
        return s


def iters_mods_steps(half=True):
    """[HessianAffine2] and [HessianAffine3] of build/iters_MODS.ini (the HessianAffine steps of the MODS ladder):
    Descriptors = RootSIFT,HalfRootSIFT with FGINNThreshold = 0.8 for the first of them only, i.e. the orientation runs in
    doHalfSIFT mode and the RootSIFT lists are matched (the reference reads the missing second threshold past the end of a
    one-element vector; it is 0 here: HalfRootSIFT lists are not matched)."""
    return [LadderStep.make((1, 2, 4, 6, 8), 360.0, half_orientation=int(half)), LadderStep.make((1, 2, 4, 6, 8), 120.0, half_orientation=int(half))]


class LadderResult(C.Structure):
    _fields_ = [("steps_done", C.c_int), ("n_views", C.c_int), ("n_detected", C.c_int * 2), ("n_described", C.c_int * 2),
                ("n_unoriented", C.c_int * 2), ("n_tentatives", C.c_int), ("n_unique", C.c_int), ("n_inliers", C.c_int),
                ("ransac_samples", C.c_int), ("ransac_lo", C.c_int), ("ransac_rejects", C.c_int), ("H", C.c_double * 9),
                ("ms_detect_describe", C.c_double), ("ms_match", C.c_double), ("ms_duplicates", C.c_double),
                ("ms_ransac", C.c_double), ("gt_true", C.c_int), ("gt_ransac_inliers", C.c_int), ("gt_true_of_ransac", C.c_int)]


def hmatrix_filter(u6, H, ransac=None):
    """HMatrixFiltering (matching.cpp:917-1012): mask of the correspondences within err_threshold of the row-major homography H."""
    u = np.ascontiguousarray(u6, np.float64)
    Hm = np.ascontiguousarray(H, np.float64).reshape(9)
    par = ransac or RansacParams.default()
    mask = np.zeros(max(len(u), 1), np.uint8)
    n = C.c_int()
    _check(lib().mods_hmatrix_filter(u.ctypes.data_as(C.c_void_p), len(u), Hm.ctypes.data_as(C.c_void_p), C.byref(par),
                                     mask.ctypes.data_as(C.c_void_p), C.byref(n)))
    return mask[:len(u)].astype(bool), n.value


def view_schedule(step, history):
    """SetVSPars for one step; `history` (list of (zoom, tilt, phi)) is extended in place.  Returns the new views."""
    prev = (ViewPar * 1024)()
    for i, v in enumerate(history):
        prev[i] = ViewPar(*v)
    n_prev = C.c_int(len(history))
    out = (ViewPar * 256)()
    n = lib().mods_view_schedule(step.scale_set, step.n_scales, step.tilt_set, step.n_tilts, C.c_double(step.phi), prev,
                                 C.byref(n_prev), 1024, out, 256)
    _check(min(n, 0))
    views = [(out[i].zoom, out[i].tilt, out[i].phi) for i in range(n)]
    history.extend(views)
    return views


class ImgRep:
    """Accumulated regions of one image in HBM (ImageRepresentation::AddRegions)."""

    def __init__(self, ctx, capacity=1 << 19):
        self.h = C.c_void_p()
        self.ctx = ctx
        _check(lib().mods_imgrep_create(ctx.h, capacity, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().mods_imgrep_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        _check(lib().mods_imgrep_clear(self.h))

    def __len__(self):
        return lib().mods_imgrep_count(self.h)

    @property
    def dev_ptr(self):
        lib().mods_imgrep_regions_dev.restype = C.c_void_p
        return lib().mods_imgrep_regions_dev(self.h)

    def append_ctx(self, img=0):
        _check(lib().mods_imgrep_append_ctx(self.h, self.ctx.h, img))

    def append_dev(self, dev_ptr, n):
        _check(lib().mods_imgrep_append_dev(self.h, C.c_void_p(dev_ptr), n))

    def append_host(self, regions):
        r = np.ascontiguousarray(regions)
        _check(lib().mods_imgrep_append_host(self.h, r.ctypes.data_as(C.c_void_p), len(r)))

    def fetch(self, begin=0, count=None):
        count = len(self) - begin if count is None else count
        out = np.zeros(max(count, 1), REGION_DTYPE)
        _check(lib().mods_imgrep_fetch(self.h, begin, count, out.ctypes.data_as(C.c_void_p)))
        return out[:count].copy()


def match_reps(ctx, q, t, q_begin=0, q_end=None, ratio=0.8, contrad=10.0, nn=50):
    """MatchImgReps for the query slice [q_begin, q_end) of ImgRep q against all of ImgRep t."""
    q_end = len(q) if q_end is None else q_end
    cap = max(q_end - q_begin, 1)
    out = np.zeros(cap, TENT_DTYPE)
    u6 = np.zeros((cap, 6), np.float64)
    laf = np.zeros((cap, 14), np.float64)
    n = C.c_int()
    _check(lib().mods_match_reps(ctx.h, q.h, q_begin, q_end, t.h, C.c_double(ratio), C.c_double(contrad), nn,
                                 out.ctypes.data_as(C.c_void_p), u6.ctypes.data_as(C.c_void_p), laf.ctypes.data_as(C.c_void_p),
                                 cap, C.byref(n)))
    return out[:n.value].copy(), u6[:n.value].copy(), laf[:n.value].copy()


def match_ladder_dev(ctx, img_ptr, w, h, steps, rep1, rep2, params=None, min_matches=15, max_matches=0, img2_ptr=None, w2=None, h2=None):
    """The step loop of mods.cpp:202-383 on one GPU; img_ptr: [2][h][w] fp32 in HBM (or two separate images)."""
    if img2_ptr is None:
        img2_ptr, w2, h2 = img_ptr + 4 * w * h, w, h
    params = params or PairParams.default()
    arr = (LadderStep * len(steps))(*steps)
    res = LadderResult()
    m = np.zeros((max(max_matches, 1), 4), np.float64)
    _check(lib().mods_match_ladder_dev(ctx.h, C.c_void_p(img_ptr), w, h, C.c_void_p(img2_ptr), w2, h2, arr, len(steps), min_matches,
                                       C.byref(params), rep1.h, rep2.h,
                                       C.byref(res), m.ctypes.data_as(C.c_void_p) if max_matches else None, max_matches))
    return res, m[:min(res.n_inliers, max_matches)]


class LadderGroup(C.Structure):
    """[Matching<i>] GroupDetectors / GroupDescriptors of one step (mods_ladder_group)."""
    _fields_ = [("n_dets", C.c_int), ("dets", C.c_int * 8), ("fginn_ratio", C.c_double), ("fginn_ratio_half", C.c_double),
                ("dist_threshold", C.c_double), ("dist_threshold_half", C.c_double)]

    @staticmethod
    def make(dets=(), ratio=0.0, ratio_half=-1.0, dist=0.0, dist_half=0.0):
        g = LadderGroup()
        g.n_dets = len(dets)
        for i, d in enumerate(dets):
            g.dets[i] = d
        g.fginn_ratio, g.fginn_ratio_half, g.dist_threshold, g.dist_threshold_half = ratio, ratio_half, dist, dist_half
        return g


def match_ladder_dets_dev(ctx, img_ptr, w, h, det_steps, det_params, reps1, reps2, params=None, min_matches=15, max_matches=0,
                          groups=None, group_pos=0):
    """mods_match_ladder_dets_dev: det_steps[d] = the steps of detector d (LadderStep, or None where the detector has no section),
    det_params[d] its HessAffParams; list the detectors sorted by name.  img_ptr: [2][h][w] fp32 in HBM."""
    n_det, n_steps = len(det_steps), max(len(x) for x in det_steps)
    arr = (LadderStep * (n_steps * n_det))()
    for d, steps in enumerate(det_steps):
        for i in range(n_steps):
            st = steps[i] if i < len(steps) and steps[i] is not None else None
            if st is None:
                st = LadderStep(); st.n_tilts = st.n_scales = -1
            arr[i * n_det + d] = st
    dets = (HessAffParams * n_det)(*det_params)
    r1 = (C.c_void_p * n_det)(*[r.h for r in reps1]); r2 = (C.c_void_p * n_det)(*[r.h for r in reps2])
    params = params or PairParams.default()
    res = LadderResult()
    m = np.zeros((max(max_matches, 1), 4), np.float64)
    garr = None
    if groups is not None:
        garr = (LadderGroup * n_steps)(*[g if g is not None else LadderGroup.make() for g in list(groups) + [None] * (n_steps - len(groups))])
    _check(lib().mods_match_ladder_groups_dev(ctx.h, C.c_void_p(img_ptr), w, h, C.c_void_p(img_ptr + 4 * w * h), w, h, arr, dets, garr, group_pos,
                                              n_steps, n_det, min_matches, C.byref(params), r1, r2, C.byref(res),
                                              m.ctypes.data_as(C.c_void_p) if max_matches else None, max_matches))
    return res, m[:min(res.n_inliers, max_matches)]


class Multi:
    """mods_multi_*: one hard pair on several GPUs (views sharded, one all-gather of the regions per step)."""

    def __init__(self, devices, w, h, rep_capacity=1 << 19):
        self.h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().mods_multi_create(arr, len(devices), w, h, rep_capacity, C.byref(self.h)))
        self.uses_rccl = bool(lib().mods_multi_uses_rccl(self.h))

    def match_ladder(self, img1, img2, steps, params=None, min_matches=15, max_matches=0):
        a = np.ascontiguousarray(img1, np.float32); b = np.ascontiguousarray(img2, np.float32)
        params = params or PairParams.default()
        arr = (LadderStep * len(steps))(*steps)
        res = LadderResult()
        m = np.zeros((max(max_matches, 1), 4), np.float64)
        _check(lib().mods_match_ladder_multi(self.h, _fp(a), a.shape[1], a.shape[0], _fp(b), b.shape[1], b.shape[0], arr, len(steps),
                                             min_matches, C.byref(params), C.byref(res),
                                             m.ctypes.data_as(C.c_void_p) if max_matches else None, max_matches))
        return res, m[:min(res.n_inliers, max_matches)]

    def match_ladder_dets(self, img1, img2, det_steps, det_params, params=None, min_matches=15, max_matches=0, groups=None, group_pos=0):
        """mods_match_ladder_groups_multi: the arguments of match_ladder_dets_dev with the images in host memory."""
        a = np.ascontiguousarray(img1, np.float32); b = np.ascontiguousarray(img2, np.float32)
        n_det, n_steps = len(det_steps), max(len(x) for x in det_steps)
        arr = (LadderStep * (n_steps * n_det))()
        for d, steps in enumerate(det_steps):
            for i in range(n_steps):
                st = steps[i] if i < len(steps) and steps[i] is not None else None
                if st is None:
                    st = LadderStep(); st.n_tilts = st.n_scales = -1
                arr[i * n_det + d] = st
        dets = (HessAffParams * n_det)(*det_params)
        params = params or PairParams.default()
        res = LadderResult()
        m = np.zeros((max(max_matches, 1), 4), np.float64)
        garr = None
        if groups is not None:
            garr = (LadderGroup * n_steps)(*[g if g is not None else LadderGroup.make() for g in list(groups) + [None] * (n_steps - len(groups))])
        _check(lib().mods_match_ladder_groups_multi(self.h, _fp(a), a.shape[1], a.shape[0], _fp(b), b.shape[1], b.shape[0], arr, dets, garr, group_pos,
                                                    n_steps, n_det, min_matches, C.byref(params), C.byref(res),
                                                    m.ctypes.data_as(C.c_void_p) if max_matches else None, max_matches))
        return res, m[:min(res.n_inliers, max_matches)]

    def bank(self, image, det=0):
        """Regions of image 1 (0) / 2 (1) of detector `det` after a run."""
        lib().mods_multi_bank_det.restype = C.c_void_p
        hnd = C.c_void_p(lib().mods_multi_bank_det(self.h, image, det))
        n = lib().mods_imgrep_count(hnd)
        out = np.zeros(max(n, 1), REGION_DTYPE)
        if n:
            _check(lib().mods_imgrep_fetch(hnd, 0, n, out.ctypes.data_as(C.c_void_p)))
        return out[:n].copy()

    def close(self):
        if self.h:
            lib().mods_multi_destroy(self.h)
            self.h = C.c_void_p()


def multi_assign(areas, n_dev):
    """Host logic of the view sharding: owner device of every view job (largest first onto the least loaded device)."""
    a = np.ascontiguousarray(areas, np.float64)
    owner = np.zeros(len(a), np.int32)
    _check(lib().mods_multi_assign(a.ctypes.data_as(C.c_void_p), len(a), n_dev, owner.ctypes.data_as(C.c_void_p)))
    return owner


def match_verify_reps(ctx, rep1, rep2, fginn_ratio=0.8, params=None, max_matches=0):
    """Pre-extracted mode (mods.cpp:196-229): match + duplicate filter + verification on banks filled by the caller."""
    params = params or PairParams.default()
    res = LadderResult()
    m = np.zeros((max(max_matches, 1), 4), np.float64)
    _check(lib().mods_match_verify_reps(ctx.h, rep1.h, rep2.h, C.c_double(fginn_ratio), C.byref(params), C.byref(res),
                                        m.ctypes.data_as(C.c_void_p) if max_matches else None, max_matches))
    return res, m[:min(res.n_inliers, max_matches)]


# ---- one pair end to end ------------------------------------------------------------------------------
class PairParams(C.Structure):
    _fields_ = [("det", HessAffParams), ("desc", DescribeParams), ("fginn_ratio", C.c_double),
                ("contradDist", C.c_double), ("nn", C.c_int), ("dup_before_ransac", C.c_int),
                ("dup_dist", C.c_double), ("dup_mode", C.c_int), ("ransac", RansacParams)]

    @staticmethod
    def default():
        """build/config_affori_classic.ini + build/iters_HessianSIFT.ini with vector_matcher = linear."""
        return PairParams(HessAffParams.default(), DescribeParams.default(), 0.8, 10.0, 50, 1, 2.0, 1,
                          RansacParams.default())


class PairResult(C.Structure):
    _fields_ = [("n_detected", C.c_int * 2), ("n_described", C.c_int * 2), ("n_tentatives", C.c_int),
                ("n_unique", C.c_int), ("n_inliers", C.c_int), ("ransac_samples", C.c_int), ("ransac_lo", C.c_int),
                ("ransac_rejects", C.c_int), ("H", C.c_double * 9), ("ms_detect_describe", C.c_double),
                ("ms_match", C.c_double), ("ms_duplicates", C.c_double), ("ms_ransac", C.c_double)]


def match_pair_dev(ctx, dev_ptr, w, h, params=None, max_matches=0):
    """Whole hot path on [2][h][w] fp32 images resident in HBM; returns (PairResult, matches[n,4])."""
    params = params or PairParams.default()
    res = PairResult()
    m = np.zeros((max(max_matches, 1), 4), np.float64)
    _check(lib().mods_match_pair_dev(ctx.h, C.c_void_p(dev_ptr), w, h, w, C.byref(params), C.byref(res),
                                     m.ctypes.data_as(C.c_void_p) if max_matches else None, max_matches))
    return res, m[:min(res.n_inliers, max_matches)]


class PinnedBuffer:
    """Page-locked host memory (mods_host_alloc) viewed as a numpy array: uploads from it are asynchronous."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = C.c_void_p()
        _check(lib().mods_host_alloc(C.c_size_t(self.nbytes), C.byref(self.ptr)))
        buf = (C.c_char * self.nbytes).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)

    def close(self):
        if self.ptr:
            self.array = None
            lib().mods_host_free(self.ptr)
            self.ptr = C.c_void_p()


class Pipeline:
    """mods_pipeline_*: GPU workers (detect/describe/match) overlapped with verify workers (duplicate
    filter + LO-RANSAC) across pairs; results in submission order."""

    def __init__(self, device, w, h, params=None, gpu_workers=1, verify_workers=1, pairs_per_batch=1):
        self.params = params or PairParams.default()
        self.h = C.c_void_p()
        _check(lib().mods_pipeline_create_ex(device, w, h, C.byref(self.params), gpu_workers, verify_workers, pairs_per_batch,
                                             C.byref(self.h)))
        self.capacity = lib().mods_pipeline_capacity(self.h)

    def submit(self, dev_ptr, tag=0):
        _check(lib().mods_pipeline_submit(self.h, C.c_void_p(dev_ptr), C.c_long(tag)))

    def submit_host(self, host_ptr, tag=0, u8=False):
        """The pair in host memory ([2][h][w] fp32, or 8-bit grey with u8=True): uploaded on the worker's stream."""
        fn = lib().mods_pipeline_submit_host_u8 if u8 else lib().mods_pipeline_submit_host
        _check(fn(self.h, C.c_void_p(host_ptr), C.c_long(tag)))

    def next(self):
        res, tag = PairResult(), C.c_long()
        _check(lib().mods_pipeline_next(self.h, C.byref(res), C.byref(tag)))
        return res, tag.value

    def next_matches(self, max_matches=1 << 16):
        """(PairResult, tag, matches[n,4]) of the oldest submitted pair: the verified matches as mods_match_pair_dev returns them."""
        res, tag = PairResult(), C.c_long()
        m = np.zeros((max_matches, 4), np.float64)
        _check(lib().mods_pipeline_next_matches(self.h, C.byref(res), C.byref(tag), m.ctypes.data_as(C.c_void_p), max_matches))
        return res, tag.value, m[:min(res.n_inliers, max_matches)]

    def timing_enable(self, stages):
        mask = 0
        for s in stages:
            mask |= 1 << STAGES.index(s)
        _check(lib().mods_pipeline_timing_enable(self.h, mask))

    def timing_read(self, stage):
        ms, n, by = C.c_double(), C.c_int(), C.c_double()
        _check(lib().mods_pipeline_timing_read(self.h, STAGES.index(stage), C.byref(ms), C.byref(n), C.byref(by)))
        return ms.value, n.value, by.value

    def graph_replays(self):
        lib().mods_pipeline_graph_replays.restype = C.c_long
        return int(lib().mods_pipeline_graph_replays(self.h))

    def cpu_seconds(self, reset=False):
        """(GPU workers, verify workers): CPU seconds their threads spent inside their stages since the last reset"""
        g, v = C.c_double(), C.c_double()
        _check(lib().mods_pipeline_cpu_seconds(self.h, C.byref(g), C.byref(v), 1 if reset else 0))
        return g.value, v.value

    def close(self):
        if self.h:
            lib().mods_pipeline_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
