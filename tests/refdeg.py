"""ctypes binding of oracle/_ref/libdegensac_ref.so: the reference's own degensac compiled from
/root/reference by oracle/ref.mk (test infrastructure; the strongest RANSAC oracle)."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libdegensac_ref.so")


class Score(C.Structure):
    _fields_ = [("I", C.c_uint), ("J", C.c_double)]


_lib = None


def available():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH)
        _lib.exp_ransacHcustom.restype = Score
        _lib.oracle_ref_pin_time.argtypes = [C.c_long]
    return _lib


ERR = {"sampson": ("HDs", "HDsi", "HDsidx"), "symm_max": ("HDsSymMax", "HDsiSymMax", "HDsSymidxMax"),
       "symm_sum": ("HDsSym", "HDsiSym", "HDsSymidx")}


def ransac_h(u6, th_sq, conf=0.99, max_sam=1000000, err="sampson", sym_check=1, seed_time=12345, lib_=None):
    """exp_ransacHcustom as LORANSACFiltering calls it (matching.cpp:731): iter_type 4, oriented 1, inlLimit 0."""
    L = lib_ or lib()
    if lib_ is None:
        L.oracle_ref_pin_time(seed_time)
    u = np.ascontiguousarray(u6, np.float64).copy()
    n = len(u)
    H = np.zeros(9, np.float64)
    inl = np.zeros(n, np.uint8)
    data_out = np.zeros(max(18 * n, 8), np.int32)
    resids = C.POINTER(C.c_double)()
    f = [C.cast(getattr(L, name), C.c_void_p) for name in ERR[err]]
    S = L.exp_ransacHcustom(u.ctypes.data_as(C.c_void_p), n, C.c_double(th_sq), C.c_double(conf), max_sam,
                            H.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p), 4,
                            data_out.ctypes.data_as(C.c_void_p), 1, C.c_uint(0), C.byref(resids), f[0], f[1], f[2],
                            sym_check)
    C.CDLL(None).free(resids)
    return dict(I=S.I, J=S.J, H=H, inl=inl, samples=int(data_out[0]), lo=int(data_out[1]), rej=int(data_out[2]))


FERR = {"sampson": ("FDs", "exFDs"), "symm": ("FDsSym", "exFDsSym")}


def ransac_f(u6, th_sq, conf=0.99, max_sam=100000, err="sampson", sym_check=0, do_lo=1, inl_limit=0, seed_time=12345, lib_=None):
    """exp_ransacFcustom as LORANSACFiltering calls it (matching.cpp:722): inlLimit 0."""
    L = lib_ or lib()
    if lib_ is None:
        L.oracle_ref_pin_time(seed_time)
    L.exp_ransacFcustom.restype = C.c_int
    u = np.ascontiguousarray(u6, np.float64).copy()
    n = len(u)
    F = np.zeros(9, np.float64)
    Hb = np.zeros(9, np.float64)
    inl = np.zeros(n, np.uint8)
    data_out = np.zeros(max(18 * n, 16), np.int32)
    resids = C.POINTER(C.c_double)()
    Ih = C.c_int(0)
    fds, exfds = (C.cast(getattr(L, name), C.c_void_p) for name in FERR[err])
    I = L.exp_ransacFcustom(u.ctypes.data_as(C.c_void_p), n, C.c_double(th_sq), C.c_double(conf), max_sam,
                            F.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p), data_out.ctypes.data_as(C.c_void_p),
                            do_lo, C.c_uint(inl_limit), C.byref(resids), Hb.ctypes.data_as(C.c_void_p), C.byref(Ih), exfds, fds,
                            sym_check)
    C.CDLL(None).free(resids)
    return dict(I=I, F=F, inl=inl, samples=int(data_out[0]), lo=int(data_out[1]), Ih=Ih.value, hist=data_out[2:n + 3].copy())
