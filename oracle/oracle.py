"""ctypes binding of oracle/libvs_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
video_stabilizer_amd/.  The oracle is the CPU restatement of the reference path described in
oracle/vs_oracle.h ("parity unpinned": the reference itself cannot be built in this image).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("VS_ORACLE_LIB", os.path.join(_HERE, "libvs_oracle.so"))   # VS_ORACLE_LIB: the sanitizer build (make -C oracle asan-test)


class Transform(C.Structure):
    _fields_ = [("A", C.c_double), ("B", C.c_double), ("TX", C.c_double), ("TY", C.c_double)]

    def tup(self):
        return (self.A, self.B, self.TX, self.TY)

    @staticmethod
    def of(A=0.0, B=0.0, TX=0.0, TY=0.0):
        return Transform(float(A), float(B), float(TX), float(TY))


class Point(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double)]


class AlignerParams(C.Structure):
    _fields_ = [("phase_correlate", C.c_int), ("phase_correlate_threshold", C.c_double),
                ("threshold", C.c_double), ("smallest_fraction", C.c_float), ("max_iters", C.c_int),
                ("pyramid_min_width", C.c_int), ("pyramid_min_height", C.c_int),
                ("max_displacement", C.c_double)]


class StabilizerParams(C.Structure):
    _fields_ = [("aligner", AlignerParams), ("lag", C.c_int), ("smoother_memory", C.c_int),
                ("lambda_", C.c_double), ("enable_smoother", C.c_int), ("crop_pixels", C.c_int),
                ("min_disp", C.c_double), ("max_disp", C.c_double), ("min_decay", C.c_double),
                ("max_decay", C.c_double), ("warp_mode", C.c_int), ("warp_border", C.c_int)]


class AlignDebug(C.Structure):
    _fields_ = [("levels", C.c_int), ("fail_reason", C.c_int), ("fail_level", C.c_int),
                ("iterations", C.c_int * 16), ("tile_size", C.c_int * 16),
                ("selected_x", C.c_int * 16), ("selected_y", C.c_int * 16),
                ("condition", C.c_double * 16), ("level_transform", Transform * 16),
                ("phase_dx", C.c_double), ("phase_dy", C.c_double), ("phase_response", C.c_double)]


WARP_LANCZOS2, WARP_BILINEAR, WARP_LANCZOS2_CONTRACTED, WARP_LANCZOS2_SEPARABLE, WARP_BILINEAR_CV = 0, 1, 2, 3, 4
SELECT_STL, SELECT_STABLE = 0, 1
BORDER_CLAMP, BORDER_CONSTANT = 0, 1
FMT_GRAY8, FMT_BGR8 = 0, 1
FMT_BGR10, FMT_BGR12, FMT_BGR16_FULL = 2, 3, 4

_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, i32, f32, f64 = C.c_void_p, C.c_int, C.c_float, C.c_double
    TP = C.POINTER(Transform)
    sig = {
        "vso_aligner_params_default": (None, [C.POINTER(AlignerParams)]),
        "vso_stabilizer_params_default": (None, [C.POINTER(StabilizerParams)]),
        "vso_lanczos2": (f32, [f32]),
        "vso_pyr_down": (None, [vp, i32, i32, i32, vp, i32, i32, i32]),
        "vso_grad_xy": (None, [vp, i32, i32, i32, vp, vp]),
        "vso_tile_size": (i32, [i32, i32]),
        "vso_grad_argmax": (None, [vp, vp, i32, i32, i32, vp, vp]),
        "vso_sparse_jac": (None, [vp, vp, i32, i32, vp, vp, i32, i32, vp, vp]),
        "vso_sparse_warpdiff": (None, [vp, vp, i32, i32, i32, vp, i32, i32, f32, f32, f32, f32, vp]),
        "vso_sparse_ica": (None, [vp, vp, i32, i32, i32, vp, i32, vp, i32, vp, vp, f32, f32, f32, f32, vp]),
        "vso_image_warp": (None, [vp, i32, i32, i32, f32, f32, f32, f32, vp, i32, i32]),
        "vso_ul_params_sparse": (None, [TP, i32, i32, vp]),
        "vso_ul_params_warp": (None, [TP, i32, i32, vp]),
        "vso_cv_inverse_matrix": (None, [TP, i32, i32, vp]),
        "vso_bgr_image_warp": (None, [vp, i32, i32, i32, i32, i32, TP, i32, i32, i32, vp, i32]),
        "vso_bgr_image_warp_f32": (None, [vp, i32, i32, i32, i32, i32, TP, i32, i32, vp, i32]),
        "vso_bgr_to_gray": (None, [vp, i32, i32, i32, i32, i32, vp, i32]),
        "vso_format_bits": (i32, [i32]),
        "vso_set_threads": (None, [i32]),
        "vso_get_threads": (i32, []),
        "vso_transform_inverse": (Transform, [TP]),
        "vso_transform_compose": (Transform, [TP, TP]),
        "vso_transform_warp": (Point, [TP, Point]),
        "vso_transform_warp_center": (Point, [TP, Point, f64, f64]),
        "vso_transform_max_corner_displacement": (f64, [TP, f64, f64]),
        "vso_select_smallest": (i32, [vp, i32, i32, f32, vp]),
        "vso_select_smallest_stable": (i32, [vp, i32, i32, f32, vp]),
        "vso_aligner_set_select_rule": (i32, [vp, i32]),
        "vso_stabilizer_set_select_rule": (i32, [vp, i32]),
        "vso_nth_element_killer": (i32, [i32, i32, f32, vp]),
        "vso_nth_element_hits_depth_limit": (i32, [vp, i32, f32]),
        "vso_hessian": (None, [vp, i32, vp, i32, vp]),
        "vso_condition_and_invert": (f64, [vp, vp]),
        "vso_optimal_dft_size": (i32, [i32]),
        "vso_fft_plan": (i32, [i32, C.POINTER(i32)]),
        "vso_fft_c2c": (i32, [vp, i32, i32]),
        "vso_phase_surface": (i32, [vp, vp, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]),
        "vso_phase_peak": (None, [vp, i32, i32, C.POINTER(f64), C.POINTER(f64), C.POINTER(f64)]),
        "vso_phase_correlate": (i32, [vp, vp, i32, i32, C.POINTER(f64), C.POINTER(f64), C.POINTER(f64)]),
        "vso_phase_correlate_u8": (i32, [vp, vp, i32, i32, i32, C.POINTER(f64), C.POINTER(f64), C.POINTER(f64)]),
        "vso_tvl1_smooth": (None, [vp, i32, f64, i32, vp]),
        "vso_smoother_create": (vp, [i32, i32, f64]),
        "vso_smoother_destroy": (None, [vp]),
        "vso_smoother_update": (i32, [vp, TP, TP]),
        "vso_aligner_create": (vp, []),
        "vso_aligner_destroy": (None, [vp]),
        "vso_aligner_align_next": (i32, [vp, vp, i32, i32, i32, i32, C.POINTER(AlignerParams), TP]),
        "vso_aligner_debug": (C.POINTER(AlignDebug), [vp]),
        "vso_aligner_level_dims": (i32, [vp, i32] + [C.POINTER(i32)] * 5),
        "vso_aligner_level_image": (vp, [vp, i32, i32]),
        "vso_aligner_level_argmax": (vp, [vp, i32, i32]),
        "vso_aligner_level_jacobian": (vp, [vp, i32, i32]),
        "vso_stabilizer_create": (vp, [C.POINTER(StabilizerParams)]),
        "vso_stabilizer_destroy": (None, [vp]),
        "vso_stabilizer_process": (i32, [vp, vp, i32, i32, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]),
        "vso_stabilizer_state": (None, [vp, TP, TP, C.POINTER(i32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def aligner_params(**kw):
    p = AlignerParams()
    lib().vso_aligner_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def stabilizer_params(**kw):
    p = StabilizerParams()
    lib().vso_stabilizer_params_default(C.byref(p))
    for k, v in kw.items():
        if hasattr(p.aligner, k) and not hasattr(p, k):
            setattr(p.aligner, k, v)
        else:
            setattr(p, k, v)
    return p


def set_threads(n):
    """worker threads of the row-parallel stage loops (1 = serial, the default); results are identical for every n"""
    lib().vso_set_threads(int(n))


def lanczos2(x):
    return lib().vso_lanczos2(float(x))


def pyr_down(img):
    img = _c(img, np.uint8)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.uint8)
    lib().vso_pyr_down(_p(img), w, h, w, _p(out), w // 2, h // 2, w // 2)
    return out


def grad_xy(img):
    img = _c(img, np.uint8)
    h, w = img.shape
    gx = np.empty((h, w), np.float32)
    gy = np.empty((h, w), np.float32)
    lib().vso_grad_xy(_p(img), w, h, w, _p(gx), _p(gy))
    return gx, gy


def tile_size(w, h):
    return lib().vso_tile_size(w, h)


def grad_argmax(gx, gy, ts=None):
    """returns (ts, lmx, lmy) with lm shaped (2, ty, tx): [0] = x coords, [1] = y coords"""
    gx = _c(gx, np.float32)
    gy = _c(gy, np.float32)
    h, w = gx.shape
    if ts is None:
        ts = tile_size(w, h)
    tx, ty = w // ts, h // ts
    lmx = np.empty((2, ty, tx), np.uint16)
    lmy = np.empty((2, ty, tx), np.uint16)
    lib().vso_grad_argmax(_p(gx), _p(gy), w, h, ts, _p(lmx), _p(lmy))
    return ts, lmx, lmy


def sparse_jac(gx, gy, lmx, lmy):
    gx = _c(gx, np.float32)
    gy = _c(gy, np.float32)
    lmx = _c(lmx, np.uint16)
    lmy = _c(lmy, np.uint16)
    h, w = gx.shape
    _, ty, tx = lmx.shape
    jx = np.empty((4, ty, tx), np.float32)
    jy = np.empty((4, ty, tx), np.float32)
    lib().vso_sparse_jac(_p(gx), _p(gy), w, h, _p(lmx), _p(lmy), tx, ty, _p(jx), _p(jy))
    return jx, jy


def ul_params_sparse(t, w, h):
    out = np.empty(4, np.float32)
    lib().vso_ul_params_sparse(C.byref(t), w, h, _p(out))
    return out


def ul_params_warp(t, w, h):
    out = np.empty(4, np.float32)
    lib().vso_ul_params_warp(C.byref(t), w, h, _p(out))
    return out


def cv_inverse_matrix(t, w, h):
    """VSO_WARP_BILINEAR_CV's output -> source matrix for the FORWARD transform t (imgproc.cpp:457-466 + cv::warpAffine's inversion)"""
    out = np.empty(6, np.float64)
    lib().vso_cv_inverse_matrix(C.byref(t), w, h, _p(out))
    return out


def sparse_warpdiff_raw(tmpl, key, lm, A, B, TX, TY):
    tmpl = _c(tmpl, np.uint8)
    key = _c(key, np.uint8)
    lm = _c(lm, np.uint16)
    h, w = key.shape
    _, ty, tx = lm.shape
    out = np.empty((ty, tx), np.uint16)
    lib().vso_sparse_warpdiff(_p(tmpl), _p(key), w, h, w, _p(lm), tx, ty, A, B, TX, TY, _p(out))
    return out


def sparse_warpdiff(tmpl, key, lm, t):
    """SparseWarpDiff wrapper (imgproc.cpp:80-106): t is the centre-based transform"""
    h, w = key.shape
    p = ul_params_sparse(t, w, h)
    return sparse_warpdiff_raw(tmpl, key, lm, float(p[0]), float(p[1]), float(p[2]), float(p[3]))


def sparse_ica_raw(tmpl, key, selx, sely, jacx, jacy, A, B, TX, TY):
    """selx/sely: (2, N) u16 (row 0 = xs, row 1 = ys); jacx/jacy: (4, N) f32"""
    tmpl = _c(tmpl, np.uint8)
    key = _c(key, np.uint8)
    selx = _c(selx, np.uint16)
    sely = _c(sely, np.uint16)
    jacx = _c(jacx, np.float32)
    jacy = _c(jacy, np.float32)
    h, w = key.shape
    out = np.empty(4, np.float64)
    lib().vso_sparse_ica(_p(tmpl), _p(key), w, h, w, _p(selx), selx.shape[1], _p(sely), sely.shape[1],
                         _p(jacx), _p(jacy), A, B, TX, TY, _p(out))
    return out


def sparse_ica(tmpl, key, selx, sely, jacx, jacy, t):
    h, w = key.shape
    p = ul_params_sparse(t, w, h)
    return sparse_ica_raw(tmpl, key, selx, sely, jacx, jacy, float(p[0]), float(p[1]), float(p[2]), float(p[3]))


def image_warp_raw(img, A, B, TX, TY, out_shape=None):
    img = _c(img, np.uint8)
    h, w = img.shape
    oh, ow = out_shape if out_shape else (h, w)
    out = np.empty((oh, ow), np.float32)
    lib().vso_image_warp(_p(img), w, h, w, A, B, TX, TY, _p(out), ow, oh)
    return out


def image_warp(img, t):
    """ImageWarp wrapper (imgproc.cpp:116-133)"""
    h, w = img.shape
    p = ul_params_warp(t, w, h)
    return image_warp_raw(img, float(p[0]), float(p[1]), float(p[2]), float(p[3]))


def bgr_image_warp(src, t, mode=WARP_LANCZOS2, border=BORDER_CLAMP, max_value=None, f32=False):
    """src: (h, w, c) uint8 or uint16, interleaved"""
    src = np.ascontiguousarray(src)
    assert src.dtype in (np.uint8, np.uint16) and src.ndim == 3
    h, w, c = src.shape
    bits = 8 if src.dtype == np.uint8 else 16
    if f32:
        out = np.empty((h, w, c), np.float32)
        lib().vso_bgr_image_warp_f32(_p(src), w, h, w * c, c, bits, C.byref(t), mode, border, _p(out), w * c)
        return out
    if max_value is None:
        max_value = 255 if bits == 8 else 65535
    out = np.empty_like(src)
    lib().vso_bgr_image_warp(_p(src), w, h, w * c, c, bits, C.byref(t), mode, border, max_value, _p(out), w * c)
    return out


def bgr_to_gray(src, shift_to_8=None):
    src = np.ascontiguousarray(src)
    h, w, _ = src.shape
    bits = 8 if src.dtype == np.uint8 else 16
    if shift_to_8 is None:
        shift_to_8 = 0 if bits == 8 else 2
    out = np.empty((h, w), np.uint8)
    lib().vso_bgr_to_gray(_p(src), w, h, w * 3, bits, shift_to_8, _p(out), w)
    return out


def t_inverse(t):
    return lib().vso_transform_inverse(C.byref(t))


def t_compose(t1, t2):
    return lib().vso_transform_compose(C.byref(t1), C.byref(t2))


def t_warp(t, x, y, center=None):
    if center is None:
        p = lib().vso_transform_warp(C.byref(t), Point(x, y))
    else:
        p = lib().vso_transform_warp_center(C.byref(t), Point(x, y), center[0], center[1])
    return p.x, p.y


def t_max_corner_displacement(t, w, h):
    return lib().vso_transform_max_corner_displacement(C.byref(t), w, h)


def select_smallest(warpdiff, fraction=0.8):
    wd = _c(warpdiff, np.uint16)
    ty, tx = wd.shape
    idx = np.empty(tx * ty, np.int32)
    n = lib().vso_select_smallest(_p(wd), tx, ty, fraction, _p(idx))
    return idx[:n].copy()


def select_smallest_stable(warpdiff, fraction=0.8):
    """the documented STL-independent rule: smallest by (abs_delta, tile index), survivors in ascending tile order"""
    wd = _c(warpdiff, np.uint16)
    ty, tx = wd.shape
    idx = np.empty(tx * ty, np.int32)
    n = lib().vso_select_smallest_stable(_p(wd), tx, ty, fraction, _p(idx))
    return idx[:n].copy()


def nth_element_killer(tx, ty, fraction=0.8):
    """(ty, tx) uint16 table on which std::nth_element(begin, begin + n*fraction, end) runs out of its depth budget"""
    out = np.empty((ty, tx), np.uint16)
    r = lib().vso_nth_element_killer(tx, ty, fraction, _p(out))
    if r < 0:
        raise ValueError("table does not fit 16 bits")
    return out


def nth_element_hits_depth_limit(warpdiff, fraction=0.8):
    wd = _c(warpdiff, np.uint16)
    return bool(lib().vso_nth_element_hits_depth_limit(_p(wd), wd.size, fraction))


def optimal_dft_size(n):
    return lib().vso_optimal_dft_size(int(n))


def fft_plan(n):
    r = (C.c_int * 32)()
    k = lib().vso_fft_plan(int(n), r)
    return None if k < 0 else list(r[:k])


def fft_c2c(x, inverse=False):
    """the build's mixed-radix transform on a complex64 vector (inverse unscaled)"""
    x = np.array(x, np.complex64, copy=True)
    if lib().vso_fft_c2c(_p(x), x.size, 1 if inverse else 0) != 0:
        raise ValueError("size %d has a prime factor other than 2, 3, 5" % x.size)
    return x


def phase_surface(a, b):
    """unshifted, unscaled correlation surface (M, N) of two equal-size float images"""
    a = _c(a, np.float32)
    b = _c(b, np.float32)
    h, w = a.shape
    M, N = C.c_int(), C.c_int()
    lib().vso_phase_surface(_p(a), _p(b), w, h, None, C.byref(M), C.byref(N))
    out = np.empty((M.value, N.value), np.float32)
    lib().vso_phase_surface(_p(a), _p(b), w, h, _p(out), C.byref(M), C.byref(N))
    return out


def phase_peak(surface):
    s = _c(surface, np.float32)
    dx, dy, r = C.c_double(), C.c_double(), C.c_double()
    lib().vso_phase_peak(_p(s), s.shape[0], s.shape[1], C.byref(dx), C.byref(dy), C.byref(r))
    return dx.value, dy.value, r.value


def phase_correlate(a, b):
    """cv::phaseCorrelate(a, b) -> (dx, dy, response); u8 or float images"""
    dx, dy, r = C.c_double(), C.c_double(), C.c_double()
    if np.asarray(a).dtype == np.uint8:
        a = _c(a, np.uint8)
        b = _c(b, np.uint8)
        rc = lib().vso_phase_correlate_u8(_p(a), _p(b), a.shape[1], a.shape[0], a.shape[1], C.byref(dx), C.byref(dy), C.byref(r))
    else:
        a = _c(a, np.float32)
        b = _c(b, np.float32)
        rc = lib().vso_phase_correlate(_p(a), _p(b), a.shape[1], a.shape[0], C.byref(dx), C.byref(dy), C.byref(r))
    if rc != 0:
        raise ValueError("vso_phase_correlate failed")
    return dx.value, dy.value, r.value


def hessian(jacx, jacy):
    jacx = _c(jacx, np.float32)
    jacy = _c(jacy, np.float32)
    H = np.empty((4, 4), np.float64)
    lib().vso_hessian(_p(jacx), jacx.shape[1], _p(jacy), jacy.shape[1], _p(H))
    return H


def condition_and_invert(H):
    H = np.array(H, np.float64, copy=True)
    Hinv = np.empty((4, 4), np.float64)
    cond = lib().vso_condition_and_invert(_p(H), _p(Hinv))
    return cond, H, Hinv


def tvl1_smooth(data, lam, iterations=100):
    d = _c(data, np.float64)
    out = np.empty_like(d)
    lib().vso_tvl1_smooth(_p(d), d.size, lam, iterations, _p(out))
    return out


class Smoother:
    def __init__(self, lag_behind, lag_ahead, lam):
        self.h = lib().vso_smoother_create(lag_behind, lag_ahead, lam)

    def update(self, meas):
        out = Transform()
        ok = lib().vso_smoother_update(self.h, C.byref(meas), C.byref(out))
        return bool(ok), out

    def __del__(self):
        if getattr(self, "h", None):
            lib().vso_smoother_destroy(self.h)
            self.h = None


def _fmt_of(frame):
    if frame.ndim == 2:
        assert frame.dtype == np.uint8
        return FMT_GRAY8
    assert frame.ndim == 3 and frame.shape[2] == 3
    return FMT_BGR8 if frame.dtype == np.uint8 else FMT_BGR10


class Aligner:
    """VideoAligner restatement (alignment.cpp)"""

    def __init__(self, select_rule=0, **params):
        self.h = lib().vso_aligner_create()
        self.params = aligner_params(**params)
        if select_rule:
            self.set_select_rule(select_rule)

    def set_select_rule(self, rule):
        """0: std::nth_element as the reference; 1 (SELECT_STABLE): smallest by (abs_delta, tile index), survivors in tile order"""
        if lib().vso_aligner_set_select_rule(self.h, int(rule)) != 0:
            raise ValueError("select rule %r" % (rule,))

    def align_next(self, frame, fmt=None):
        frame = np.ascontiguousarray(frame)
        fmt = _fmt_of(frame) if fmt is None else fmt
        hh, ww = frame.shape[:2]
        stride = ww * (1 if fmt == FMT_GRAY8 else 3)
        t = Transform()
        r = lib().vso_aligner_align_next(self.h, _p(frame), ww, hh, stride, fmt, C.byref(self.params), C.byref(t))
        if r < 0:
            raise RuntimeError("oracle aligner error %d" % r)
        return bool(r), t

    def debug(self):
        # a snapshot: the struct behind the pointer is overwritten by the next align_next
        return AlignDebug.from_buffer_copy(lib().vso_aligner_debug(self.h).contents)

    def level(self, i):
        w, h, tx, ty, ts = (C.c_int() for _ in range(5))
        if lib().vso_aligner_level_dims(self.h, i, *(C.byref(v) for v in (w, h, tx, ty, ts))) != 0:
            raise IndexError(i)
        w, h, tx, ty, ts = (v.value for v in (w, h, tx, ty, ts))

        def arr(ptr, dtype, shape):
            n = int(np.prod(shape))
            buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()

        d = {"w": w, "h": h, "tx": tx, "ty": ty, "ts": ts}
        d["img"] = [arr(lib().vso_aligner_level_image(self.h, s, i), np.uint8, (h, w)) for s in (0, 1)]
        if tx > 0:
            d["argmax"] = [arr(lib().vso_aligner_level_argmax(self.h, i, s), np.uint16, (2, ty, tx)) for s in (0, 1)]
            d["jac"] = [arr(lib().vso_aligner_level_jacobian(self.h, i, s), np.float32, (4, ty, tx)) for s in (0, 1)]
        return d

    def __del__(self):
        if getattr(self, "h", None):
            lib().vso_aligner_destroy(self.h)
            self.h = None


class Stabilizer:
    """VideoStabilizer restatement (stabilizer.cpp)"""

    def __init__(self, select_rule=0, **params):
        self.params = stabilizer_params(**params)
        self.h = lib().vso_stabilizer_create(C.byref(self.params))
        if select_rule and lib().vso_stabilizer_set_select_rule(self.h, int(select_rule)) != 0:
            raise ValueError("select rule %r" % (select_rule,))

    def process(self, frame, fmt=None):
        frame = np.ascontiguousarray(frame)
        fmt = _fmt_of(frame) if fmt is None else fmt
        hh, ww = frame.shape[:2]
        c = max(self.params.crop_pixels, 0)
        out = np.empty((hh - 2 * c, ww - 2 * c, 3), frame.dtype)
        ow, oh = C.c_int(), C.c_int()
        r = lib().vso_stabilizer_process(self.h, _p(frame), ww, hh, ww * 3, fmt, _p(out), C.byref(ow), C.byref(oh))
        if r < 0:
            raise RuntimeError("oracle stabilizer error %d" % r)
        return out if r == 1 else None

    def state(self):
        m, a, s = Transform(), Transform(), C.c_int()
        lib().vso_stabilizer_state(self.h, C.byref(m), C.byref(a), C.byref(s))
        return m, a, bool(s.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().vso_stabilizer_destroy(self.h)
            self.h = None
