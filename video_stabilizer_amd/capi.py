"""ctypes binding of libvs_amd.so (the C ABI declared in include/vs_amd.h).

This is plumbing for the Python tests and bench.py: every compute call goes through the C ABI
into the hand-written gfx950 kernels.  There is no fallback -- a missing library raises.

numpy arrays are passed as VS_MEM_HOST; for VS_MEM_DEVICE pass raw device pointers (ints), e.g.
torch tensors' data_ptr().
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VS_AMD_LIB", os.path.join(_HERE, "libvs_amd.so"))

MEM_HOST, MEM_DEVICE = 0, 1
FMT_GRAY8, FMT_BGR8 = 0, 1
FMT_BGR10, FMT_BGR12, FMT_BGR16_FULL = 2, 3, 4      # u16 containers: bits the samples use
WARP_LANCZOS2, WARP_BILINEAR, WARP_LANCZOS2_FAST, WARP_LANCZOS2_SEP, WARP_BILINEAR_CV = 0, 1, 2, 3, 4
BORDER_CLAMP, BORDER_CONSTANT = 0, 1
SELECT_STL_HOST, SELECT_DEVICE, SELECT_STABLE = 0, 1, 2
BATCH_EXCLUSIVE, BATCH_SHARED = 0, 1
ABI_VERSION = 5                                     # VS_ABI_VERSION of include/vs_amd.h


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


class AlignInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("fail_reason", C.c_int32), ("fail_level", C.c_int32),
                ("levels", C.c_int32), ("iterations", C.c_int32 * 16), ("condition", C.c_double * 16),
                ("phase_dx", C.c_double), ("phase_dy", C.c_double), ("phase_response", C.c_double),
                ("selected_x", C.c_int32 * 16), ("selected_y", C.c_int32 * 16), ("level_transform", Transform * 16)]


STAGES = ["ingest", "pyr_down", "keyframe", "warpdiff", "select", "gather", "gn", "phase"]


class StageTimings(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_int64 * 8), ("frames", C.c_int64), ("gn_iterations", C.c_int64)]


class VsError(RuntimeError):
    pass


_vp, _i32, _f32, _f64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
_TP = C.POINTER(Transform)
_IP = C.POINTER(C.c_int)

# name -> (restype, argtypes).  This table is also what tests/test_capi_symbols.py checks
# against include/vs_amd.h.
SIGNATURES = {
    "vs_last_error": (C.c_char_p, []),
    "vs_version": (C.c_char_p, []),
    "vs_abi_version": (_i32, []),
    "vs_sizeof_align_info": (_sz, []),
    "vs_test_fail_alloc": (_i32, [_i32]),
    "vs_debug_bounds_check": (_i32, []),
    "vs_debug_bounds_selftest": (_i32, []),
    "vs_device_count": (_i32, []),
    "vs_shader_clock_probe": (_i32, [C.c_void_p, C.POINTER(C.c_double)]),
    "vs_stream_retire": (_i32, [_vp]),
    "vs_format_bits": (_i32, [_i32]),
    "vs_format_max_value": (_i32, [_i32]),
    "vs_aligner_stream": (_vp, [_vp]),
    "vs_aligner_wait_stream": (_i32, [_vp, _vp]),
    "vs_stabilizer_stream": (_vp, [_vp]),
    "vs_stabilizer_wait_stream": (_i32, [_vp, _vp]),
    "vs_aligner_params_default": (None, [C.POINTER(AlignerParams)]),
    "vs_stabilizer_params_default": (None, [C.POINTER(StabilizerParams)]),
    "vs_transform_inverse": (Transform, [_TP]),
    "vs_transform_compose": (Transform, [_TP, _TP]),
    "vs_transform_warp": (Point, [_TP, Point]),
    "vs_transform_warp_center": (Point, [_TP, Point, _f64, _f64]),
    "vs_transform_max_corner_displacement": (_f64, [_TP, _f64, _f64]),
    "vs_tile_size": (_i32, [_i32, _i32]),
    "vs_ul_params_sparse": (None, [_TP, _i32, _i32, _vp]),
    "vs_ul_params_warp": (None, [_TP, _i32, _i32, _vp]),
    "vs_cv_inverse_matrix": (None, [_TP, _i32, _i32, _vp]),
    "vs_smoother_create": (_vp, [_i32, _i32, _f64]),
    "vs_smoother_destroy": (None, [_vp]),
    "vs_smoother_update": (_i32, [_vp, _TP, _TP]),
    "vs_tvl1_smooth": (None, [_vp, _i32, _f64, _i32, _vp]),
    "vs_calib_copy12": (_i32, [_vp, _vp, _sz, _vp]),
    "vs_pyr_down": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vs_grad_xy": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "vs_grad_argmax": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "vs_sparse_jac": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "vs_keyframe_fused": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "vs_sparse_warpdiff": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _i32, _vp]),
    "vs_sparse_ica": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _i32, _vp]),
    "vs_select_smallest": (_i32, [_vp, _i32, _i32, _i32, _f32, _vp, _vp, _i32, _vp]),
    "vs_select_smallest_stable": (_i32, [_vp, _i32, _i32, _i32, _f32, _vp, _i32, _vp]),
    "vs_phase_correlate": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vs_optimal_dft_size": (_i32, [_i32]),
    "vs_image_warp": (_i32, [_vp, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _i32, _i32, _i32, _vp]),
    "vs_bgr_image_warp": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _TP, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "vs_bgr_image_warp_batch": (_i32, [_vp, _sz, _i32, _i32, _i32, _i32, _i32, _i32, _TP, _i32, _i32, _i32, _vp, _sz, _i32, _i32, _vp]),
    "vs_bgr_image_warp_roi_batch": (_i32, [_vp, _sz, _i32, _i32, _i32, _i32, _i32, _i32, _TP, _i32, _i32, _i32,
                                           _i32, _i32, _i32, _i32, _vp, _sz, _i32, _i32, _vp]),
    "vs_bgr_image_warp_f32": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _TP, _i32, _i32, _vp, _i32, _i32, _vp]),
    "vs_bgr_to_gray": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "vs_aligner_create": (_vp, [C.POINTER(AlignerParams), _i32]),
    "vs_aligner_destroy": (None, [_vp]),
    "vs_aligner_set_select_mode": (_i32, [_vp, _i32]),
    "vs_stabilizer_set_select_mode": (_i32, [_vp, _i32]),
    "vs_aligner_get_select_mode": (_i32, [_vp]),
    "vs_stabilizer_get_select_mode": (_i32, [_vp]),
    "vs_aligner_set_batch_mode": (_i32, [_vp, _i32]),
    "vs_aligner_reset": (_i32, [_vp]),
    "vs_aligner_align_next": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(AlignerParams), _TP]),
    "vs_aligner_align_batch": (_i32, [_vp, _vp, _sz, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(AlignerParams), _TP, C.POINTER(C.c_int32)]),
    "vs_aligner_align_clips": (_i32, [_vp, _vp, _sz, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(AlignerParams), _TP, C.POINTER(C.c_int32)]),
    "vs_aligner_enable_timing": (_i32, [_vp, _i32]),
    "vs_aligner_get_timings": (_i32, [_vp, C.POINTER(StageTimings)]),
    "vs_aligner_get_info": (_i32, [_vp, _i32, C.POINTER(AlignInfo)]),
    "vs_aligner_level_dims": (_i32, [_vp, _i32, _IP, _IP, _IP, _IP, _IP]),
    "vs_aligner_read_level_image": (_i32, [_vp, _i32, _i32, _vp]),
    "vs_aligner_read_level_argmax": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "vs_aligner_read_level_jacobian": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "vs_stabilizer_create": (_vp, [C.POINTER(StabilizerParams), _i32]),
    "vs_stabilizer_destroy": (None, [_vp]),
    "vs_stabilizer_process": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _IP, _IP]),
    "vs_stabilizer_process_batch": (_i32, [_vp, _vp, _sz, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, C.POINTER(C.c_int32), _IP, _IP]),
    "vs_stabilizer_process_clips": (_i32, [_vp, _vp, _sz, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, C.POINTER(C.c_int32), _IP, _IP]),
    "vs_stabilizer_reset": (_i32, [_vp]),
    "vs_stabilizer_state": (None, [_vp, _TP, _TP, _IP]),
}

_lib = None


def lib():
    """Load libvs_amd.so.  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VsError("libvs_amd.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    # VS_AMD_LIB_PARTIAL=1 (tests/test_host_sanitizers.py): VS_AMD_LIB is a build of the HOST translation unit alone (vs_host.cpp under
    # AddressSanitizer / UBSan, plain g++): only the symbols it has are bound; anything else raises AttributeError where it is used
    partial = os.environ.get("VS_AMD_LIB_PARTIAL") == "1"
    for name, (res, args) in SIGNATURES.items():
        if partial and not hasattr(L, name):
            continue
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.vs_abi_version() != ABI_VERSION or L.vs_sizeof_align_info() != C.sizeof(AlignInfo):
        raise VsError("libvs_amd.so has ABI %d (vs_align_info %d bytes); this binding was written for ABI %d (%d bytes)"
                      % (L.vs_abi_version(), L.vs_sizeof_align_info(), ABI_VERSION, C.sizeof(AlignInfo)))
    _lib = L
    return L


def _check(r):
    if r < 0:
        raise VsError("vs_amd error %d: %s" % (r, lib().vs_last_error().decode()))
    return r


def _p(a):
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def device_count():
    return lib().vs_device_count()


def shader_clock_probe(stream=None):
    """shader clock in MHz as a VALU-bound probe kernel on `stream` sees it (s_memtime / s_memrealtime)"""
    mhz = C.c_double(0.0)
    _check(lib().vs_shader_clock_probe(C.c_void_p(stream) if stream else None, C.byref(mhz)))
    return mhz.value


def debug_bounds_check():
    """bounds build only: violations since the last call (0 = clean); raises VsError in the regular library"""
    r = lib().vs_debug_bounds_check()
    if r < 0:
        raise VsError("vs_amd error %d: %s" % (r, lib().vs_last_error().decode()))
    return r, (lib().vs_last_error().decode() if r else "")


def test_fail_alloc(k):
    """fault injection: the k-th allocation of the library from now on fails once (0 disarms); returns the allocations seen since the last call"""
    return lib().vs_test_fail_alloc(int(k))


def aligner_params(**kw):
    p = AlignerParams()
    lib().vs_aligner_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def stabilizer_params(**kw):
    p = StabilizerParams()
    lib().vs_stabilizer_params_default(C.byref(p))
    for k, v in kw.items():
        if hasattr(p.aligner, k) and not hasattr(p, k):
            setattr(p.aligner, k, v)
        else:
            setattr(p, k, v)
    return p


# ---- host-side algebra ---------------------------------------------------------------------
def t_inverse(t):
    return lib().vs_transform_inverse(C.byref(t))


def t_compose(t1, t2):
    return lib().vs_transform_compose(C.byref(t1), C.byref(t2))


def t_warp(t, x, y, center=None):
    if center is None:
        p = lib().vs_transform_warp(C.byref(t), Point(x, y))
    else:
        p = lib().vs_transform_warp_center(C.byref(t), Point(x, y), center[0], center[1])
    return p.x, p.y


def t_max_corner_displacement(t, w, h):
    return lib().vs_transform_max_corner_displacement(C.byref(t), w, h)


def tile_size(w, h):
    return lib().vs_tile_size(w, h)


def ul_params_sparse(t, w, h):
    out = np.empty(4, np.float32)
    lib().vs_ul_params_sparse(C.byref(t), w, h, _p(out))
    return out


def ul_params_warp(t, w, h):
    out = np.empty(4, np.float32)
    lib().vs_ul_params_warp(C.byref(t), w, h, _p(out))
    return out


def cv_inverse_matrix(t, w, h):
    """VS_WARP_BILINEAR_CV's output -> source matrix for the FORWARD transform t (cv::warpAffine's inversion of imgproc.cpp:457-466's matrix)"""
    out = np.empty(6, np.float64)
    lib().vs_cv_inverse_matrix(C.byref(t), w, h, _p(out))
    return out


def tvl1_smooth(data, lam, iterations=100):
    d = _c(data, np.float64)
    out = np.empty_like(d)
    lib().vs_tvl1_smooth(_p(d), d.size, lam, iterations, _p(out))
    return out


class Smoother:
    """L1SmootherCenter (smoother.hpp:10-30)"""

    def __init__(self, lag_behind, lag_ahead, lam):
        self.h = lib().vs_smoother_create(lag_behind, lag_ahead, lam)

    def update(self, meas):
        out = Transform()
        ok = lib().vs_smoother_update(self.h, C.byref(meas), C.byref(out))
        return bool(ok), out

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.vs_smoother_destroy(self.h)
            self.h = None


# ---- kernel level (host-staged numpy convenience wrappers) ---------------------------------
def pyr_down(img):
    img = _c(img, np.uint8)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), np.uint8)
    _check(lib().vs_pyr_down(_p(img), w, h, w, _p(out), w // 2, h // 2, w // 2, MEM_HOST, None))
    return out


def grad_xy(img):
    img = _c(img, np.uint8)
    h, w = img.shape
    gx = np.empty((h, w), np.float32)
    gy = np.empty((h, w), np.float32)
    _check(lib().vs_grad_xy(_p(img), w, h, w, _p(gx), _p(gy), MEM_HOST, None))
    return gx, gy


def grad_argmax(gx, gy, ts=None):
    gx = _c(gx, np.float32)
    gy = _c(gy, np.float32)
    h, w = gx.shape
    if ts is None:
        ts = tile_size(w, h)
    tx, ty = w // ts, h // ts
    lmx = np.empty((2, ty, tx), np.uint16)
    lmy = np.empty((2, ty, tx), np.uint16)
    _check(lib().vs_grad_argmax(_p(gx), _p(gy), w, h, ts, _p(lmx), _p(lmy), MEM_HOST, None))
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
    _check(lib().vs_sparse_jac(_p(gx), _p(gy), w, h, _p(lmx), _p(lmy), tx, ty, _p(jx), _p(jy), MEM_HOST, None))
    return jx, jy


def keyframe_fused(img, ts=None):
    img = _c(img, np.uint8)
    h, w = img.shape
    if ts is None:
        ts = tile_size(w, h)
    tx, ty = w // ts, h // ts
    lmx = np.empty((2, ty, tx), np.uint16)
    lmy = np.empty((2, ty, tx), np.uint16)
    jx = np.empty((4, ty, tx), np.float32)
    jy = np.empty((4, ty, tx), np.float32)
    _check(lib().vs_keyframe_fused(_p(img), w, h, w, ts, _p(lmx), _p(lmy), _p(jx), _p(jy), MEM_HOST, None))
    return ts, lmx, lmy, jx, jy


def sparse_warpdiff_raw(tmpl, key, lm, A, B, TX, TY):
    tmpl = _c(tmpl, np.uint8)
    key = _c(key, np.uint8)
    lm = _c(lm, np.uint16)
    h, w = key.shape
    _, ty, tx = lm.shape
    out = np.empty((ty, tx), np.uint16)
    _check(lib().vs_sparse_warpdiff(_p(tmpl), _p(key), w, h, w, _p(lm), tx, ty, A, B, TX, TY, _p(out), MEM_HOST, None))
    return out


def sparse_warpdiff(tmpl, key, lm, t):
    """SparseWarpDiff (imgproc.cpp:80-106): t is the centre-based transform"""
    h, w = key.shape
    p = ul_params_sparse(t, w, h)
    return sparse_warpdiff_raw(tmpl, key, lm, float(p[0]), float(p[1]), float(p[2]), float(p[3]))


def sparse_ica_raw(tmpl, key, selx, sely, jacx, jacy, A, B, TX, TY):
    tmpl = _c(tmpl, np.uint8)
    key = _c(key, np.uint8)
    selx = _c(selx, np.uint16)
    sely = _c(sely, np.uint16)
    jacx = _c(jacx, np.float32)
    jacy = _c(jacy, np.float32)
    h, w = key.shape
    out = np.empty(4, np.float64)
    _check(lib().vs_sparse_ica(_p(tmpl), _p(key), w, h, w, _p(selx), selx.shape[1], _p(sely), sely.shape[1],
                               _p(jacx), _p(jacy), A, B, TX, TY, _p(out), MEM_HOST, None))
    return out


def sparse_ica(tmpl, key, selx, sely, jacx, jacy, t):
    """SparseICA (imgproc.cpp:46-78)"""
    h, w = key.shape
    p = ul_params_sparse(t, w, h)
    return sparse_ica_raw(tmpl, key, selx, sely, jacx, jacy, float(p[0]), float(p[1]), float(p[2]), float(p[3]))


def select_smallest(warpdiff, fraction=0.8):
    """warpdiff: (ty,tx) or (n,ty,tx) u16.  returns (list of idx arrays, status array)"""
    wd = _c(warpdiff, np.uint16)
    if wd.ndim == 2:
        wd = wd[None]
    n, ty, tx = wd.shape
    idx = np.empty((n, ty * tx), np.int32)
    status = np.empty(n, np.int32)
    cnt = _check(lib().vs_select_smallest(_p(wd), n, tx, ty, fraction, _p(idx), _p(status), MEM_HOST, None))
    return [idx[i, :cnt].copy() for i in range(n)], status


def select_smallest_stable(warpdiff, fraction=0.8):
    """the same step under VS_SELECT_STABLE's rule.  warpdiff: (ty,tx) or (n,ty,tx) u16.  returns a list of idx arrays"""
    wd = _c(warpdiff, np.uint16)
    if wd.ndim == 2:
        wd = wd[None]
    n, ty, tx = wd.shape
    idx = np.empty((n, ty * tx), np.int32)
    cnt = _check(lib().vs_select_smallest_stable(_p(wd), n, tx, ty, fraction, _p(idx), MEM_HOST, None))
    return [idx[i, :cnt].copy() for i in range(n)]


def optimal_dft_size(n):
    return lib().vs_optimal_dft_size(int(n))


def phase_correlate(a, b, want_surface=False):
    """cv::phaseCorrelate on two u8 images -> (dx, dy, response[, unshifted surface (M, N)])"""
    a = _c(a, np.uint8)
    b = _c(b, np.uint8)
    h, w = a.shape
    res = np.zeros(3, np.float64)
    surf = np.empty((optimal_dft_size(h), optimal_dft_size(w)), np.float32) if want_surface else None
    _check(lib().vs_phase_correlate(_p(a), _p(b), w, h, w, MEM_HOST, None, _p(surf) if want_surface else None, _p(res)))
    return (res[0], res[1], res[2], surf) if want_surface else (res[0], res[1], res[2])


def image_warp_raw(img, A, B, TX, TY, out_shape=None):
    img = _c(img, np.uint8)
    h, w = img.shape
    oh, ow = out_shape if out_shape else (h, w)
    out = np.empty((oh, ow), np.float32)
    _check(lib().vs_image_warp(_p(img), w, h, w, A, B, TX, TY, _p(out), ow, oh, MEM_HOST, None))
    return out


def image_warp(img, t):
    """ImageWarp (imgproc.cpp:116-133)"""
    h, w = img.shape
    p = ul_params_warp(t, w, h)
    return image_warp_raw(img, float(p[0]), float(p[1]), float(p[2]), float(p[3]))


def bgr_image_warp(src, t, mode=WARP_LANCZOS2, border=BORDER_CLAMP, max_value=None, f32=False):
    src = np.ascontiguousarray(src)
    assert src.dtype in (np.uint8, np.uint16) and src.ndim == 3
    h, w, c = src.shape
    bits = 8 if src.dtype == np.uint8 else 16
    if f32:
        out = np.empty((h, w, c), np.float32)
        _check(lib().vs_bgr_image_warp_f32(_p(src), w, h, w * c, c, bits, C.byref(t), mode, border, _p(out), w * c, MEM_HOST, None))
        return out
    if max_value is None:
        max_value = 255 if bits == 8 else 65535
    out = np.empty_like(src)
    _check(lib().vs_bgr_image_warp(_p(src), w, h, w * c, c, bits, C.byref(t), mode, border, max_value, _p(out), w * c, MEM_HOST, None))
    return out


def bgr_image_warp_batch(src, ts, mode=WARP_LANCZOS2, border=BORDER_CLAMP, max_value=None):
    """src (n,h,w,c) numpy; ts list of Transform"""
    src = np.ascontiguousarray(src)
    n, h, w, c = src.shape
    bits = 8 if src.dtype == np.uint8 else 16
    if max_value is None:
        max_value = 255 if bits == 8 else 65535
    arr = (Transform * n)(*ts)
    out = np.empty_like(src)
    _check(lib().vs_bgr_image_warp_batch(_p(src), h * w * c, n, w, h, w * c, c, bits, arr, mode, border, max_value,
                                         _p(out), h * w * c, w * c, MEM_HOST, None))
    return out


def bgr_image_warp_roi_batch(src, ts, roi, mode=WARP_LANCZOS2, border=BORDER_CLAMP, max_value=None):
    """only the window roi = (x, y, w, h) of every warped frame: (n, roi_h, roi_w, c)"""
    src = np.ascontiguousarray(src)
    n, h, w, c = src.shape
    bits = 8 if src.dtype == np.uint8 else 16
    if max_value is None:
        max_value = 255 if bits == 8 else 65535
    rx, ry, rw, rh = roi
    arr = (Transform * n)(*ts)
    out = np.empty((n, rh, rw, c), src.dtype)
    _check(lib().vs_bgr_image_warp_roi_batch(_p(src), h * w * c, n, w, h, w * c, c, bits, arr, mode, border, max_value,
                                             rx, ry, rw, rh, _p(out), rh * rw * c, rw * c, MEM_HOST, None))
    return out


def bgr_image_warp_batch_device(src_ptr, n, w, h, c, bits, ts, dst_ptr, mode=WARP_LANCZOS2, border=BORDER_CLAMP,
                                max_value=None, stream=None):
    """device-resident form: dense frames, enqueue only"""
    if max_value is None:
        max_value = 255 if bits == 8 else 65535
    arr = ts if isinstance(ts, C.Array) else (Transform * n)(*ts)     # a ctypes array from align_* is passed through as is
    _check(lib().vs_bgr_image_warp_batch(_p(src_ptr), h * w * c, n, w, h, w * c, c, bits, arr, mode, border, max_value,
                                         _p(dst_ptr), h * w * c, w * c, MEM_DEVICE, stream))


def bgr_to_gray(src, shift_to_8=None):
    src = np.ascontiguousarray(src)
    h, w, _ = src.shape
    bits = 8 if src.dtype == np.uint8 else 16
    if shift_to_8 is None:
        shift_to_8 = 0 if bits == 8 else 2
    out = np.empty((h, w), np.uint8)
    _check(lib().vs_bgr_to_gray(_p(src), w, h, w * 3, bits, shift_to_8, _p(out), w, MEM_HOST, None))
    return out


# ---- engine level ---------------------------------------------------------------------------
def _fmt_of(frame_dtype, ndim_tail):
    if ndim_tail == 2:
        return FMT_GRAY8
    return FMT_BGR8 if frame_dtype == np.uint8 else FMT_BGR10      # u16 numpy frames are 10-bit unless the caller passes fmt=


class Aligner:
    """VideoAligner (alignment.hpp:51-99) on the GPU engine."""

    def __init__(self, device=0, select_mode=SELECT_DEVICE, **params):
        self.params = aligner_params(**params)
        self.h = lib().vs_aligner_create(C.byref(self.params), device)
        if not self.h:
            raise VsError("vs_aligner_create failed: %s" % lib().vs_last_error().decode())
        _check(lib().vs_aligner_set_select_mode(self.h, select_mode))

    def set_select_mode(self, mode):
        """SELECT_DEVICE (default: libstdc++'s nth_element order, replicated on the device), SELECT_STABLE (the documented
        STL-independent rule: smallest by (abs_delta, tile index), survivors in tile order) or SELECT_STL_HOST"""
        _check(lib().vs_aligner_set_select_mode(self.h, mode))

    def select_mode(self):
        """the selection mode in force (the VS_SELECT_MODE environment override included)"""
        return _check(lib().vs_aligner_get_select_mode(self.h))

    def set_batch_mode(self, mode):
        """BATCH_SHARED: full batches run through the small-footprint solver kernel (same bits; for callers that overlap
        other GPU work, e.g. the previous clip's warp)"""
        _check(lib().vs_aligner_set_batch_mode(self.h, mode))

    def align_next(self, frame, fmt=None):
        """frame: numpy (h,w) u8 gray, (h,w,3) u8/u16 BGR (u16 = 10-bit unless fmt says FMT_BGR12 / FMT_BGR16_FULL).
        returns (ok, Transform)"""
        frame = np.ascontiguousarray(frame)
        fmt = _fmt_of(frame.dtype, frame.ndim) if fmt is None else fmt
        hh, ww = frame.shape[:2]
        stride = ww * (1 if fmt == FMT_GRAY8 else 3)
        t = Transform()
        r = _check(lib().vs_aligner_align_next(self.h, _p(frame), ww, hh, stride, fmt, MEM_HOST, C.byref(self.params), C.byref(t)))
        return bool(r), t

    def align_batch(self, frames, fmt=None):
        """frames: numpy (n,h,w[,3]).  returns (status[n], [Transform]*n)"""
        frames = np.ascontiguousarray(frames)
        n = frames.shape[0]
        fmt = _fmt_of(frames.dtype, frames.ndim - 1) if fmt is None else fmt
        hh, ww = frames.shape[1:3]
        ch = 1 if fmt == FMT_GRAY8 else 3
        out = (Transform * n)()
        status = (C.c_int32 * n)()
        _check(lib().vs_aligner_align_batch(self.h, _p(frames), hh * ww * ch, n, ww, hh, ww * ch, fmt, MEM_HOST,
                                            C.byref(self.params), out, status))
        return list(status), list(out)

    def align_batch_device(self, ptr, n, w, h, fmt, stride=None, frame_stride=None):
        """device-resident frames (raw device pointer)"""
        ch = 1 if fmt == FMT_GRAY8 else 3
        stride = stride or w * ch
        frame_stride = frame_stride or h * stride
        out = (Transform * n)()
        status = (C.c_int32 * n)()
        _check(lib().vs_aligner_align_batch(self.h, _p(ptr), frame_stride, n, w, h, stride, fmt, MEM_DEVICE,
                                            C.byref(self.params), out, status))
        return list(status), list(out)

    def align_clips(self, frames, n_clips, mem_ptr=None, w=None, h=None, fmt=None, raw=False):
        """frames: numpy (n_clips*fpc, h, w[,3]) (host) -- or pass mem_ptr/w/h/fmt with frames = total frame count
        for dense device-resident clips.  returns (status list, transforms list)"""
        if mem_ptr is None:
            frames = np.ascontiguousarray(frames)
            n = frames.shape[0]
            fmt = _fmt_of(frames.dtype, frames.ndim - 1) if fmt is None else fmt
            h, w = frames.shape[1:3]
            ptr, mem = _p(frames), MEM_HOST
        else:
            n, ptr, mem = int(frames), _p(mem_ptr), MEM_DEVICE
        ch = 1 if fmt == FMT_GRAY8 else 3
        out = (Transform * n)()
        status = (C.c_int32 * n)()
        _check(lib().vs_aligner_align_clips(self.h, ptr, h * w * ch, n_clips, n // n_clips, w, h, w * ch, fmt, mem,
                                            C.byref(self.params), out, status))
        if raw:
            return status, out       # ctypes arrays: no per-frame Python objects (bench.py hands `out` to the warp call)
        return list(status), list(out)

    def reset(self):
        _check(lib().vs_aligner_reset(self.h))

    def enable_timing(self, on=True):
        _check(lib().vs_aligner_enable_timing(self.h, 1 if on else 0))

    def timings(self):
        t = StageTimings()
        _check(lib().vs_aligner_get_timings(self.h, C.byref(t)))
        d = {name: {"ms": t.ms[i], "launches": t.launches[i]} for i, name in enumerate(STAGES)}
        d["frames"] = t.frames
        d["gn_iterations"] = t.gn_iterations
        return d

    def wait_stream(self, stream_handle):
        """order the handle's stream behind everything enqueued so far on the caller's stream (raw hipStream_t value)"""
        _check(lib().vs_aligner_wait_stream(self.h, C.c_void_p(stream_handle)))

    def info(self, i=0):
        inf = AlignInfo()
        _check(lib().vs_aligner_get_info(self.h, i, C.byref(inf)))
        return inf

    def level(self, i, level):
        w, h, tx, ty, ts = (C.c_int() for _ in range(5))
        _check(lib().vs_aligner_level_dims(self.h, level, *(C.byref(v) for v in (w, h, tx, ty, ts))))
        w, h, tx, ty, ts = (v.value for v in (w, h, tx, ty, ts))
        d = {"w": w, "h": h, "tx": tx, "ty": ty, "ts": ts}
        img = np.empty((h, w), np.uint8)
        _check(lib().vs_aligner_read_level_image(self.h, i, level, _p(img)))
        d["img"] = img
        return d

    def keyframe_tables(self, i, level):
        w, h, tx, ty, ts = (C.c_int() for _ in range(5))
        _check(lib().vs_aligner_level_dims(self.h, level, *(C.byref(v) for v in (w, h, tx, ty, ts))))
        tx, ty = tx.value, ty.value
        out = {"argmax": [], "jac": []}
        for s in (0, 1):
            am = np.empty((2, ty, tx), np.uint16)
            jc = np.empty((4, ty, tx), np.float32)
            _check(lib().vs_aligner_read_level_argmax(self.h, i, level, s, _p(am)))
            _check(lib().vs_aligner_read_level_jacobian(self.h, i, level, s, _p(jc)))
            out["argmax"].append(am)
            out["jac"].append(jc)
        return out

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.vs_aligner_destroy(self.h)
            self.h = None


class Stabilizer:
    """VideoStabilizer (stabilizer.hpp:32-56) on the GPU engine."""

    def __init__(self, device=0, select_mode=None, **params):
        self.params = stabilizer_params(**params)
        self.h = lib().vs_stabilizer_create(C.byref(self.params), device)
        if not self.h:
            raise VsError("vs_stabilizer_create failed: %s" % lib().vs_last_error().decode())
        if select_mode is not None:
            self.set_select_mode(select_mode)

    def set_select_mode(self, mode):
        """the selection rule of the stabilizer's aligner (SELECT_DEVICE by default, SELECT_STABLE, SELECT_STL_HOST)"""
        _check(lib().vs_stabilizer_set_select_mode(self.h, mode))

    def select_mode(self):
        return _check(lib().vs_stabilizer_get_select_mode(self.h))

    def process(self, frame, fmt=None):
        frame = np.ascontiguousarray(frame)
        fmt = _fmt_of(frame.dtype, frame.ndim) if fmt is None else fmt
        hh, ww = frame.shape[:2]
        c = max(self.params.crop_pixels, 0)
        out = np.empty((hh - 2 * c, ww - 2 * c, 3), frame.dtype)
        ow, oh = C.c_int(), C.c_int()
        r = _check(lib().vs_stabilizer_process(self.h, _p(frame), ww, hh, ww * 3, fmt, MEM_HOST, _p(out), C.byref(ow), C.byref(oh)))
        return out if r == 1 else None

    def reset(self):
        _check(lib().vs_stabilizer_reset(self.h))

    def wait_stream(self, stream_handle):
        _check(lib().vs_stabilizer_wait_stream(self.h, C.c_void_p(stream_handle)))

    def process_batch(self, frames, out=None, fmt=None):
        """frames (n,h,w,3) numpy.  returns (outputs (n,oh,ow,3), has_output list).  out: a buffer of that shape to reuse
        (frames without an output are left untouched in it)"""
        frames = np.ascontiguousarray(frames)
        n, hh, ww = frames.shape[:3]
        fmt = _fmt_of(frames.dtype, 3) if fmt is None else fmt
        c = max(self.params.crop_pixels, 0)
        if out is None:
            out = np.zeros((n, hh - 2 * c, ww - 2 * c, 3), frames.dtype)
        assert out.shape == (n, hh - 2 * c, ww - 2 * c, 3) and out.dtype == frames.dtype and out.flags.c_contiguous
        has = (C.c_int32 * n)()
        ow, oh = C.c_int(), C.c_int()
        _check(lib().vs_stabilizer_process_batch(self.h, _p(frames), hh * ww * 3, n, ww, hh, ww * 3, fmt, MEM_HOST, _p(out),
                                                 out[0].size, has, C.byref(ow), C.byref(oh)))
        return out, list(has)

    def process_batch_device(self, ptr, n, w, h, fmt, out_ptr):
        """dense device-resident frames in, dense cropped frames out (raw device pointers)"""
        c = max(self.params.crop_pixels, 0)
        has = (C.c_int32 * n)()
        ow, oh = C.c_int(), C.c_int()
        r = _check(lib().vs_stabilizer_process_batch(self.h, _p(ptr), h * w * 3, n, w, h, w * 3, fmt, MEM_DEVICE, _p(out_ptr),
                                                     (h - 2 * c) * (w - 2 * c) * 3, has, C.byref(ow), C.byref(oh)))
        return r, list(has)

    def process_clips(self, frames, n_clips, fmt=None):
        """frames (n_clips*fpc, h, w, 3) numpy: every clip through a fresh stabilizer, batched together"""
        frames = np.ascontiguousarray(frames)
        n, hh, ww = frames.shape[:3]
        fmt = _fmt_of(frames.dtype, 3) if fmt is None else fmt
        c = max(self.params.crop_pixels, 0)
        out = np.zeros((n, hh - 2 * c, ww - 2 * c, 3), frames.dtype)
        has = (C.c_int32 * n)()
        ow, oh = C.c_int(), C.c_int()
        _check(lib().vs_stabilizer_process_clips(self.h, _p(frames), hh * ww * 3, n_clips, n // n_clips, ww, hh, ww * 3, fmt, MEM_HOST,
                                                 _p(out), out[0].size, has, C.byref(ow), C.byref(oh)))
        return out, list(has)

    def process_clips_device(self, ptr, n_clips, frames_per_clip, w, h, fmt, out_ptr):
        c = max(self.params.crop_pixels, 0)
        n = n_clips * frames_per_clip
        has = (C.c_int32 * n)()
        ow, oh = C.c_int(), C.c_int()
        r = _check(lib().vs_stabilizer_process_clips(self.h, _p(ptr), h * w * 3, n_clips, frames_per_clip, w, h, w * 3, fmt, MEM_DEVICE,
                                                     _p(out_ptr), (h - 2 * c) * (w - 2 * c) * 3, has, C.byref(ow), C.byref(oh)))
        return r, list(has)

    def state(self):
        m, a, s = Transform(), Transform(), C.c_int()
        lib().vs_stabilizer_state(self.h, C.byref(m), C.byref(a), C.byref(s))
        return m, a, bool(s.value)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.vs_stabilizer_destroy(self.h)
            self.h = None
