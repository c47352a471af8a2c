"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/hk_oracle.c (built into oracle/_build/libhk_oracle.so by
``python -m homonim_amd.build``).  Same call shape as oracle_np so tests can swap one for the other.
"""
import ctypes as C
import math
import os

import numpy as np

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build', 'libhk_oracle.so')
_lib = None
_MODEL = {'gain': 0, 'gain-blk-offset': 1, 'gain-offset': 2}
_f32p, _f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)


def available() -> bool:
    return os.path.exists(_LIB)


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(_LIB)
        lib.hk_oracle_fit_apply.restype = C.c_int
        lib.hk_oracle_fit_apply.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f32p, C.c_int, C.c_float, _f32p, C.c_int,
            C.c_float, C.c_int, C.c_int, _f64p, _f32p, C.c_int, _f32p, C.POINTER(C.c_uint64), C.c_int
        ]
        lib.hk_oracle_block_norm.restype = C.c_int
        lib.hk_oracle_block_norm.argtypes = [_f32p, C.c_int, C.c_float, _f32p, C.c_int, C.c_float, C.c_int, C.c_int, _f64p]
        lib.hk_oracle_apply.restype = None
        lib.hk_oracle_apply.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _f32p]
        lib.hk_oracle_fill_nodata.restype = C.c_int
        lib.hk_oracle_fill_nodata.argtypes = [_f32p, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_double]
        lib.hk_oracle_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def _nd(nodata):
    if nodata is None:
        return 0, 0.0
    if math.isnan(float(nodata)):
        return 1, float('nan')
    return 2, float(nodata)


def max_threads() -> int:
    return int(_load().hk_oracle_max_threads())


def fit_block_norm(src, src_nodata, ref, ref_nodata) -> np.ndarray:
    """ float64 flavour of kernel_model.py:216-229 (see hk_oracle.c). """
    src = np.ascontiguousarray(src, np.float32)
    ref = np.ascontiguousarray(ref, np.float32)
    norm = np.zeros(2, np.float64)
    sm, sv = _nd(src_nodata)
    rm, rv = _nd(ref_nodata)
    rc = _load().hk_oracle_block_norm(src.ctypes.data_as(_f32p), sm, sv, ref.ctypes.data_as(_f32p), rm, rv,
                                      src.shape[0], src.shape[1], norm.ctypes.data_as(_f64p))
    assert rc == 0
    return norm


def fill_nodata(image: np.ndarray, mask: np.ndarray, max_search_distance: float = 100.0) -> np.ndarray:
    """ C twin of oracle_np.fill_nodata (restated GDALFillNodata; parity with GDAL unpinned). """
    out = np.array(image, dtype=np.float32, copy=True, order='C')
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    rc = _load().hk_oracle_fill_nodata(out.ctypes.data_as(_f32p), m.ctypes.data_as(C.POINTER(C.c_ubyte)), out.shape[0],
                                       out.shape[1], float(max_search_distance))
    assert rc == 0
    return out


def fit_apply(model, src, src_nodata, ref, ref_nodata, kernel_shape=(5, 5), find_r2=False, r2_inpaint_thresh=0.25,
              norm_model=None, want_params=True, want_corr=True, n_threads=0):
    """ -> (params | None, corr | None, n_fail).  gain-blk-offset needs ``norm_model`` (float64[2]).  gain-offset with
    a threshold runs the whole reference branch incl. in-painting when n_fail > 0 (kernel_model.py:361-371). """
    src = np.ascontiguousarray(src, np.float32)
    ref = np.ascontiguousarray(ref, np.float32)
    h, w = src.shape
    thresh = r2_inpaint_thresh if model == 'gain-offset' else None
    with_r2 = bool(find_r2 or thresh is not None)
    nb = 3 if with_r2 else 2
    params = np.empty((nb, h, w), np.float32) if want_params else None
    corr = np.empty((h, w), np.float32) if want_corr else None
    fail = C.c_uint64(0)
    norm = None if norm_model is None else np.ascontiguousarray(norm_model, np.float64)
    sm, sv = _nd(src_nodata)
    rm, rv = _nd(ref_nodata)
    rc = _load().hk_oracle_fit_apply(
        _MODEL[model], int(kernel_shape[0]), int(kernel_shape[1]), int(bool(find_r2)), int(thresh is not None),
        float(thresh) if thresh is not None else 0.0, src.ctypes.data_as(_f32p), sm, sv, ref.ctypes.data_as(_f32p), rm,
        rv, h, w, norm.ctypes.data_as(_f64p) if norm is not None else None,
        params.ctypes.data_as(_f32p) if want_params else None, nb, corr.ctypes.data_as(_f32p) if want_corr else None,
        C.byref(fail), int(n_threads)
    )
    if rc != 0:
        raise ValueError('hk_oracle_fit_apply: bad arguments')
    return params, corr, int(fail.value)
