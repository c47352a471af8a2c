"""
ctypes binding of include/homonim_hk.h (libhomonim_hk.so) -- the thin FFI layer between the Python host code and the
hand-written HIP kernels.  This is the stub a homonim maintainer would add to homonim/kernel_model.py.

There is NO CPU fallback: if the library is missing or no gfx950 device is present, every entry point raises
``DeviceError`` loudly.
"""
import atexit
import ctypes as C
import sys
import weakref
import math
import os
import threading
from typing import Optional

import numpy as np

from homonim_amd.errors import DeviceError

# HOMONIM_AMD_LIB overrides the library path (kernel-variant experiments); the default is the in-tree build.
_LIB_PATH = os.environ.get('HOMONIM_AMD_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib',
                                                               'libhomonim_hk.so')

HK_OK, HK_ERR_ARG, HK_ERR_HIP, HK_ERR_NODEVICE, HK_ERR_UNSUPPORTED, HK_ERR_NOMEM, HK_ERR_ALREADY = 0, -1, -2, -3, -4, -5, -6
MODEL_CODES = {'gain': 0, 'gain-blk-offset': 1, 'gain-offset': 2}
NODATA_NONE, NODATA_NAN, NODATA_VALUE = 0, 1, 2


class FitDesc(C.Structure):
    _fields_ = [
        ('model', C.c_int32), ('kh', C.c_int32), ('kw', C.c_int32), ('find_r2', C.c_int32),
        ('has_r2_thresh', C.c_int32), ('r2_thresh', C.c_float), ('src_nodata_mode', C.c_int32),
        ('src_nodata', C.c_float), ('ref_nodata_mode', C.c_int32), ('ref_nodata', C.c_float),
    ]  # yapf: disable


class IoDesc(C.Structure):
    _fields_ = [('src_dtype', C.c_int32), ('ref_dtype', C.c_int32), ('out_dtype', C.c_int32),
                ('out_has_nodata', C.c_int32), ('out_nodata', C.c_double)]


class SpaceDesc(C.Structure):
    _fields_ = [('down', C.c_double * 4), ('up', C.c_double * 4), ('down_resampling', C.c_int32),
                ('up_resampling', C.c_int32), ('mask_partial', C.c_int32)]


# numpy dtype name -> hk_dtype
DTYPE_CODES = {'float32': 0, 'uint8': 1, 'uint16': 2, 'int16': 3, 'uint32': 4, 'int32': 5, 'float64': 6}


class OutWindow(C.Structure):
    _fields_ = [('stride', C.c_int64), ('band_stride', C.c_int64), ('row0', C.c_int32), ('col0', C.c_int32),
                ('rows', C.c_int32), ('cols', C.c_int32), ('param_stride', C.c_int64)]


class DevJob(C.Structure):
    _fields_ = [
        ('src', C.c_void_p), ('ref', C.c_void_p), ('gain', C.c_void_p), ('offset', C.c_void_p), ('r2', C.c_void_p),
        ('corr', C.c_void_p), ('norm', C.c_void_p), ('fail_count', C.c_void_p), ('n_bands', C.c_int32),
        ('height', C.c_int32), ('width', C.c_int32), ('stride', C.c_int64), ('band_stride', C.c_int64),
        ('seg_rows', C.c_int32), ('stream', C.c_int32),
        ('out_row0', C.c_int32), ('out_col0', C.c_int32), ('out_rows', C.c_int32), ('out_cols', C.c_int32),
        ('scratch', C.c_void_p), ('scratch_bytes', C.c_uint64),
    ]  # yapf: disable


_P = C.POINTER
_f32p, _f64p, _u64p = _P(C.c_float), _P(C.c_double), _P(C.c_uint64)

# name -> (restype, argtypes); kept in one table so tests can check every symbol of the header is exported
ABI_VERSION = 6   # HK_ABI_VERSION of the include/homonim_hk.h these mirrors were written against
# entry points declared in include/homonim_hk_devtools.h (measurement / test aids), the rest in include/homonim_hk.h
DEVTOOLS = ('hk_synth_fill_dev', 'hk_stream_probe_dev', 'hk_debug_stage_stamps', 'hk_r2_certificate_constants', 'hk_debug_staging_counters',
            'hk_debug_build_ledger', 'hk_debug_checksum_dev', 'hk_debug_fail_after_d2h')

SIGNATURES = {
    'hk_abi_version': (C.c_int, []),
    'hk_backend_name': (C.c_char_p, []),
    'hk_dev_job_scratch_bytes': (C.c_uint64, [C.c_int32, C.c_int32, C.c_int64, C.c_int64]),
    'hk_block_norm_split_exchange_doubles': (C.c_uint64, [C.c_int32]),
    'hk_block_norm_split_dev': (C.c_int, [C.c_void_p, _P(FitDesc), _P(DevJob), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'hk_comm_unique_id': (C.c_int, [C.c_void_p]),
    'hk_comm_init': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    'hk_comm_destroy': (C.c_int, [C.c_void_p]),
    'hk_comm_info': (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32)]),
    'hk_comm_allreduce_f64_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]),
    'hk_block_norm_split_comm_dev': (C.c_int, [C.c_void_p, _P(FitDesc), _P(DevJob), C.c_void_p]),
    'hk_counts_pending': (C.c_int, [_P(C.c_uint64), C.c_int32]),
    'hk_last_error': (C.c_char_p, []),
    'hk_device_count': (C.c_int, [_P(C.c_int)]),
    'hk_device_pci_bus_id': (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    'hk_ctx_create': (C.c_int, [C.c_int, C.c_int, _P(C.c_void_p)]),
    'hk_ctx_destroy': (C.c_int, [C.c_void_p]),
    'hk_ctx_sync': (C.c_int, [C.c_void_p]),
    'hk_block_norm': (C.c_int, [C.c_void_p, _P(FitDesc), _f32p, C.c_int64, _f32p, C.c_int64, C.c_int32, C.c_int32, _f64p]),
    'hk_compare_sums': (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int32, C.c_float, _f32p, C.c_int64, C.c_int32, C.c_float,
                                  C.c_int32, C.c_int32, _f64p]),
    'hk_fit': (C.c_int, [C.c_void_p, _P(FitDesc), _f32p, C.c_int64, _f32p, C.c_int64, C.c_int32, C.c_int32, _f64p,
                         _f32p, C.c_int32, _f64p, _u64p]),
    'hk_apply': (C.c_int, [C.c_void_p, _f32p, C.c_int64, _f32p, C.c_int32, C.c_int32, _f32p]),
    'hk_fit_apply': (C.c_int, [C.c_void_p, _P(FitDesc), _f32p, C.c_int64, _f32p, C.c_int64, C.c_int32, C.c_int32,
                               _f64p, _f32p, C.c_int32, _f32p, _f64p, _u64p]),
    'hk_refspace_fit_apply': (C.c_int, [C.c_void_p, _P(FitDesc), _P(IoDesc), _P(SpaceDesc), C.c_void_p, C.c_int64, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, _f32p, C.c_int32,
                                        C.c_void_p, _u64p]),
    'hk_reproject': (C.c_int, [C.c_void_p, _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_double, C.c_double,
                               C.c_double, C.c_double, C.c_int32, _f32p, C.c_int32, C.c_int32, C.c_float]),
    'hk_partial_mask': (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int32, C.c_float, _f32p, C.c_int32, _f32p, C.c_int64,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p, _f32p, _P(C.c_uint8)]),
    'hk_fit_apply_io': (C.c_int, [C.c_void_p, _P(FitDesc), _P(IoDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                  C.c_int32, C.c_int32, _f64p, _f32p, C.c_int32, C.c_void_p, _f64p, _u64p]),
    'hk_fit_apply_block': (C.c_int, [C.c_void_p, _P(FitDesc), _P(IoDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                     C.c_int32, C.c_int32, _f64p, C.c_void_p, C.c_int32, C.c_void_p, _P(OutWindow), _f64p, _u64p]),
    'hk_host_alloc': (C.c_int, [C.c_void_p, C.c_size_t, _P(C.c_void_p)]),
    'hk_host_free': (C.c_int, [C.c_void_p, C.c_void_p]),
    'hk_host_register': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'hk_host_unregister': (C.c_int, [C.c_void_p, C.c_void_p]),
    'hk_dev_alloc': (C.c_int, [C.c_void_p, C.c_size_t, _P(C.c_void_p)]),
    'hk_dev_free': (C.c_int, [C.c_void_p, C.c_void_p]),
    'hk_memcpy_h2d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'hk_memcpy_d2h': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    'hk_memset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    'hk_fit_apply_dev': (C.c_int, [C.c_void_p, _P(FitDesc), _P(DevJob)]),
    'hk_inpaint_dev': (C.c_int, [C.c_void_p, _P(FitDesc), _P(DevJob), _P(C.c_uint64)]),
    'hk_fail_counts_async': (C.c_int, [C.c_void_p, _P(DevJob), _P(C.c_uint64), C.c_void_p]),
    'hk_inpaint_dev_counts': (C.c_int, [C.c_void_p, _P(FitDesc), _P(DevJob), _P(C.c_uint64), _P(C.c_uint64)]),
    'hk_event_sync': (C.c_int, [C.c_void_p, C.c_void_p]),
    'hk_r2_certificate_constants': (C.c_int, [C.c_float, _P(C.c_double), _P(C.c_double), _P(C.c_float), _P(C.c_float)]),
    'hk_block_norm_dev': (C.c_int, [C.c_void_p, _P(FitDesc), _P(DevJob), C.c_void_p]),
    'hk_block_norm_batch_dev': (C.c_int, [C.c_void_p, _P(FitDesc), _P(DevJob), C.c_int32, C.c_void_p]),
    'hk_fit_apply_batch_dev': (C.c_int, [C.c_void_p, _P(FitDesc), _P(DevJob), C.c_int32]),
    'hk_fail_counts_batch_async': (C.c_int, [C.c_void_p, _P(DevJob), C.c_int32, _P(C.c_uint64), C.c_void_p]),
    'hk_compare_sums_dev': (C.c_int, [C.c_void_p, _P(DevJob), C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_void_p]),
    'hk_synth_fill_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                    C.c_int64, C.c_uint64, C.c_int32, C.c_int32]),
    'hk_stream_probe_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32]),
    'hk_event_create': (C.c_int, [C.c_void_p, _P(C.c_void_p)]),
    'hk_event_destroy': (C.c_int, [C.c_void_p, C.c_void_p]),
    'hk_event_record': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'hk_stream_wait_event': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    'hk_event_elapsed_ms': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_float)]),
    'hk_stream_sync': (C.c_int, [C.c_void_p, C.c_int32]),
    'hk_selftest': (C.c_int, [C.c_void_p]),
    'hk_debug_stage_stamps': (C.c_int, [C.c_void_p, _P(C.c_uint64), C.c_int32]),
    'hk_debug_staging_counters': (C.c_int, [_P(C.c_uint64), C.c_int32]),
    'hk_debug_build_ledger': (C.c_int, [C.c_char_p, C.c_size_t, _P(C.c_size_t), C.c_int32]),
    'hk_debug_fail_after_d2h': (C.c_int, [C.c_int32]),
    'hk_debug_checksum_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P(C.c_uint64)]),
}  # yapf: disable

COMM_ID_BYTES = 128   # HK_COMM_ID_BYTES = sizeof(ncclUniqueId)


def r2_certificate_constants(thresh: float):
    """ (pass_below, fail_above, kappa, kappa_fail) of the r2-mask decision for `thresh` (hk_r2_certificate_constants; host-only). """
    pb, fa, k, kf = C.c_double(), C.c_double(), C.c_float(), C.c_float()
    _check(load_library().hk_r2_certificate_constants(C.c_float(thresh), C.byref(pb), C.byref(fa), C.byref(k), C.byref(kf)))
    return pb.value, fa.value, k.value, kf.value


def comm_unique_id() -> bytes:
    """ A fresh communicator id (hk_comm_unique_id = ncclGetUniqueId): rank 0 calls this and distributes the bytes. """
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _check(load_library().hk_comm_unique_id(buf))
    return bytes(buf)


_lib = None
_lib_lock = threading.Lock()
# page-lock registrations made through Context.pin: (address, bytes) -> [users, registered by us]
_pins = {}
_pin_lock = threading.Lock()


def lib_path() -> str:
    return _LIB_PATH


def load_library():
    """ dlopen libhomonim_hk.so and declare every prototype.  Raises DeviceError if it has not been built. """
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise DeviceError(
                    f'{_LIB_PATH} not found: build it with `python -m homonim_amd.build` (needs hipcc). '
                    'homonim_amd has no CPU fallback.'
                )
            lib = C.CDLL(_LIB_PATH)
            try:
                lib.hk_abi_version.restype = C.c_int
                found = lib.hk_abi_version()
            except AttributeError:
                found = None
            if found != ABI_VERSION:   # struct layouts differ between versions and carry no size field: do not call further
                raise DeviceError(f'{_LIB_PATH} has ABI version {found}, this binding needs {ABI_VERSION}: rebuild it with '
                                  '`python -m homonim_amd.build`')
            for name, (restype, argtypes) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


def _check(rc: int):
    if rc == HK_OK:
        return
    msg = load_library().hk_last_error().decode('utf-8', 'replace')
    if rc == HK_ERR_ARG:
        raise ValueError(msg)
    raise DeviceError(f'libhomonim_hk error {rc}: {msg}')


def nodata_code(nodata):
    """ RasterArray.nodata -> (mode, value) of the C ABI. """
    if nodata is None:
        return NODATA_NONE, 0.0
    nodata = float(nodata)
    if math.isnan(nodata):
        return NODATA_NAN, float('nan')
    return NODATA_VALUE, nodata


def make_desc(model: str, kernel_shape, find_r2: bool, r2_inpaint_thresh: Optional[float], src_nodata, ref_nodata):
    d = FitDesc()
    d.model = MODEL_CODES[str(getattr(model, 'value', model))]
    d.kh, d.kw = int(kernel_shape[0]), int(kernel_shape[1])
    d.find_r2 = int(bool(find_r2))
    d.has_r2_thresh = int(r2_inpaint_thresh is not None)
    d.r2_thresh = float(r2_inpaint_thresh) if r2_inpaint_thresh is not None else 0.0
    d.src_nodata_mode, d.src_nodata = nodata_code(src_nodata)
    d.ref_nodata_mode, d.ref_nodata = nodata_code(ref_nodata)
    return d


def _as_f32_2d(a: np.ndarray, name: str) -> np.ndarray:
    if a.ndim != 2:
        raise ValueError(f'`{name}` must be 2-D')
    if a.dtype != np.float32 or a.strides[1] != 4 or a.strides[0] % 4 != 0 or a.strides[0] < a.shape[1] * 4:
        a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _as_2d_native(a: np.ndarray, name: str) -> np.ndarray:
    """ 2-D raster in one of the dtypes the device converts itself (DTYPE_CODES); rows may be strided. """
    if a.ndim != 2:
        raise ValueError(f'`{name}` must be 2-D')
    if a.dtype.name not in DTYPE_CODES:
        a = a.astype(np.float32)
    it = a.dtype.itemsize
    if a.strides[1] != it or a.strides[0] % it != 0 or a.strides[0] < a.shape[1] * it:
        a = np.ascontiguousarray(a)
    return a


def _ptr(a: np.ndarray, typ=_f32p):
    return a.ctypes.data_as(typ)


_live_contexts = weakref.WeakSet()


@atexit.register
def _close_live_contexts():
    """ Contexts still open when the interpreter exits are destroyed HERE -- while the library, the HIP runtime and this module
    are all intact -- instead of from garbage collection during shutdown. """
    for c in list(_live_contexts):
        try:
            c.close()
        except Exception:
            pass


class Context:
    """ One GPU context (hk_ctx): a device + a pool of streams.  Thread-safe; share it between worker threads. """

    def __init__(self, device: int = 0, n_streams: int = 4):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.hk_ctx_create(int(device), int(n_streams), C.byref(h))
        if rc != HK_OK:
            raise DeviceError(
                f"cannot create a GPU context on device {device}: {self._lib.hk_last_error().decode('utf-8', 'replace')}"
            )
        self._h = h
        self.device = device
        self.n_streams = n_streams
        _live_contexts.add(self)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.hk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        # never during interpreter shutdown: by then the order in which the runtime's own objects die is not ours to rely on (a
        # context that is still open is closed by the atexit hook below, before any teardown starts)
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def selftest(self):
        _check(self._lib.hk_selftest(self._h))

    def sync(self):
        _check(self._lib.hk_ctx_sync(self._h))

    # -- host-pointer calls (numpy in / numpy out) --------------------------------------------------------------------
    def block_norm(self, desc: FitDesc, src: np.ndarray, ref: np.ndarray) -> np.ndarray:
        src, ref = _as_f32_2d(src, 'src'), _as_f32_2d(ref, 'ref')
        norm = np.zeros(2, np.float64)
        _check(self._lib.hk_block_norm(self._h, C.byref(desc), _ptr(src), src.strides[0] // 4, _ptr(ref),
                                       ref.strides[0] // 4, src.shape[0], src.shape[1], _ptr(norm, _f64p)))
        return norm

    def compare_sums(self, src: np.ndarray, src_nodata, ref: np.ndarray, ref_nodata) -> np.ndarray:
        """ float64[7] = [sum s, sum r, sum s^2, sum r^2, sum s*r, sum (r-s)^2, N] over the jointly valid pixels
        (homonim/compare.py:243-255) """
        src, ref = _as_f32_2d(src, 'src'), _as_f32_2d(ref, 'ref')
        if src.shape != ref.shape:
            raise ValueError('`src` and `ref` shapes differ')
        sums = np.zeros(7, np.float64)
        (sm, sv), (rm, rv) = nodata_code(src_nodata), nodata_code(ref_nodata)
        _check(self._lib.hk_compare_sums(self._h, _ptr(src), src.strides[0] // 4, sm, sv, _ptr(ref), ref.strides[0] // 4,
                                         rm, rv, src.shape[0], src.shape[1], _ptr(sums, _f64p)))
        return sums

    def fit_apply(self, desc: FitDesc, src: np.ndarray, ref: np.ndarray, n_param_bands: int, want_params: bool,
                  want_corr: bool, norm_in: Optional[np.ndarray] = None, out_params: Optional[np.ndarray] = None,
                  out_corr: Optional[np.ndarray] = None, out_dtype: str = 'float32', out_nodata: Optional[float] = None):
        """
        -> (params | None, corr | None, norm, r2_fail_count).

        ``src`` / ``ref`` may be float32 or any dtype of DTYPE_CODES (converted on the device as rasterio would on
        read); ``out_dtype`` / ``out_nodata`` convert the corrected block on the device as the reference does on write
        (round half-to-even, clip, masked pixels -> ``out_nodata``).  ``out_params`` / ``out_corr`` let the caller
        supply (e.g. pinned) C-contiguous output arrays.
        """
        src, ref = _as_2d_native(src, 'src'), _as_2d_native(ref, 'ref')
        if src.shape != ref.shape:
            raise ValueError("'ref_ra' and 'src_ra' must have the same CRS, transform and shape")
        h, w = src.shape
        out_dtype = np.dtype(out_dtype)
        if out_dtype.name not in DTYPE_CODES:
            raise ValueError(f'unsupported output dtype {out_dtype}')
        keep_nan = out_nodata is None or (isinstance(out_nodata, float) and math.isnan(out_nodata))
        params = corr = None
        if want_params:
            params = out_params if out_params is not None else np.empty((n_param_bands, h, w), np.float32)
            assert params.shape == (n_param_bands, h, w) and params.dtype == np.float32 and params.flags['C_CONTIGUOUS']
        if want_corr:
            corr = out_corr if out_corr is not None else np.empty((h, w), out_dtype)
            assert corr.shape == (h, w) and corr.dtype == out_dtype and corr.flags['C_CONTIGUOUS']
        norm = np.zeros(2, np.float64)
        fail = C.c_uint64(0)
        nin = None
        if norm_in is not None:
            nin = np.ascontiguousarray(norm_in, dtype=np.float64)
        io = IoDesc(DTYPE_CODES[src.dtype.name], DTYPE_CODES[ref.dtype.name], DTYPE_CODES[out_dtype.name],
                    0 if (keep_nan and out_dtype.kind == 'f') or out_nodata is None else 1,
                    0.0 if out_nodata is None or keep_nan else float(out_nodata))
        vp = C.c_void_p
        _check(self._lib.hk_fit_apply_io(
            self._h, C.byref(desc), C.byref(io), src.ctypes.data_as(vp), src.strides[0] // src.dtype.itemsize,
            ref.ctypes.data_as(vp), ref.strides[0] // ref.dtype.itemsize, h, w,
            _ptr(nin, _f64p) if nin is not None else None, _ptr(params) if want_params else None, n_param_bands,
            corr.ctypes.data_as(vp) if want_corr else None, _ptr(norm, _f64p), C.byref(fail)))
        return params, corr, norm, int(fail.value)

    def fit_apply_block(self, desc: FitDesc, src: np.ndarray, ref: np.ndarray, window, corr_dst: Optional[np.ndarray],
                        params_dst: Optional[np.ndarray] = None, norm_in: Optional[np.ndarray] = None,
                        out_nodata: Optional[float] = None):
        """
        One block of RasterFuse.process (homonim/fuse.py:295-319) in one call: fit + apply on the read-block
        ``src`` / ``ref`` (2-D views, any dtype of DTYPE_CODES), and the window ``(row0, col0, rows, cols)`` of it -- the
        out-block -- written straight into ``corr_dst`` (2-D view of the caller's corrected raster where the out-block
        belongs, any DTYPE_CODES dtype, unit column stride) and ``params_dst`` (3-D view, float32).  With page-locked
        arrays (``pin`` / ``pinned_empty``) the transfers are asynchronous.  -> (norm, r2_fail_count)
        """
        src, ref = _as_2d_native(src, 'src'), _as_2d_native(ref, 'ref')
        if src.shape != ref.shape:
            raise ValueError("'ref_ra' and 'src_ra' must have the same CRS, transform and shape")
        h, w = src.shape
        row0, col0, rows, cols = (int(v) for v in window)
        vp = C.c_void_p
        n_param = 0
        stride = band_stride = 0
        out_dtype = np.dtype(np.float32)
        pstride = 0
        if corr_dst is not None:
            if corr_dst.shape != (rows, cols) or corr_dst.strides[1] != corr_dst.dtype.itemsize or \
                    corr_dst.strides[0] % corr_dst.dtype.itemsize or corr_dst.strides[0] < cols * corr_dst.dtype.itemsize:
                raise ValueError('`corr_dst` must be a (rows, cols) view of the window with unit column stride')
            out_dtype = corr_dst.dtype
            stride = corr_dst.strides[0] // corr_dst.dtype.itemsize
        if params_dst is not None:
            if params_dst.dtype != np.float32 or params_dst.ndim != 3 or params_dst.shape[1:] != (rows, cols) or \
                    params_dst.strides[2] != 4 or params_dst.strides[1] % 4 or params_dst.strides[0] % 4 or \
                    params_dst.strides[1] < 4 * cols or params_dst.strides[0] < 0:
                raise ValueError('`params_dst` must be a float32 (bands, rows, cols) view of the window with unit column stride')
            n_param = params_dst.shape[0]
            pstride = params_dst.strides[1] // 4  # its own row stride: the corrected and parameter rasters need not share one
            band_stride = params_dst.strides[0] // 4
            if corr_dst is None:
                stride = pstride
        if out_dtype.name not in DTYPE_CODES:
            raise ValueError(f'unsupported output dtype {out_dtype}')
        keep_nan = out_nodata is None or (isinstance(out_nodata, float) and math.isnan(out_nodata))
        io = IoDesc(DTYPE_CODES[src.dtype.name], DTYPE_CODES[ref.dtype.name], DTYPE_CODES[out_dtype.name],
                    0 if (keep_nan and out_dtype.kind == 'f') or out_nodata is None else 1,
                    0.0 if out_nodata is None or keep_nan else float(out_nodata))
        win = OutWindow(stride, band_stride, row0, col0, rows, cols, pstride)
        norm = np.zeros(2, np.float64)
        fail = C.c_uint64(0)
        nin = np.ascontiguousarray(norm_in, dtype=np.float64) if norm_in is not None else None
        _check(self._lib.hk_fit_apply_block(
            self._h, C.byref(desc), C.byref(io), src.ctypes.data_as(vp), src.strides[0] // src.dtype.itemsize,
            ref.ctypes.data_as(vp), ref.strides[0] // ref.dtype.itemsize, h, w,
            _ptr(nin, _f64p) if nin is not None else None,
            params_dst.ctypes.data_as(vp) if params_dst is not None else None, n_param,
            corr_dst.ctypes.data_as(vp) if corr_dst is not None else None, C.byref(win), _ptr(norm, _f64p), C.byref(fail)))
        return norm, int(fail.value)

    def apply(self, src: np.ndarray, params: np.ndarray) -> np.ndarray:
        src = _as_f32_2d(src, 'src')
        params = np.ascontiguousarray(params[:2], dtype=np.float32)
        if params.shape[-2:] != src.shape:
            raise ValueError("'param_ra' and 'src_ra' must have the same CRS, transform and shape")
        out = np.empty(src.shape, np.float32)
        _check(self._lib.hk_apply(self._h, _ptr(src), src.strides[0] // 4, _ptr(params), src.shape[0], src.shape[1],
                                  _ptr(out)))
        return out

    def refspace_fit_apply(self, desc: FitDesc, src: np.ndarray, ref: np.ndarray, down, up, down_resampling: int,
                           up_resampling: int, mask_partial: bool, n_param_bands: int, want_params: bool,
                           out_dtype: str = 'float32', out_nodata: Optional[float] = None,
                           out_corr: Optional[np.ndarray] = None):
        """ hk_refspace_fit_apply: RefSpaceModel.fit + apply of one block pair on different grids, all on the device.
        -> (params on the reference grid | None, corrected on the source grid, r2_fail_count) """
        src, ref = _as_2d_native(src, 'src'), _as_2d_native(ref, 'ref')
        out_dtype = np.dtype(out_dtype)
        keep_nan = out_nodata is None or (isinstance(out_nodata, float) and math.isnan(out_nodata))
        io = IoDesc(DTYPE_CODES[src.dtype.name], DTYPE_CODES[ref.dtype.name], DTYPE_CODES[out_dtype.name],
                    0 if (keep_nan and out_dtype.kind == 'f') or out_nodata is None else 1,
                    0.0 if out_nodata is None or keep_nan else float(out_nodata))
        sp = SpaceDesc()
        for i in range(4):
            sp.down[i], sp.up[i] = float(down[i]), float(up[i])
        sp.down_resampling, sp.up_resampling, sp.mask_partial = int(down_resampling), int(up_resampling), int(mask_partial)
        params = np.empty((n_param_bands, *ref.shape), np.float32) if want_params else None
        corr = out_corr if out_corr is not None else np.empty(src.shape, out_dtype)
        assert corr.shape == src.shape and corr.dtype == out_dtype and corr.flags['C_CONTIGUOUS']
        fail = C.c_uint64(0)
        vp = C.c_void_p
        _check(self._lib.hk_refspace_fit_apply(
            self._h, C.byref(desc), C.byref(io), C.byref(sp), src.ctypes.data_as(vp), src.strides[0] // src.dtype.itemsize,
            src.shape[0], src.shape[1], ref.ctypes.data_as(vp), ref.strides[0] // ref.dtype.itemsize, ref.shape[0],
            ref.shape[1], _ptr(params) if want_params else None, n_param_bands, corr.ctypes.data_as(vp), C.byref(fail)))
        return params, corr, int(fail.value)

    def reproject(self, src: np.ndarray, src_nodata, mapping, dst_shape, resampling: int, dst_fill: float) -> np.ndarray:
        """ hk_reproject: (bands, h, w) or (h, w) float32 -> same rank on the destination grid. """
        arr = np.ascontiguousarray(src, dtype=np.float32)
        squeeze = arr.ndim == 2
        if squeeze:
            arr = arr[None]
        nb, sh, sw = arr.shape
        dh, dw = int(dst_shape[0]), int(dst_shape[1])
        out = np.empty((nb, dh, dw), np.float32)
        mode, val = nodata_code(src_nodata)
        kx, ox, ky, oy = [float(v) for v in mapping]
        _check(self._lib.hk_reproject(self._h, _ptr(arr), nb, sh, sw, mode, val, kx, ox, ky, oy, int(resampling),
                                      _ptr(out), dh, dw, float(dst_fill)))
        return out[0] if squeeze else out

    def partial_mask(self, in_arr: np.ndarray, in_nodata, params: np.ndarray, kernel_shape, src: Optional[np.ndarray] = None,
                     want_params: bool = False, want_corr: bool = False, want_mask: bool = False, coverage: bool = False):
        """ mask_partial on a shared grid (hk_partial_mask) -> (masked params | None, corrected | None, mask | None).
        ``coverage``: ``in_arr`` is a coverage fraction (a mask re-projected with `average`), valid where >= 1. """
        in_arr = _as_f32_2d(in_arr, 'in')
        params = np.ascontiguousarray(params, dtype=np.float32)
        if params.ndim != 3 or params.shape[-2:] != in_arr.shape:
            raise ValueError("'param_ra' and 'src_ra' must have the same CRS, transform and shape")
        h, w = in_arr.shape
        if src is not None:
            src = _as_f32_2d(src, 'src')
        mode, val = (3, 0.0) if coverage else nodata_code(in_nodata)
        p_out = np.empty_like(params) if want_params else None
        c_out = np.empty((h, w), np.float32) if want_corr else None
        m_out = np.empty((h, w), np.uint8) if want_mask else None
        _check(self._lib.hk_partial_mask(
            self._h, _ptr(in_arr), in_arr.strides[0] // 4, mode, val, _ptr(params), params.shape[0],
            _ptr(src) if src is not None else None, (src.strides[0] // 4) if src is not None else 0, h, w,
            int(kernel_shape[0]), int(kernel_shape[1]), _ptr(p_out) if want_params else None,
            _ptr(c_out) if want_corr else None, m_out.ctypes.data_as(_P(C.c_uint8)) if want_mask else None))
        return p_out, c_out, m_out

    # -- pinned host memory (async H2D / D2H) ----------------------------------------------------------------------------
    def pinned_empty(self, shape, dtype=np.float32) -> np.ndarray:
        """ A numpy array in page-locked host memory: copies to / from it overlap with kernels of other calls.  The
        memory is released when the array (and every view of it) is garbage collected. """
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        _check(self._lib.hk_host_alloc(self._h, max(nbytes, 1), C.byref(p)))
        lib, addr = self._lib, p.value

        class _Owner:
            def __del__(self_inner):
                if sys.is_finalizing():  # the process is going away: its page-locked memory goes with it
                    return
                try:
                    lib.hk_host_free(None, C.c_void_p(addr))  # page-locked memory outlives the context it came from
                except Exception:
                    pass

        buf = (C.c_byte * max(nbytes, 1)).from_address(addr)
        buf._owner = _Owner()
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def pin(self, arr: np.ndarray) -> bool:
        """ Page-lock an existing C-contiguous array in place (hipHostRegister); undo with ``unpin``.  Registrations are
        counted per address range across the process: concurrent users of one array (two ``RasterFuse.process`` calls on
        the same source raster) share one registration and the last ``unpin`` removes it.  Memory that is page-locked
        already by somebody else (``pinned_empty``, an enclosing registration) is left as it is.  -> whether the range is
        page-locked now. """
        if not arr.flags['C_CONTIGUOUS']:
            raise ValueError('only C-contiguous arrays can be page-locked in place')
        key = (arr.ctypes.data, arr.nbytes)
        with _pin_lock:
            ent = _pins.get(key)
            if ent is not None:
                ent[0] += 1
                return True
            rc = self._lib.hk_host_register(self._h, C.c_void_p(key[0]), key[1])
            if rc == HK_ERR_ALREADY:
                _pins[key] = [1, False]   # not ours to unregister
                return True
            _check(rc)
            _pins[key] = [1, True]
            return True

    def unpin(self, arr: np.ndarray):
        key = (arr.ctypes.data, arr.nbytes)
        with _pin_lock:
            ent = _pins.get(key)
            if ent is None:
                return
            ent[0] -= 1
            if ent[0] > 0:
                return
            del _pins[key]
            if ent[1]:
                _check(self._lib.hk_host_unregister(self._h, C.c_void_p(key[0])))

    def stream_probe_dev(self, a_dptr: int, b_dptr: int, out_dptr: int, nbytes: int, stream: int = 0):
        """ One launch of the flat 2-read 1-write float4 stream out = a + b over three device buffers (hk_stream_probe_dev). """
        _check(self._lib.hk_stream_probe_dev(self._h, C.c_void_p(a_dptr), C.c_void_p(b_dptr), C.c_void_p(out_dptr), nbytes, stream))

    def checksum_dev(self, plane_dptr: int, stride: int, height: int, width: int, stream: int = 0) -> int:
        """ Sum of the 32-bit patterns of a height x width window of a device-resident float32 plane, modulo 2^64
        (hk_debug_checksum_dev: exact and order-free; synchronises the stream). """
        out = C.c_uint64(0)
        _check(self._lib.hk_debug_checksum_dev(self._h, C.c_void_p(plane_dptr), stride, height, width, stream, C.byref(out)))
        return int(out.value)

    # -- device-resident helpers (bench / streaming) ------------------------------------------------------------------
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _check(self._lib.hk_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, dptr: int):
        _check(self._lib.hk_dev_free(self._h, C.c_void_p(dptr)))

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        _check(self._lib.hk_memcpy_h2d(self._h, C.c_void_p(dptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def d2h(self, arr: np.ndarray, dptr: int, nbytes: Optional[int] = None):
        assert arr.flags['C_CONTIGUOUS']
        _check(self._lib.hk_memcpy_d2h(self._h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(dptr),
                                       arr.nbytes if nbytes is None else nbytes))

    def memset(self, dptr: int, value: int, nbytes: int):
        _check(self._lib.hk_memset(self._h, C.c_void_p(dptr), value, nbytes))

    def block_norm_split_phase(self, desc: FitDesc, job: DevJob, phase: int, world_size: int, xchg_dev: int, norm_dev: int):
        """ Queue phase 0..5 of the split-block statistics on this rank's slab (see homonim_amd/split_norm.py). """
        _check(self._lib.hk_block_norm_split_dev(self._h, C.byref(desc), C.byref(job), phase, world_size,
                                                 C.c_void_p(xchg_dev), C.c_void_p(norm_dev)))

    def split_exchange_doubles(self, n_bands: int) -> int:
        return int(self._lib.hk_block_norm_split_exchange_doubles(n_bands))

    # -- the library's own RCCL communicator (one per context; the split-block statistics are its one user) ------------
    def comm_init(self, unique_id: bytes, rank: int, world_size: int):
        """ Join the RCCL communicator named by ``unique_id`` (``comm_unique_id()`` of rank 0, handed over by the launcher;
        ``homonim_amd.dist.init_comm`` does that).  Collective: returns when every rank has joined. """
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f'the communicator id has {COMM_ID_BYTES} bytes')
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(self._lib.hk_comm_init(self._h, buf, rank, world_size))

    def comm_destroy(self):
        _check(self._lib.hk_comm_destroy(self._h))

    def comm_info(self):
        """ -> (rank, world_size) of the context's communicator; (-1, 0) without one. """
        r, w = C.c_int32(-1), C.c_int32(0)
        _check(self._lib.hk_comm_info(self._h, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)

    def comm_allreduce_f64_dev(self, buf_dptr: int, count: int, stream: int = 0):
        _check(self._lib.hk_comm_allreduce_f64_dev(self._h, C.c_void_p(buf_dptr), count, stream))

    def block_norm_split_comm_dev(self, desc: FitDesc, job: DevJob, norm_dev: int):
        """ Split-block statistics of this rank's slab over the context's communicator: six phases + five RCCL all-reduces
        queued on the job's stream (hk_block_norm_split_comm_dev; asynchronous, collective). """
        _check(self._lib.hk_block_norm_split_comm_dev(self._h, C.byref(desc), C.byref(job), C.c_void_p(norm_dev)))

    def job_scratch_bytes(self, job: DevJob) -> int:
        """ Size of the optional DevJob.scratch (gain-offset with an r2 threshold: the in-painting's inputs). """
        return int(self._lib.hk_dev_job_scratch_bytes(job.n_bands, job.height, job.stride, job.band_stride))

    def fit_apply_dev(self, desc: FitDesc, job: DevJob):
        _check(self._lib.hk_fit_apply_dev(self._h, C.byref(desc), C.byref(job)))

    def counts_pending(self, counts: np.ndarray) -> bool:
        """ Do the counters of a launch (``fail_counts_async``) call for its second half, ``inpaint_dev_counts``?  (failing
        pixels in some band: hk_counts_pending) """
        c = np.ascontiguousarray(counts, dtype=np.uint64)
        return bool(self._lib.hk_counts_pending(c.ctypes.data_as(_P(C.c_uint64)), int(c.size)))

    def fail_counts_async(self, job: DevJob, host_counts: np.ndarray, ready_event: int):
        """ Queue the copy of the job's failure counters into a PINNED uint64 array, their clearing and `ready_event`. """
        assert host_counts.dtype == np.uint64 and host_counts.flags['C_CONTIGUOUS']
        _check(self._lib.hk_fail_counts_async(self._h, C.byref(job), host_counts.ctypes.data_as(_P(C.c_uint64)),
                                              C.c_void_p(ready_event)))

    def event_sync(self, ev: int):
        _check(self._lib.hk_event_sync(self._h, C.c_void_p(ev)))

    def inpaint_dev_counts(self, desc: FitDesc, job: DevJob, counts: np.ndarray) -> int:
        """ In-paint the bands whose count (host array, from fail_counts_async) is non-zero; only queues work. """
        n = C.c_uint64(0)
        counts = np.ascontiguousarray(counts, np.uint64)
        _check(self._lib.hk_inpaint_dev_counts(self._h, C.byref(desc), C.byref(job), counts.ctypes.data_as(_P(C.c_uint64)),
                                               C.byref(n)))
        return int(n.value)

    def inpaint_dev(self, desc: FitDesc, job: DevJob) -> int:
        """ In-paint the bands of a device-resident job whose r2 mask has failures; returns the failure count. """
        n = C.c_uint64(0)
        _check(self._lib.hk_inpaint_dev(self._h, C.byref(desc), C.byref(job), C.byref(n)))
        return int(n.value)

    def block_norm_dev(self, desc: FitDesc, job: DevJob, norm_dptr: int):
        _check(self._lib.hk_block_norm_dev(self._h, C.byref(desc), C.byref(job), C.c_void_p(norm_dptr)))

    @staticmethod
    def job_array(jobs) -> 'C.Array':
        """ The jobs of a batched launch as one contiguous ctypes array (build it once, pass it to every step). """
        if isinstance(jobs, C.Array):
            return jobs
        arr = (DevJob * len(jobs))()
        for i, j in enumerate(jobs):
            C.memmove(C.byref(arr, i * C.sizeof(DevJob)), C.byref(j), C.sizeof(DevJob))
        return arr

    def block_norm_batch_dev(self, desc: FitDesc, jobs, norm_dptr: int):
        """ The block statistics of many device-resident jobs in one launch per kernel stage (hk_block_norm_batch_dev):
        job 0's n_bands x 2 float64 first, then job 1's ... into `norm_dptr`; all on jobs[0].stream. """
        arr = self.job_array(jobs)
        _check(self._lib.hk_block_norm_batch_dev(self._h, C.byref(desc), arr, len(arr), C.c_void_p(norm_dptr)))

    def fit_apply_batch_dev(self, desc: FitDesc, jobs):
        """ Many device-resident jobs as ONE fused launch on jobs[0].stream (hk_fit_apply_batch_dev); every job reads its own
        ``norm`` pointer and, with an r2 threshold, is finished by its own ``inpaint_dev*`` call. """
        arr = self.job_array(jobs)
        _check(self._lib.hk_fit_apply_batch_dev(self._h, C.byref(desc), arr, len(arr)))

    def fail_counts_batch_async(self, jobs, host_counts: np.ndarray, ready_event: int):
        """ ``fail_counts_async`` for the jobs of a batch: all their counters, job after job, into one PINNED uint64 array. """
        arr = self.job_array(jobs)
        assert host_counts.dtype == np.uint64 and host_counts.flags['C_CONTIGUOUS']
        assert host_counts.size >= sum(j.n_bands for j in arr)
        _check(self._lib.hk_fail_counts_batch_async(self._h, arr, len(arr), host_counts.ctypes.data_as(_P(C.c_uint64)),
                                                    C.c_void_p(ready_event)))

    def debug_stage_stamps(self, reset: bool = True) -> np.ndarray:
        """ Stage counters of a -DHK_STAMPS build of the fused kernel (hk_debug_stage_stamps; zeros in the shipped build). """
        out = (C.c_uint64 * 16)()
        _check(self._lib.hk_debug_stage_stamps(self._h, out, 1 if reset else 0))
        return np.array(list(out), dtype=np.uint64)

    def compare_sums_dev(self, job: DevJob, src_nodata, ref_nodata, sums_dptr: int):
        (sm, sv), (rm, rv) = nodata_code(src_nodata), nodata_code(ref_nodata)
        _check(self._lib.hk_compare_sums_dev(self._h, C.byref(job), sm, sv, rm, rv, C.c_void_p(sums_dptr)))

    def synth_fill_dev(self, src_dptr, ref_dptr, n_bands, height, width, stride, band_stride, seed=0, nodata_variant=0,
                       stream=0):
        _check(self._lib.hk_synth_fill_dev(self._h, C.c_void_p(src_dptr), C.c_void_p(ref_dptr), n_bands, height, width,
                                           stride, band_stride, seed, nodata_variant, stream))

    def event(self) -> int:
        e = C.c_void_p()
        _check(self._lib.hk_event_create(self._h, C.byref(e)))
        return e.value

    def event_destroy(self, ev: int):
        self._lib.hk_event_destroy(self._h, C.c_void_p(ev))

    def event_record(self, ev: int, stream: int = 0):
        _check(self._lib.hk_event_record(self._h, C.c_void_p(ev), stream))

    def stream_wait_event(self, stream: int, ev: int):
        """ Work queued on `stream` after this call waits (on the device) for `ev` (hk_stream_wait_event). """
        _check(self._lib.hk_stream_wait_event(self._h, stream, C.c_void_p(ev)))

    def event_elapsed_ms(self, start: int, stop: int) -> float:
        ms = C.c_float(0)
        _check(self._lib.hk_event_elapsed_ms(self._h, C.c_void_p(start), C.c_void_p(stop), C.byref(ms)))
        return float(ms.value)

    def stream_sync(self, stream: int = 0):
        _check(self._lib.hk_stream_sync(self._h, stream))


_default_ctx = None
_default_lock = threading.Lock()


def default_context() -> Context:
    """ Process-wide context on the device named by HOMONIM_AMD_DEVICE (else LOCAL_RANK, else 0). """
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            dev = int(os.environ.get('HOMONIM_AMD_DEVICE', os.environ.get('LOCAL_RANK', '0')))
            _default_ctx = Context(dev, n_streams=int(os.environ.get('HOMONIM_AMD_STREAMS', '4')))
    return _default_ctx


_ctx_cache = {}


def get_context(device: int, n_streams: int = 4) -> Context:
    """ Process-wide cached context per (device, n_streams): creating one costs stream + slab allocations. """
    with _default_lock:
        key = (int(device), int(n_streams))
        ctx = _ctx_cache.get(key)
        if ctx is None or ctx.handle is None:
            ctx = _ctx_cache[key] = Context(*key)
    return ctx


def device_count() -> int:
    n = C.c_int(0)
    load_library().hk_device_count(C.byref(n))
    return int(n.value)


def staging_counters(reset: bool = False):
    """ (copies queued straight from / to page-locked caller arrays, chunks through the pinned staging ring) since the process
    started or the last reset (hk_debug_staging_counters; test aid). """
    out = (C.c_uint64 * 2)()
    _check(load_library().hk_debug_staging_counters(out, 1 if reset else 0))
    return int(out[0]), int(out[1])


def debug_fail_after_d2h(on: bool):
    """ Fault injection of the test-suite (hk_debug_fail_after_d2h): host-pointer fits fail behind their queued result copies. """
    _check(load_library().hk_debug_fail_after_d2h(1 if on else 0))


def build_ledger(reset: bool = False) -> dict:
    """ {kernel build: launches since the library was loaded or the last reset} for EVERY kernel build in the library, launched or
    not (hk_debug_build_ledger; test aid -- tests/conftest.py keeps the ledger of which builds met the oracle).  Records of one name
    (a kernel with several launch sites) are added up.  No device call: works without a GPU. """
    lib = load_library()
    need = C.c_size_t(0)
    _check(lib.hk_debug_build_ledger(None, 0, C.byref(need), 0))
    buf = C.create_string_buffer(int(need.value) + 4096)   # (room for nothing: records only appear when the library is loaded)
    _check(lib.hk_debug_build_ledger(buf, len(buf), C.byref(need), 1 if reset else 0))
    out = {}
    for line in buf.value.decode().splitlines():
        name, _, count = line.rpartition('\t')
        out[name] = out.get(name, 0) + int(count)
    return out


def device_pci_bus_id(device: int) -> str:
    """ PCI bus address of HIP device ``device`` as sysfs spells it, e.g. '0000:c1:00.0' (homonim_amd/topology.py). """
    buf = C.create_string_buffer(32)
    _check(load_library().hk_device_pci_bus_id(int(device), buf, 32))
    return buf.value.decode()
