"""
homonim_amd -- MI355X-native (gfx950) implementation of homonim's sliding-window kernel-model fit/apply hot path.

Python host code -> ctypes C ABI (include/homonim_hk.h) -> hand-written HIP kernels (homonim_amd/csrc).
Public names follow the reference package (``homonim.KernelModel``, ``homonim.enums.Model`` ...).
"""
from homonim_amd.enums import Model, ProcCrs, Resampling
from homonim_amd.errors import ConfigWarning, DeviceError, HomonimError
from homonim_amd.geo import Affine, CRS, Window
from homonim_amd.kernel_model import KernelModel, RefSpaceModel, SrcSpaceModel
from homonim_amd.raster_array import RasterArray
from homonim_amd.fuse import RasterFuse
from homonim_amd.compare import RasterCompare

__version__ = '0.1.0'
__all__ = [
    'Model', 'ProcCrs', 'Resampling', 'ConfigWarning', 'DeviceError', 'HomonimError', 'Affine', 'CRS', 'Window',
    'KernelModel', 'RefSpaceModel', 'SrcSpaceModel', 'RasterArray', 'RasterFuse', 'RasterCompare',
]
