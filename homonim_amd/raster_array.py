"""
``RasterArray``: the masked, geo-referenced ndarray the kernel models exchange -- the boundary type of the hot path.

Same public attribute names and observable semantics as the in-memory part of the reference class
(homonim/raster_array.py: constructor checks :72-90, ``from_profile`` :95-127, ``array``/``mask``/``nodata``
properties :223-351, ``profile``/``proj_profile`` :281-296, ``copy`` :389-391):

* the validity mask is derived lazily from ``nodata`` with nan-aware equality and cached; assigning ``array``,
  ``mask`` or ``nodata`` drops the cache;
* ``nodata = None`` means "every pixel valid"; changing a numeric/NaN nodata re-labels the currently masked pixels;
* multi-band arrays are band-major and a pixel is valid if it is valid in ANY band.

``reproject`` (raster_array.py:526-578) runs on the GPU for same-CRS, north-up, axis-aligned grids with the nearest /
bilinear / cubic_spline / average kernels -- a restatement of GDAL's warp kernels, see hk_resample.hip.  Dataset IO
stays outside this package (GDAL).
"""
from typing import Dict, Optional, Tuple

import numpy as np

from homonim_amd.enums import Resampling
from homonim_amd.errors import ImageProfileError
from homonim_amd.geo import Window, _is_affine, _is_crs, grid_mapping, window_transform
from homonim_amd.utils import nan_equals

_PROFILE_GEO_KEYS = ('crs', 'transform', 'nodata')
_PROFILE_ALLOC_KEYS = ('width', 'height', 'count', 'dtype')


class RasterArray:
    default_nodata = float('nan')
    default_dtype = 'float32'

    __slots__ = ('_array', '_crs', '_transform', '_nodata', '_mask')

    def __init__(self, array: np.ndarray, crs, transform, nodata: Optional[float] = default_nodata,
                 window: Optional[Window] = None):
        if array.ndim not in (2, 3):
            raise ValueError('`array` must be have 2 or 3 dimensions with bands along the first dimension')
        if window is not None and tuple(array.shape[-2:]) != (window.height, window.width):
            raise ValueError('`window` and `array` width and height must match')
        if not _is_crs(crs):
            raise TypeError('`crs` must be a CRS instance')
        if not _is_affine(transform):
            raise TypeError('`transform` must be an Affine instance')
        self._array = array
        self._crs = crs
        self._transform = transform if window is None else window_transform(window, transform)
        self._nodata = nodata
        self._mask = None

    # -- construction -------------------------------------------------------------------------------------------------
    @classmethod
    def from_profile(cls, array: Optional[np.ndarray], profile: Dict, window: Optional[Window] = None) -> 'RasterArray':
        """ Build from a rasterio-style profile dict; ``array=None`` allocates a nodata-filled (count, h, w) array. """
        if any(k not in profile for k in _PROFILE_GEO_KEYS):
            raise ImageProfileError("'profile' should include 'crs', 'transform' and 'nodata' keys")
        if array is None:
            if any(k not in profile for k in _PROFILE_ALLOC_KEYS):
                raise ImageProfileError("'profile' should include 'width', 'height', 'count' and 'dtype' keys")
            array = np.full(
                (profile['count'], profile['height'], profile['width']), profile['nodata'], dtype=profile['dtype']
            )
        return cls(array, profile['crs'], profile['transform'], nodata=profile['nodata'], window=window)

    def copy(self) -> 'RasterArray':
        """ Deep copy. """
        return RasterArray(self._array.copy(), self._crs, self._transform, nodata=self._nodata)

    # -- geometry -----------------------------------------------------------------------------------------------------
    @property
    def crs(self):
        return self._crs

    @property
    def transform(self):
        return self._transform

    @property
    def shape(self) -> Tuple[int, int]:
        """ (height, width) """
        return tuple(self._array.shape[-2:])

    @property
    def height(self) -> int:
        return self._array.shape[-2]

    @property
    def width(self) -> int:
        return self._array.shape[-1]

    @property
    def count(self) -> int:
        return 1 if self._array.ndim == 2 else self._array.shape[0]

    @property
    def dtype(self) -> str:
        return self._array.dtype.name

    @property
    def res(self) -> Tuple[float, float]:
        """ (x, y) pixel size """
        return self._transform.a, -self._transform.e

    @property
    def profile(self) -> Dict:
        h, w = self.shape
        return dict(crs=self._crs, transform=self._transform, nodata=self._nodata, count=self.count, width=w,
                    height=h, dtype=self.dtype)

    @property
    def proj_profile(self) -> Dict:
        return dict(crs=self._crs, transform=self._transform, shape=self.shape)

    # -- data, mask, nodata -------------------------------------------------------------------------------------------
    @property
    def array(self) -> np.ndarray:
        return self._array

    @array.setter
    def array(self, value: np.ndarray):
        if tuple(value.shape[-2:]) != self.shape:
            raise ValueError("'value' and 'array' shapes must match")
        self._array = value
        self._mask = None

    def _valid(self) -> np.ndarray:
        if self._nodata is None:
            return np.ones(self.shape, dtype=bool)
        valid = ~nan_equals(self._array, self._nodata)
        return valid if valid.ndim == 2 else valid.any(axis=0)

    @property
    def mask(self) -> np.ndarray:
        """ 2-D bool, True where the pixel is valid. """
        if self._mask is None:
            self._mask = self._valid()
        return self._mask

    @mask.setter
    def mask(self, value: np.ndarray):
        # pixels outside `value` become nodata; the cache is dropped (other pixels may equal nodata too)
        self._array[..., ~value] = self._nodata
        self._mask = None

    @property
    def mask_ra(self) -> 'RasterArray':
        """ The mask as a uint8 RasterArray without nodata. """
        return RasterArray(self.mask.astype('uint8', copy=False), self._crs, self._transform, nodata=None)

    @property
    def nodata(self) -> Optional[float]:
        return self._nodata

    @nodata.setter
    def nodata(self, value: Optional[float]):
        relabel = value is not None and self._nodata is not None and not nan_equals(value, self._nodata)
        if relabel:
            self._array[..., ~self.mask] = value
        if relabel or value is None or self._nodata is None:
            self._nodata = value
            self._mask = None

    # -- re-sampling --------------------------------------------------------------------------------------------------
    def reproject(self, crs=None, transform=None, shape: Optional[Tuple[int, int]] = None,
                  nodata: Optional[float] = default_nodata, dtype: str = default_dtype,
                  resampling: Resampling = Resampling.lanczos, context=None) -> 'RasterArray':
        """
        Re-sample onto another grid of the same CRS (raster_array.py:526-578).  ``transform`` needs ``shape``; the default
        is this array's own grid.  Returns a float32 RasterArray with ``nodata`` where nothing valid contributes
        (0 when ``nodata`` is None, as GDAL leaves the zero-initialised destination).
        """
        if transform is not None and shape is None:
            raise ValueError('If `transform` is specified, `shape` is required')
        if isinstance(resampling, str):
            resampling = Resampling[resampling]
        crs = crs or self._crs
        if crs != self._crs:
            raise NotImplementedError('re-projection between different CRSs is not built (GDAL warp)')
        transform = transform or self._transform
        shape = tuple(shape or self.shape)
        if np.dtype(dtype or self.dtype) != np.float32:
            raise NotImplementedError('re-projection yields float32 only')
        from homonim_amd import _hk  # deferred: the carrier itself needs no GPU
        ctx = context or _hk.default_context()
        fill = 0.0 if nodata is None else float(nodata)
        kx, ox, ky, oy = grid_mapping(self._transform, transform)
        src = self._array
        # grids of opposite orientation along an axis (south-up against north-up, mirrored columns): flip the source along
        # it -- in the flipped array the source coordinate is (size - coordinate), i.e. factor -k and offset size - o.  All of
        # GDAL's re-sampling kernels are symmetric, so this is the same re-sampling.
        if ky < 0:
            src, ky, oy = src[..., ::-1, :], -ky, src.shape[-2] - oy
        if kx < 0:
            src, kx, ox = src[..., ::-1], -kx, src.shape[-1] - ox
        out = ctx.reproject(src, self._nodata, (kx, ox, ky, oy), shape, int(resampling), fill)
        return RasterArray(out, crs, transform, nodata=nodata)
