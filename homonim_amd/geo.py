"""
Minimal geo-referencing value types for the ``RasterArray`` carrier.  The reference uses ``rasterio.crs.CRS`` and
``rasterio.Affine`` (homonim/raster_array.py:79-90); rasterio/GDAL are not dependencies of this package, so these
small stand-alone equivalents carry the same information.  If rasterio is installed its own types are accepted
wherever these are.
"""
from collections import namedtuple
from typing import Tuple


class Affine(namedtuple('Affine', 'a b c d e f')):
    """ 2-D affine geo-transform, x = a*col + b*row + c, y = d*col + e*row + f (rasterio / affine convention). """
    __slots__ = ()

    @classmethod
    def identity(cls) -> 'Affine':
        return cls(1., 0., 0., 0., 1., 0.)

    @classmethod
    def translation(cls, xoff: float, yoff: float) -> 'Affine':
        return cls(1., 0., xoff, 0., 1., yoff)

    @classmethod
    def scale(cls, sx: float, sy: float = None) -> 'Affine':
        return cls(sx, 0., 0., 0., sx if sy is None else sy, 0.)

    def __mul__(self, other):
        if isinstance(other, Affine) or (hasattr(other, 'a') and hasattr(other, 'f')):
            a, b, c, d, e, f = self
            oa, ob, oc, od, oe, of = other.a, other.b, other.c, other.d, other.e, other.f
            return Affine(
                a * oa + b * od, a * ob + b * oe, a * oc + b * of + c, d * oa + e * od, d * ob + e * oe,
                d * oc + e * of + f
            )
        x, y = other
        return (self.a * x + self.b * y + self.c, self.d * x + self.e * y + self.f)

    def __invert__(self):
        det = self.a * self.e - self.b * self.d
        if det == 0:
            raise ValueError('transform is not invertible')
        ia, ib, id_, ie = self.e / det, -self.b / det, -self.d / det, self.a / det
        return Affine(ia, ib, -(ia * self.c + ib * self.f), id_, ie, -(id_ * self.c + ie * self.f))


class CRS:
    """ Opaque coordinate reference system label (compared by its string). """

    def __init__(self, name: str = 'EPSG:3857'):
        self._name = str(name)

    @classmethod
    def from_string(cls, s: str) -> 'CRS':
        return cls(s)

    def to_string(self) -> str:
        return self._name

    def __eq__(self, other):
        return isinstance(other, CRS) and other._name.lower() == self._name.lower()

    def __hash__(self):
        return hash(self._name.lower())

    def __repr__(self):
        return f"CRS('{self._name}')"


class Window(namedtuple('Window', 'col_off row_off width height')):
    """ Pixel window (rasterio.windows.Window field order). """
    __slots__ = ()

    def toslices(self) -> Tuple[slice, slice]:
        return (slice(self.row_off, self.row_off + self.height), slice(self.col_off, self.col_off + self.width))


def window_transform(window: Window, transform: Affine) -> Affine:
    """ rasterio.windows.transform: the transform of a window into ``transform``. """
    return transform * Affine.translation(window.col_off, window.row_off)


def _is_crs(obj) -> bool:
    if isinstance(obj, CRS):
        return True
    return type(obj).__module__.startswith('rasterio') and type(obj).__name__ == 'CRS'


def _is_affine(obj) -> bool:
    if isinstance(obj, Affine):
        return True
    return type(obj).__name__ == 'Affine' and all(hasattr(obj, k) for k in 'abcdef')


def grid_mapping(src_transform, dst_transform) -> Tuple[float, float, float, float]:
    """
    (kx, ox, ky, oy) with ``src_col = kx * dst_col + ox`` and ``src_row = ky * dst_row + oy`` on continuous pixel
    coordinates (integers = pixel edges) between two axis-aligned grids of one CRS.  A negative factor means the two
    grids run in opposite directions along that axis (e.g. a south-up raster against a north-up one): the device
    re-samplers take positive factors, ``RasterArray.reproject`` flips the source array for the others.
    """
    s, d = src_transform, dst_transform
    if s.b or s.d or d.b or d.d:
        raise NotImplementedError('re-projection between rotated / sheared grids is not built')
    if not (s.a and s.e and d.a and d.e):
        raise NotImplementedError('degenerate geo-transform')
    kx, ky = d.a / s.a, d.e / s.e
    return kx, (d.c - s.c) / s.a, ky, (d.f - s.f) / s.e
