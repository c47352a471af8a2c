"""
Enumerations of the hot-path API surface.  Values mirror the reference so configs/CLI strings are interchangeable:
``Model`` <- homonim/enums.py:22-41, ``ProcCrs`` <- homonim/enums.py:45-53.  ``Resampling`` stands in for
``rasterio.enums.Resampling`` (rasterio is not a dependency of this package); only the members the reference's
``KernelModel.create_config`` exposes by default are meaningful here.
"""
from enum import Enum, IntEnum


class Model(str, Enum):
    """ Linear model variants for correcting to surface reflectance (homonim/enums.py:22-41). """
    gain = 'gain'
    gain_blk_offset = 'gain-blk-offset'
    gain_offset = 'gain-offset'


class ProcCrs(str, Enum):
    """ CRS and pixel grid in which images are processed (homonim/enums.py:45-53). """
    auto = 'auto'
    src = 'src'
    ref = 'ref'


class Resampling(IntEnum):
    """ Same names and integer values as rasterio.enums.Resampling (GDAL GRA_* codes). """
    nearest = 0
    bilinear = 1
    cubic = 2
    cubic_spline = 3
    lanczos = 4
    average = 5
    mode = 6
    gauss = 7
    max = 8
    min = 9
    med = 10
    q1 = 11
    q3 = 12
    sum = 13
    rms = 14
