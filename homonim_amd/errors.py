"""
Exception and warning types of the package.

The names are those of homonim/errors.py:21-62 -- they are part of the call surface (``except homonim.errors.IoError``
keeps working after switching packages); what raises each of them HERE is noted on the class.
"""


class HomonimError(Exception):
    """ Base of everything this package raises on its own account. """


class UnsupportedImageError(HomonimError):
    """ Kept for API compatibility (the reference raises it for 12-bit JPEG GeoTIFFs; homonim_amd/tiff.py reports
    unsupported files as IoError). """


class ImageContentError(HomonimError):
    """ RasterFuse / RasterCompare: the reference raster does not cover the source raster. """


class BlockSizeError(HomonimError):
    """ fuse.block_pairs: ``max_block_mem`` leaves blocks smaller than the kernel overlap. """


class ImageProfileError(HomonimError):
    """ RasterArray.from_profile: the profile dict lacks a required key. """


class ImageFormatError(HomonimError):
    """ Kept for API compatibility (band / format validation lives with GDAL in the reference). """


class IoError(HomonimError):
    """ A closed RasterFuse / RasterCompare was used, or homonim_amd/tiff.py met a file outside its TIFF subset. """


class HomonimWarning(RuntimeWarning):
    """ Base of the package's warnings. """


class BandMatchWarning(HomonimWarning):
    """ Kept for API compatibility (wavelength band matching is not part of this package: bands pair in file order). """


class ImageFormatWarning(HomonimWarning):
    """ Kept for API compatibility. """


class ConfigWarning(HomonimWarning):
    """ A legal but questionable configuration: gain-offset kernels under 25 pixels, very small auto block shapes. """


class DeviceError(HomonimError, RuntimeError):
    """ (this package only) libhomonim_hk.so is missing, no gfx950 GPU is usable, or a HIP call failed -- there is no CPU
    path to fall back to. """
