""" Exception / warning classes, same names as homonim/errors.py:21-62 so callers' ``except`` clauses keep working. """


class HomonimError(Exception):
    """ Root exception class. """


class UnsupportedImageError(HomonimError):
    pass


class ImageContentError(HomonimError):
    pass


class BlockSizeError(HomonimError):
    """ Raised when the image block size is invalid. """


class ImageProfileError(HomonimError):
    """ Raised when an image profile is invalid. """


class ImageFormatError(HomonimError):
    pass


class IoError(HomonimError):
    pass


class HomonimWarning(RuntimeWarning):
    """ Homonim runtime warning. """


class BandMatchWarning(HomonimWarning):
    pass


class ImageFormatWarning(HomonimWarning):
    pass


class ConfigWarning(HomonimWarning):
    """ Warn about configuration issues. """


class DeviceError(HomonimError, RuntimeError):
    """ (this package only) the HIP library is missing or the GPU call failed. """
