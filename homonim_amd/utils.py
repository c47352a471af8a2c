"""
The four pure helpers of homonim/utils.py the hot path touches, restated (same names, arguments and errors):
``nan_equals`` (:54-56), ``validate_kernel_shape`` (:104-133), ``overlap_for_kernel`` (:136-153),
``validate_threads`` (:156-164).
"""
import warnings
from multiprocessing import cpu_count
from typing import Tuple, Union

import numpy as np

from homonim_amd.enums import Model
from homonim_amd.errors import ConfigWarning


def nan_equals(a: Union[np.ndarray, float], b: Union[np.ndarray, float]) -> np.ndarray:
    """ Element-wise a == b, with nan == nan. """
    return (a == b) | (np.isnan(a) & np.isnan(b))


def validate_kernel_shape(kernel_shape: Tuple[int, int], model: Model = Model.gain_blk_offset) -> Tuple[int, int]:
    """ Check a kernel (height, width) for validity; ValueError / ConfigWarning exactly where the reference has them. """
    ks = np.array(kernel_shape).astype(int)
    if not np.all(np.mod(ks, 2) == 1):
        raise ValueError('`kernel_shape` must be odd in both dimensions.')
    if model == Model.gain_offset:
        if np.prod(ks) < 2:
            raise ValueError('`kernel_shape` area should contain at least 2 elements for the gain-offset model.')
        elif np.prod(ks) < 25:
            warnings.warn(
                'A `kernel_shape` of at least 25 elements is recommended for the gain-offset model.',
                category=ConfigWarning
            )
    if not np.all(ks >= 1):
        raise ValueError('`kernel_shape` must be a minimum of one in both dimensions.')
    return tuple(int(k) for k in ks)


def overlap_for_kernel(kernel_shape: Tuple[int, int]) -> Tuple[int, int]:
    """ Block overlap (rows, cols) = ceil(kernel_shape / 2) (utils.py:136-153). """
    ks = np.array(kernel_shape).astype(int)
    return tuple(int(v) for v in np.ceil(ks / 2).astype('int'))


def validate_threads(threads: int) -> int:
    """ 0 = all processors; more than the processor count is an error (utils.py:156-164). """
    _cpu_count = cpu_count()
    threads = _cpu_count if threads == 0 else threads
    if threads > _cpu_count:
        raise ValueError(f"'threads' is limited to the number of processors ({_cpu_count})")
    return threads
