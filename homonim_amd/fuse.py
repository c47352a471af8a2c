"""
Block scheduler + ``RasterFuse`` surface for the MI355X hot path.

The reference's ``RasterFuse.process`` (homonim/fuse.py:321-408) walks (band x spatial block) work items produced by
``RasterPairReader.block_pairs`` (homonim/raster_pair.py:342-428) and, per item, runs ``read -> model.fit ->
model.apply -> write`` on a thread pool (fuse.py:295-319,396-408).  This module keeps that call surface -- the
``process()`` signature and the three configuration dict factories -- on in-memory same-grid rasters:

* ``auto_block_shape`` / ``block_pairs`` reproduce the reference partition exactly (halve the longer side until a
  block fits ``max_block_mem``, overlap = ceil(kernel/2)), because ``gain-blk-offset`` results depend on it
  (its normalisation statistics are taken over each in-block, kernel_model.py:216-229);
* every block is one fused fit+apply kernel launch through the C ABI; worker threads check streams out of the
  context pool so H2D / kernel / D2H of different blocks overlap (ctypes releases the GIL);
* blocks are independent: they shard round-robin over the GPUs of a node (one context per device inside one process)
  and/or over ranks (one process per GPU), with no data-path collective.

GeoTIFF reading/writing, band matching and re-projection stay outside (GDAL IO).
"""
import math
import os
import threading
from concurrent.futures import ThreadPoolExecutor, as_completed
from typing import Dict, Iterable, Iterator, List, NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np

from homonim_amd import _hk, utils
from homonim_amd.enums import Model, ProcCrs
from homonim_amd.errors import BlockSizeError, ConfigWarning, IoError
from homonim_amd.geo import Affine, CRS, Window
from homonim_amd.kernel_model import KernelModel, RefSpaceModel, SrcSpaceModel
from homonim_amd.raster_array import RasterArray


class BlockPair(NamedTuple):
    """ Matching block windows of a source / reference pair (homonim/raster_pair.py:45-58). """
    band_i: int
    src_in_block: Window   # overlapping window that is read
    ref_in_block: Window
    src_out_block: Window  # non-overlapping window that is written
    ref_out_block: Window
    outer: bool            # the in-block touches the image boundary


def auto_block_shape(shape: Tuple[int, int], max_block_mem: float = math.inf, dtype_size: int = 4) -> Tuple[int, int]:
    """
    Block (rows, cols) for an image of ``shape``: keep halving the longer side (rows on ties) until a float32 block
    fits ``max_block_mem`` megabytes (2**20 bytes); <= 0 or inf = whole image (homonim/raster_pair.py:227-269 with
    equal source / reference resolutions, i.e. mem_scale 1).
    """
    limit = math.inf if (max_block_mem is None or max_block_mem <= 0) else float(max_block_mem) * 2 ** 20
    rows, cols = float(shape[0]), float(shape[1])
    while rows * cols * dtype_size > limit:
        if rows >= cols:
            rows /= 2
        else:
            cols /= 2
    if rows < 1 or cols < 1:
        raise BlockSizeError("The auto block shape is smaller than a pixel.  Increase 'max_block_mem'.")
    block = (int(math.ceil(rows)), int(math.ceil(cols)))
    if (block[0] < 256 or block[1] < 256) and (block[0] < shape[0] or block[1] < shape[1]):
        import warnings
        warnings.warn(
            f'The auto block shape is small: {block}.  Increase `max_block_mem` to improve processing times.',
            category=ConfigWarning
        )
    return block


def block_pairs(shape: Tuple[int, int], n_bands: int, overlap: Tuple[int, int] = (0, 0),
                max_block_mem: float = math.inf) -> Iterator[BlockPair]:
    """
    (band x block) work items in the reference's order: bands outermost, then blocks row-major
    (homonim/raster_pair.py:379-428).  In-blocks overlap their neighbours by ``overlap`` on every side and are clipped
    to the image; out-blocks tile the image exactly.
    """
    height, width = int(shape[0]), int(shape[1])
    ov_r, ov_c = int(overlap[0]), int(overlap[1])
    blk_r, blk_c = auto_block_shape((height, width), max_block_mem)
    if blk_r <= ov_r or blk_c <= ov_c:
        raise BlockSizeError('The auto block shape is smaller than the overlap.  Increase `max_block_mem`.')
    row_starts = range(-ov_r, height - ov_r, blk_r)
    col_starts = range(-ov_c, width - ov_c, blk_c)
    for band_i in range(n_bands):
        for r0 in row_starts:
            for c0 in col_starts:
                in_r0, in_c0 = max(r0, 0), max(c0, 0)
                in_r1, in_c1 = min(r0 + blk_r + 2 * ov_r, height), min(c0 + blk_c + 2 * ov_c, width)
                out_r0, out_c0 = max(r0 + ov_r, 0), max(c0 + ov_c, 0)
                out_r1, out_c1 = min(r0 + blk_r + ov_r, height), min(c0 + blk_c + ov_c, width)
                outer = in_r0 <= 0 or in_c0 <= 0 or in_r1 >= height or in_c1 >= width
                win_in = Window(in_c0, in_r0, in_c1 - in_c0, in_r1 - in_r0)
                win_out = Window(out_c0, out_r0, out_c1 - out_c0, out_r1 - out_r0)
                yield BlockPair(band_i, win_in, win_in, win_out, win_out, outer)


def shard(items: Sequence, index: int, count: int) -> List:
    """ The work items of shard ``index`` of ``count`` (round-robin): shards are disjoint and cover ``items``. """
    if count < 1 or not (0 <= index < count):
        raise ValueError(f'bad shard {index} of {count}')
    return list(items[index::count])


class RasterFuse:
    """
    Correct a source raster to surface reflectance by fusion with a reference raster on the same grid.

    src, ref : float32 arrays (bands, height, width) or (height, width); ``RasterArray`` instances are accepted too.
    The other constructor arguments mirror homonim.RasterFuse / RasterPairReader where they make sense in memory.
    """

    create_model_config = staticmethod(KernelModel.create_config)

    def __init__(self, src: Union[np.ndarray, RasterArray], ref: Union[np.ndarray, RasterArray],
                 src_nodata: Optional[float] = float('nan'), ref_nodata: Optional[float] = float('nan'),
                 proc_crs: ProcCrs = ProcCrs.auto, crs: Optional[CRS] = None, transform: Optional[Affine] = None):
        if isinstance(src, RasterArray):
            src, src_nodata, crs, transform = src.array, src.nodata, src.crs, src.transform
        if isinstance(ref, RasterArray):
            ref, ref_nodata = ref.array, ref.nodata
        src = np.asarray(src)
        ref = np.asarray(ref)
        if src.ndim == 2:
            src = src[None]
        if ref.ndim == 2:
            ref = ref[None]
        if src.ndim != 3 or ref.ndim != 3:
            raise ValueError('`src` and `ref` must be 2-D or 3-D (bands first) arrays')
        if src.shape[-2:] != ref.shape[-2:]:
            raise NotImplementedError('source and reference must share a grid (re-projection: SURVEY.md section 8f)')
        if ref.shape[0] < src.shape[0]:
            raise ValueError('`ref` has fewer bands than `src`')
        self._src, self._ref = src, ref
        self._src_nodata, self._ref_nodata = src_nodata, ref_nodata
        self._crs = crs or CRS()
        self._transform = transform or Affine.identity()
        # equal resolutions: auto resolves to the reference grid (homonim/raster_pair.py:193-224)
        self._proc_crs = ProcCrs.ref if ProcCrs(proc_crs) == ProcCrs.auto else ProcCrs(proc_crs)
        self._closed = False
        self._write_lock = threading.Lock()

    # -- context manager parity with the reference (files there, nothing to open here) --------------------------------
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._closed = True

    @property
    def proc_crs(self) -> ProcCrs:
        return self._proc_crs

    @property
    def src_bands(self) -> Tuple[int, ...]:
        return tuple(range(1, self._src.shape[0] + 1))

    @property
    def shape(self) -> Tuple[int, int]:
        return tuple(self._src.shape[-2:])

    # -- configuration factories (homonim/fuse.py:89-149) -------------------------------------------------------------
    @staticmethod
    def create_block_config(threads: int = 0, max_block_mem: float = 100) -> Dict:
        return dict(threads=utils.validate_threads(threads), max_block_mem=max_block_mem)

    @staticmethod
    def create_out_profile(driver: str = 'GTiff', dtype: str = RasterArray.default_dtype,
                           nodata: Optional[float] = RasterArray.default_nodata,
                           creation_options: Optional[Dict] = None) -> Dict:
        creation_options = creation_options or dict(
            tiled=True, blockxsize=512, blockysize=512, compress='deflate', interleave='band',
            photometric='minisblack', bigtiff='if_safer'
        )
        return dict(driver=driver, dtype=dtype, nodata=nodata, creation_options=creation_options)

    @staticmethod
    def create_device_config(devices: Optional[Sequence[int]] = None, streams: int = 4, rank: int = 0,
                             world_size: int = 1) -> Dict:
        """ (this package only) GPUs of this process, streams per GPU, and this process's shard of the block list. """
        return dict(devices=None if devices is None else list(devices), streams=int(streams), rank=int(rank),
                    world_size=int(world_size))

    def block_pairs(self, overlap: Tuple[int, int] = (0, 0), max_block_mem: float = math.inf) -> Iterable[BlockPair]:
        return block_pairs(self.shape, self._src.shape[0], overlap, max_block_mem)

    # -- the block loop -----------------------------------------------------------------------------------------------
    def _read(self, bp: BlockPair) -> Tuple[RasterArray, RasterArray]:
        rs, cs = bp.src_in_block.toslices()
        tf = self._transform * Affine.translation(bp.src_in_block.col_off, bp.src_in_block.row_off)
        src_ra = RasterArray(self._src[bp.band_i][rs, cs], self._crs, tf, nodata=self._src_nodata)
        ref_ra = RasterArray(self._ref[bp.band_i][rs, cs], self._crs, tf, nodata=self._ref_nodata)
        return src_ra, ref_ra

    @staticmethod
    def _crop(bp: BlockPair) -> Tuple[slice, slice]:
        """ slices of the out-block inside the in-block (the halo crop of homonim/raster_array.py:478-491) """
        r0 = bp.src_out_block.row_off - bp.src_in_block.row_off
        c0 = bp.src_out_block.col_off - bp.src_in_block.col_off
        return slice(r0, r0 + bp.src_out_block.height), slice(c0, c0 + bp.src_out_block.width)

    def _process_block(self, bp: BlockPair, model: KernelModel, corr: np.ndarray, params: Optional[np.ndarray],
                       out_nodata: Optional[float]):
        """ read -> fused fit+apply (+ output dtype conversion) on the GPU -> write (homonim/fuse.py:295-319) """
        src_ra, ref_ra = self._read(bp)
        corr_ra, param_ra = model.fit_apply(src_ra, ref_ra, want_params=params is not None, out_dtype=corr.dtype.name,
                                            out_nodata=out_nodata)
        crop = self._crop(bp)
        rs, cs = bp.src_out_block.toslices()
        corr[bp.band_i][rs, cs] = corr_ra.array[crop]
        if params is not None:
            n_src = self._src.shape[0]
            for pi in range(param_ra.count):  # band order of the reference's parameter file (fuse.py:316)
                params[pi * n_src + bp.band_i][rs, cs] = param_ra.array[pi][crop]

    def process(self, corr_filename: Optional[Union[str, os.PathLike]] = None, model: Model = KernelModel.default_model,
                kernel_shape: Tuple[int, int] = KernelModel.default_kernel_shape,
                param_filename: Optional[Union[str, os.PathLike, bool]] = None, build_ovw: bool = True,
                overwrite: bool = False, model_config: Optional[Dict] = None, out_profile: Optional[Dict] = None,
                block_config: Optional[Dict] = None, device_config: Optional[Dict] = None):
        """
        Same arguments as homonim.RasterFuse.process (fuse.py:321-332) plus ``device_config``.  Returns
        ``(corrected, params)``: float32 arrays (bands, H, W) and (n_param_bands * bands, H, W) or None.  When
        ``corr_filename`` / ``param_filename`` are paths the arrays are also saved with ``numpy.save`` (GeoTIFF output
        belongs to the GDAL side of the reference; ``build_ovw`` and ``out_profile['driver'|'creation_options']`` are
        accepted and ignored).  With ``world_size > 1`` only this rank's blocks are filled in (others stay nodata).
        """
        if self._closed:
            raise IoError('The raster pair has been closed')
        model_type = Model(model)
        overlap = utils.overlap_for_kernel(kernel_shape)
        model_config = RasterFuse.create_model_config(**(model_config or {}))
        block_config = RasterFuse.create_block_config(**(block_config or {}))
        out_profile = RasterFuse.create_out_profile(**(out_profile or {}))
        device_config = RasterFuse.create_device_config(**(device_config or {}))
        want_params = param_filename is not None and param_filename is not False
        for fn in (corr_filename, param_filename):
            if isinstance(fn, (str, os.PathLike)) and os.path.exists(fn) and not overwrite:
                raise FileExistsError(f"Corrected / parameter file exists and won't be overwritten: {fn}")

        model_cls = SrcSpaceModel if self._proc_crs == ProcCrs.src else RefSpaceModel
        devices = device_config['devices']
        if devices is None:
            devices = [int(os.environ.get('HOMONIM_AMD_DEVICE', os.environ.get('LOCAL_RANK', '0')))]
        models = []
        for dev in devices:
            m = model_cls(model_type, kernel_shape, find_r2=want_params, **model_config)
            m.context = _hk.get_context(dev, device_config['streams'])  # cached per process
            models.append(m)

        n_src = self._src.shape[0]
        nodata = out_profile['nodata']
        out_dtype = np.dtype(out_profile['dtype'])
        if out_dtype.name not in _hk.DTYPE_CODES:
            raise ValueError(f"unsupported output dtype '{out_profile['dtype']}'")
        fill = (np.nan if out_dtype.kind == 'f' else 0) if nodata is None else nodata
        corr = np.full((n_src, *self.shape), fill, dtype=out_dtype)
        n_param = 3 if models[0]._emit_r2 else 2
        params = np.full((n_param * n_src, *self.shape), np.nan, dtype=np.float32) if want_params else None

        blocks = list(self.block_pairs(overlap=overlap, max_block_mem=block_config['max_block_mem']))
        blocks = shard(blocks, device_config['rank'], device_config['world_size'])
        if block_config['threads'] == 1 and len(models) == 1:
            for bp in blocks:
                self._process_block(bp, models[0], corr, params, nodata)
        else:
            workers = max(block_config['threads'], len(models))
            with ThreadPoolExecutor(max_workers=workers) as ex:
                futures = [
                    ex.submit(self._process_block, bp, models[i % len(models)], corr, params, nodata)
                    for i, bp in enumerate(blocks)
                ]
                for f in as_completed(futures):
                    f.result()  # re-raise worker exceptions (fuse.py:404-408)

        if isinstance(corr_filename, (str, os.PathLike)):
            np.save(corr_filename, corr)
        if want_params and isinstance(param_filename, (str, os.PathLike)):
            np.save(param_filename, params)
        return corr, params


def convert_dtype(array: np.ndarray, dtype: str, nodata: Optional[float]) -> np.ndarray:
    """
    Host-side statement of ``RasterArray._convert_array_dtype`` (homonim/raster_array.py:353-387): round half-to-even and
    clip for integer types, NaN (the internal nodata) -> ``nodata``.  ``RasterFuse.process`` does this conversion on the
    device (hk_convert.hip); this helper documents the semantics and serves host-side callers.
    """
    invalid = np.isnan(array)
    out = array
    if np.issubdtype(np.dtype(dtype), np.integer):
        info = np.iinfo(dtype)
        out = np.clip(np.round(array.astype(np.promote_types(array.dtype, dtype))), info.min, info.max)
    with np.errstate(invalid='ignore', over='ignore'):
        out = out.astype(dtype, copy=(out is array))
    if nodata is not None:
        out[invalid] = nodata
    return out
