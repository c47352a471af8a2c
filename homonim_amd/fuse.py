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

Rasters may also be given / written as GeoTIFF paths (homonim_amd/tiff.py: the classic-TIFF subset the reference's
rasters use); band matching by wavelength and re-projection BETWEEN CRSs stay outside (GDAL).
"""
import math
import os
import threading
from concurrent.futures import ThreadPoolExecutor, as_completed
from typing import Dict, Iterable, Iterator, List, NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np

from homonim_amd import _hk, utils
from homonim_amd.enums import Model, ProcCrs
from homonim_amd.errors import BlockSizeError, ConfigWarning, DeviceError, IoError
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


class Grid(NamedTuple):
    """ A north-up raster grid: geo-transform + size (what the block partition needs of a rasterio dataset). """
    transform: Affine
    height: int
    width: int

    @property
    def res(self) -> Tuple[float, float]:
        return abs(self.transform.a), abs(self.transform.e)

    def window_bounds(self, win: Window) -> Tuple[float, float, float, float]:
        """ (left, bottom, right, top) of a pixel window. """
        t = self.transform
        return (t.c + win.col_off * t.a, t.f + (win.row_off + win.height) * t.e, t.c + (win.col_off + win.width) * t.a,
                t.f + win.row_off * t.e)

    @property
    def bounds(self) -> Tuple[float, float, float, float]:
        return self.window_bounds(Window(0, 0, self.width, self.height))

    def window(self, left: float, bottom: float, right: float, top: float) -> Window:
        """ The (fractional) pixel window of a bounding box. """
        t = self.transform
        return Window((left - t.c) / t.a, (top - t.f) / t.e, (right - left) / t.a, (bottom - top) / t.e)


def expand_window_to_grid(win: Window, expand_pixels: Tuple[int, int] = (0, 0)) -> Window:
    """ The smallest whole-pixel window containing ``win`` grown by ``expand_pixels`` (rows, cols)
    (homonim/utils.py:59-81). """
    col_off = math.floor(win.col_off - expand_pixels[1])
    row_off = math.floor(win.row_off - expand_pixels[0])
    col_frac = (win.col_off - expand_pixels[1]) - col_off
    row_frac = (win.row_off - expand_pixels[0]) - row_off
    width = math.ceil(win.width + 2 * expand_pixels[1] + col_frac)
    height = math.ceil(win.height + 2 * expand_pixels[0] + row_frac)
    return Window(int(col_off), int(row_off), int(width), int(height))


def round_window_to_grid(win: Window) -> Window:
    """ ``win`` with its edges rounded (half to even) to whole pixels (homonim/utils.py:84-101). """
    r0, r1 = round(win.row_off), round(win.row_off + win.height)
    c0, c1 = round(win.col_off), round(win.col_off + win.width)
    return Window(int(c0), int(r0), int(c1 - c0), int(r1 - r0))


def pair_windows(src: Grid, ref: Grid) -> Tuple[Window, Window]:
    """ (src_win, ref_win): the reference window covering the source, and the (possibly boundless) source window covering
    that -- so that blocks re-project between the two grids without losing data (homonim/raster_pair.py:289-291). """
    ref_win = expand_window_to_grid(ref.window(*src.bounds))
    src_win = expand_window_to_grid(src.window(*ref.window_bounds(ref_win)))
    return src_win, ref_win


def resolve_proc_crs(src: Grid, ref: Grid, proc_crs: ProcCrs = ProcCrs.auto) -> ProcCrs:
    """ auto -> the grid with the larger pixels (homonim/raster_pair.py:193-224). """
    src_smaller = src.res[0] * src.res[1] <= ref.res[0] * ref.res[1]
    proc_crs = ProcCrs(proc_crs)
    if proc_crs == ProcCrs.auto:
        return ProcCrs.ref if src_smaller else ProcCrs.src
    if (proc_crs == ProcCrs.src and src_smaller) or (proc_crs == ProcCrs.ref and not src_smaller):
        import warnings
        rec = ProcCrs.ref if src_smaller else ProcCrs.src
        warnings.warn(f'proc_crs={rec} is recommended for these pixel sizes.', category=ConfigWarning)
    return proc_crs


def block_pairs_multires(src: Grid, ref: Grid, proc_crs: ProcCrs, n_bands: int, overlap: Tuple[int, int] = (0, 0),
                         max_block_mem: float = math.inf) -> Iterator[BlockPair]:
    """
    The reference's block partition for a source / reference pair on DIFFERENT grids of one CRS
    (homonim/raster_pair.py:227-269,342-428): blocks are cut on the processing grid, then mapped to whole-pixel windows
    of the other grid (in-blocks expanded, out-blocks rounded).  Source in-blocks may reach beyond the image (boundless).
    """
    src_win, ref_win = pair_windows(src, ref)
    proc_is_ref = ProcCrs(proc_crs) == ProcCrs.ref
    proc, other = (ref, src) if proc_is_ref else (src, ref)
    proc_win = ref_win if proc_is_ref else src_win
    src_area, ref_area = src.res[0] * src.res[1], ref.res[0] * ref.res[1]
    if proc_is_ref:
        mem_scale = src_area / ref_area if ref_area > src_area else 1.
    else:
        mem_scale = 1. if ref_area > src_area else ref_area / src_area
    mem = math.inf if (max_block_mem is None or max_block_mem <= 0) else max_block_mem * mem_scale
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', ConfigWarning)  # the small-block warning is re-issued below in source pixels
        blk_r, blk_c = auto_block_shape((proc_win.height, proc_win.width), mem)
    if (blk_r / mem_scale < 256 or blk_c / mem_scale < 256) and (blk_r < proc_win.height or blk_c < proc_win.width):
        warnings.warn(f'The auto block shape is small: {(blk_r, blk_c)}.  Increase `max_block_mem` to improve '
                      'processing times.', category=ConfigWarning)
    ov_r, ov_c = int(overlap[0]), int(overlap[1])
    if blk_r <= ov_r or blk_c <= ov_c:
        raise BlockSizeError('The auto block shape is smaller than the overlap.  Increase `max_block_mem`.')
    p_r0, p_c0 = proc_win.row_off, proc_win.col_off
    p_r1, p_c1 = p_r0 + proc_win.height, p_c0 + proc_win.width
    for band_i in range(n_bands):
        for r in range(p_r0 - ov_r, p_r1 - ov_r, blk_r):
            for c in range(p_c0 - ov_c, p_c1 - ov_c, blk_c):
                in_r0, in_c0 = max(r, p_r0), max(c, p_c0)
                in_r1, in_c1 = min(r + blk_r + 2 * ov_r, p_r1), min(c + blk_c + 2 * ov_c, p_c1)
                out_r0, out_c0 = max(r + ov_r, p_r0), max(c + ov_c, p_c0)
                out_r1, out_c1 = min(r + blk_r + ov_r, p_r1), min(c + blk_c + ov_c, p_c1)
                outer = in_r0 <= p_r0 or in_c0 <= p_c0 or in_r1 >= p_r1 or in_c1 >= p_c1
                proc_in = Window(in_c0, in_r0, in_c1 - in_c0, in_r1 - in_r0)
                proc_out = Window(out_c0, out_r0, out_c1 - out_c0, out_r1 - out_r0)
                other_in = expand_window_to_grid(other.window(*proc.window_bounds(proc_in)))
                other_out = round_window_to_grid(other.window(*proc.window_bounds(proc_out)))
                if proc_is_ref:
                    yield BlockPair(band_i, other_in, proc_in, other_out, proc_out, outer)
                else:
                    yield BlockPair(band_i, proc_in, other_in, proc_out, other_out, outer)


def _as_affine(transform, name: str) -> Optional[Affine]:
    if transform is None or isinstance(transform, Affine):
        return transform
    try:
        vals = [float(v) for v in transform]
    except TypeError:
        raise ValueError(f'`{name}` must be an Affine or a sequence of its six coefficients') from None
    if len(vals) not in (6, 9):
        raise ValueError(f'`{name}` must be an Affine or a sequence of its six coefficients')
    return Affine(*vals[:6])


def north_up(array: np.ndarray, transform: Optional[Affine]):
    """ (array, transform) with rows running north to south and columns west to east: a raster whose geo-transform has a
    positive row step (south-up) or a negative column step is flipped along that axis -- what the reference's
    ``utils.same_orientation_crs`` obtains by re-projecting it through a WarpedVRT.  Rotated grids are not handled. """
    if transform is None:
        return array, transform
    if transform.b or transform.d:
        raise NotImplementedError('rotated / sheared rasters are not built (GDAL warp)')
    a, b, c, d, e, f = transform[:6]
    if e > 0:
        array = np.ascontiguousarray(array[..., ::-1, :])
        f, e = f + e * array.shape[-2], -e
    if a < 0:
        array = np.ascontiguousarray(array[..., ::-1])
        c, a = c + a * array.shape[-1], -a
    return array, Affine(a, b, c, d, e, f)


def shard(items: Sequence, index: int, count: int, contiguous: bool = False) -> List:
    """ The work items of shard ``index`` of ``count``: round-robin, or -- ``contiguous`` -- consecutive runs of (almost)
    equal length, which keeps a shard inside as few bands as possible (the block list is band-major,
    homonim/raster_pair.py:381-389).  Either way the shards are disjoint and cover ``items``. """
    if count < 1 or not (0 <= index < count):
        raise ValueError(f'bad shard {index} of {count}')
    if contiguous:
        n = len(items)
        return list(items[index * n // count:(index + 1) * n // count])
    return list(items[index::count])


class RasterFuse:
    """
    Correct a source raster to surface reflectance by fusion with a reference raster of the same CRS.

    src, ref : float32 arrays (bands, height, width) or (height, width), ``RasterArray`` instances, or GeoTIFF paths
    (nodata, CRS and geo-transform then come from the files; bands are paired in file order).
    The other constructor arguments mirror homonim.RasterFuse / RasterPairReader where they make sense in memory.
    """

    create_model_config = staticmethod(KernelModel.create_config)

    def __init__(self, src: Union[np.ndarray, RasterArray], ref: Union[np.ndarray, RasterArray],
                 src_nodata: Optional[float] = float('nan'), ref_nodata: Optional[float] = float('nan'),
                 proc_crs: ProcCrs = ProcCrs.auto, crs: Optional[CRS] = None, transform: Optional[Affine] = None,
                 ref_transform: Optional[Affine] = None):
        """
        ``transform`` / ``ref_transform``: geo-transforms of the source / reference rasters (same CRS, north-up).  When
        they (or the shapes) differ, blocks are cut on the processing grid and re-sampled on the device as the
        reference does through GDAL (RefSpaceModel / SrcSpaceModel); the reference must cover the source.
        """
        self._src_filename = self._ref_filename = None
        if isinstance(src, (str, os.PathLike)):  # a GeoTIFF, as homonim.RasterFuse(src_filename, ref_filename, ...)
            from homonim_amd.tiff import read_tiff
            self._src_filename = os.fspath(src)
            tif = read_tiff(src)
            src, src_nodata, crs, transform = tif.array, tif.nodata, tif.crs, tif.transform
        if isinstance(ref, (str, os.PathLike)):
            from homonim_amd.tiff import read_tiff
            self._ref_filename = os.fspath(ref)
            tif = read_tiff(ref)
            if crs is not None and tif.crs != crs:
                raise NotImplementedError('source and reference CRSs differ: re-projection between CRSs is not built (GDAL warp)')
            ref, ref_nodata, ref_transform = tif.array, tif.nodata, tif.transform
        if isinstance(src, RasterArray):
            src, src_nodata, crs, transform = src.array, src.nodata, src.crs, src.transform
        if isinstance(ref, RasterArray):
            ref, ref_nodata, ref_transform = ref.array, ref.nodata, ref.transform
        src = np.asarray(src)
        ref = np.asarray(ref)
        if src.ndim == 2:
            src = src[None]
        if ref.ndim == 2:
            ref = ref[None]
        if src.ndim != 3 or ref.ndim != 3:
            raise ValueError('`src` and `ref` must be 2-D or 3-D (bands first) arrays')
        # transforms may come as plain 6-tuples (a, b, c, d, e, f)
        transform = _as_affine(transform, 'transform')
        ref_transform = _as_affine(ref_transform, 'ref_transform')
        # south-up (or column-mirrored) rasters are brought to north-up first, as the reference does through a WarpedVRT
        # (homonim/utils.py:190-209; for an axis-aligned raster that re-projection is the flip); outputs are north-up
        src, transform = north_up(src, transform)
        ref, ref_transform = north_up(ref, ref_transform)
        if ref.shape[0] < src.shape[0]:
            raise ValueError('`ref` has fewer bands than `src`')
        self._src, self._ref = src, ref
        self._src_nodata, self._ref_nodata = src_nodata, ref_nodata
        self._crs = crs or CRS()
        self._transform = transform or Affine.identity()
        self._ref_transform = ref_transform or self._transform
        self._src_grid = Grid(self._transform, *src.shape[-2:])
        self._ref_grid = Grid(self._ref_transform, *ref.shape[-2:])
        self._same_grid = self._src_grid == self._ref_grid
        if not self._same_grid:
            l, b, r, t = self._src_grid.bounds
            rl, rb, rr, rt = self._ref_grid.bounds
            if l < rl or r > rr or b < rb or t > rt:  # homonim/raster_pair.py:93-94
                from homonim_amd.errors import ImageContentError
                raise ImageContentError('Reference extent does not cover source image')
        # auto resolves to the grid with the larger pixels, the reference grid on ties (homonim/raster_pair.py:193-224)
        self._proc_crs = resolve_proc_crs(self._src_grid, self._ref_grid, proc_crs)
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
                             world_size: int = 1, contiguous: bool = False, pin: bool = True,
                             separate_contexts: bool = False) -> Dict:
        """ (this package only) GPUs of this process, streams per GPU, this process's shard of the block list
        (round-robin, or ``contiguous`` runs), whether every entry of ``devices`` gets a context of its own even when a
        device is listed twice, and ``pin``: True (default) -- the rasters are registered in place for the duration of the
        block loop (``hk_host_register``) and copied directly: 10-17 % more end-to-end throughput than through the staging
        ring (profiles/r04_streamed_host.txt); a raster that cannot be registered (``ulimit -l``, foreign memory) and every
        array the library did not page-lock itself travel through the library's own page-locked staging ring, so the runtime
        is never handed pageable memory either way (the registered path ran clean as the first process of five fresh
        leases, profiles/r04_abort_followup.txt); False -- everything through the staging ring: the GPU never touches
        caller-allocated pages. """
        return dict(devices=None if devices is None else list(devices), streams=int(streams), rank=int(rank),
                    world_size=int(world_size), contiguous=bool(contiguous), pin=bool(pin),
                    separate_contexts=bool(separate_contexts))

    def block_pairs(self, overlap: Tuple[int, int] = (0, 0), max_block_mem: float = math.inf) -> Iterable[BlockPair]:
        if self._same_grid:
            return block_pairs(self.shape, self._src.shape[0], overlap, max_block_mem)
        return block_pairs_multires(self._src_grid, self._ref_grid, self._proc_crs, self._src.shape[0], overlap,
                                    max_block_mem)

    @staticmethod
    def _read_boundless(band: np.ndarray, nodata: Optional[float], win: Window) -> Tuple[np.ndarray, Optional[float]]:
        """ A window of a band that may reach beyond it, filled with nodata outside (RasterArray.from_rio_dataset,
        homonim/raster_array.py:172-188: NaN when the raster has no nodata value). """
        h, w = band.shape
        r0, c0 = max(win.row_off, 0), max(win.col_off, 0)
        r1, c1 = min(win.row_off + win.height, h), min(win.col_off + win.width, w)
        inside = r0 == win.row_off and c0 == win.col_off and r1 == win.row_off + win.height and c1 == win.col_off + win.width
        if inside:
            return band[r0:r1, c0:c1], nodata
        fill = float('nan') if nodata is None else nodata
        dtype = np.float32 if (nodata is None or (isinstance(fill, float) and math.isnan(fill))) else band.dtype
        out = np.full((win.height, win.width), fill, dtype=dtype)
        if r1 > r0 and c1 > c0:
            out[r0 - win.row_off:r1 - win.row_off, c0 - win.col_off:c1 - win.col_off] = band[r0:r1, c0:c1]
        return out, fill

    def _process_block_multires(self, bp: BlockPair, model: KernelModel, corr: np.ndarray, params: Optional[np.ndarray],
                                out_nodata: Optional[float]):
        """ read (boundless) -> fit on the processing grid -> apply on the source grid -> write the out-block
        (homonim/fuse.py:295-319 with RefSpaceModel / SrcSpaceModel doing the re-sampling on the device) """
        s_arr, s_nd = self._read_boundless(self._src[bp.band_i], self._src_nodata, bp.src_in_block)
        r_arr, r_nd = self._read_boundless(self._ref[bp.band_i], self._ref_nodata, bp.ref_in_block)
        src_tf = self._transform * Affine.translation(bp.src_in_block.col_off, bp.src_in_block.row_off)
        ref_tf = self._ref_transform * Affine.translation(bp.ref_in_block.col_off, bp.ref_in_block.row_off)
        fused = isinstance(model, RefSpaceModel)  # one upload / download per block, everything else stays in HBM
        if fused:
            src_ra = RasterArray(s_arr, self._crs, src_tf, nodata=s_nd)
            ref_ra = RasterArray(r_arr, self._crs, ref_tf, nodata=r_nd)
            corr_ra, param_ra = model.fit_apply(src_ra, ref_ra, want_params=params is not None, out_dtype=corr.dtype.name,
                                                out_nodata=out_nodata)
        else:
            src_ra = RasterArray(np.ascontiguousarray(s_arr, dtype=np.float32), self._crs, src_tf, nodata=s_nd)
            ref_ra = RasterArray(np.ascontiguousarray(r_arr, dtype=np.float32), self._crs, ref_tf, nodata=r_nd)
            param_ra = model.fit(src_ra, ref_ra)
            corr_ra = model.apply(src_ra, param_ra)

        def put(dst_plane, block_arr, in_win, out_win):
            h, w = dst_plane.shape
            r0, c0 = max(out_win.row_off, 0), max(out_win.col_off, 0)
            r1, c1 = min(out_win.row_off + out_win.height, h), min(out_win.col_off + out_win.width, w)
            if r1 > r0 and c1 > c0:
                dst_plane[r0:r1, c0:c1] = block_arr[r0 - in_win.row_off:r1 - in_win.row_off,
                                                    c0 - in_win.col_off:c1 - in_win.col_off]

        block = corr_ra.array
        if not fused and (corr.dtype != np.float32 or
                          not (out_nodata is None or (isinstance(out_nodata, float) and math.isnan(out_nodata)))):
            block = convert_dtype(block, corr.dtype.name, out_nodata)
        put(corr[bp.band_i], block, bp.src_in_block, bp.src_out_block)
        if params is not None:
            n_src = self._src.shape[0]
            p_in, p_out = (bp.ref_in_block, bp.ref_out_block) if self._proc_crs == ProcCrs.ref else \
                (bp.src_in_block, bp.src_out_block)
            for pi in range(param_ra.count):
                put(params[pi * n_src + bp.band_i], param_ra.array[pi], p_in, p_out)

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
        crop = self._crop(bp)
        rs, cs = bp.src_out_block.toslices()
        if model.fuses_into(src_ra, ref_ra):
            # the out-block goes straight from the device into the corrected / parameter rasters (no block-sized
            # temporaries, no host-side crop copy; asynchronous when the rasters are page-locked)
            n_src = self._src.shape[0]
            params_dst = params[bp.band_i::n_src][:, rs, cs] if params is not None else None
            window = (crop[0].start, crop[1].start, bp.src_out_block.height, bp.src_out_block.width)
            model.fit_apply_into(src_ra, ref_ra, window, corr[bp.band_i][rs, cs], params_dst, out_nodata=out_nodata)
            return
        corr_ra, param_ra = model.fit_apply(src_ra, ref_ra, want_params=params is not None, out_dtype=corr.dtype.name,
                                            out_nodata=out_nodata)
        corr[bp.band_i][rs, cs] = corr_ra.array[crop]
        if params is not None:
            n_src = self._src.shape[0]
            for pi in range(param_ra.count):  # band order of the reference's parameter file (fuse.py:316)
                params[pi * n_src + bp.band_i][rs, cs] = param_ra.array[pi][crop]

    def process(self, corr_filename: Optional[Union[str, os.PathLike]] = None, model: Model = KernelModel.default_model,
                kernel_shape: Tuple[int, int] = KernelModel.default_kernel_shape,
                param_filename: Optional[Union[str, os.PathLike, bool]] = None, build_ovw: bool = True,
                overwrite: bool = False, model_config: Optional[Dict] = None, out_profile: Optional[Dict] = None,
                block_config: Optional[Dict] = None, device_config: Optional[Dict] = None,
                corr_out: Optional[np.ndarray] = None):
        """
        Same arguments as homonim.RasterFuse.process (fuse.py:321-332) plus ``device_config``.  Returns
        ``(corrected, params)``: float32 arrays (bands, H, W) and (n_param_bands * bands, H, W) or None.  When
        ``corr_filename`` / ``param_filename`` are paths the arrays are also written there: ``.tif`` as a tiled DEFLATE
        GeoTIFF with the reference's FUSE_* provenance tags (homonim_amd/tiff.py), anything else with ``numpy.save``
        (``build_ovw`` and ``out_profile['driver'|'creation_options']`` are accepted and ignored).  With ``world_size > 1`` only this rank's blocks are filled in (others stay nodata).
        ``corr_out``: a caller-owned corrected raster (bands, H, W) of the output dtype to fill instead of a new one, e.g.
        page-locked memory of a pipeline that processes many rasters (it is NOT pre-filled with nodata).
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
        models, own_contexts, seen = [], [], set()
        for dev in devices:
            m = model_cls(model_type, kernel_shape, find_r2=want_params, **model_config)
            if device_config['separate_contexts'] and dev in seen:
                m.context = _hk.Context(dev, device_config['streams'])  # a second, independent context on this device
                own_contexts.append(m.context)
            else:
                m.context = _hk.get_context(dev, device_config['streams'])  # cached per process
            seen.add(dev)
            models.append(m)

        n_src = self._src.shape[0]
        nodata = out_profile['nodata']
        out_dtype = np.dtype(out_profile['dtype'])
        if out_dtype.name not in _hk.DTYPE_CODES:
            raise ValueError(f"unsupported output dtype '{out_profile['dtype']}'")
        fill = (np.nan if out_dtype.kind == 'f' else 0) if nodata is None else nodata
        if corr_out is not None:
            if corr_out.shape != (n_src, *self.shape) or corr_out.dtype != out_dtype or not corr_out.flags['C_CONTIGUOUS']:
                raise ValueError('`corr_out` must be a C-contiguous (bands, height, width) array of the output dtype')
            corr = corr_out
        else:
            corr = np.full((n_src, *self.shape), fill, dtype=out_dtype)
        n_param = 3 if models[0]._emit_r2 else 2
        # parameters live on the processing grid (homonim/fuse.py:254-293): the reference's when proc_crs == ref
        param_shape = self._ref.shape[-2:] if (self._proc_crs == ProcCrs.ref and not self._same_grid) else self.shape
        params = np.full((n_param * n_src, *param_shape), np.nan, dtype=np.float32) if want_params else None
        process_block = self._process_block if self._same_grid else self._process_block_multires

        blocks = list(self.block_pairs(overlap=overlap, max_block_mem=block_config['max_block_mem']))
        blocks = shard(blocks, device_config['rank'], device_config['world_size'], device_config['contiguous'])
        # on request: page-lock the rasters in place for the duration of the block loop (direct copies); by default pageable
        # rasters go through the context's pinned staging ring (hk_api.hip stage_h2d / stage_d2h) block by block
        pinned = self._pin(models[0].context, [self._src, self._ref, corr, params]) if device_config['pin'] else []
        try:
            self._run_blocks(blocks, models, process_block, corr, params, nodata, block_config)
        finally:
            for arr in pinned:
                try:
                    models[0].context.unpin(arr)
                except Exception:
                    pass
            for c in own_contexts:
                c.close()

        if isinstance(corr_filename, (str, os.PathLike)) or (want_params and isinstance(param_filename, (str, os.PathLike))):
            # provenance tags of homonim/fuse.py:193-207
            meta = dict(FUSE_SRC_FILE=os.path.basename(self._src_filename or 'memory'),
                        FUSE_REF_FILE=os.path.basename(self._ref_filename or 'memory'), FUSE_PROC_CRS=self._proc_crs.name,
                        FUSE_MODEL=model_type.name, FUSE_KERNEL_SHAPE=tuple(kernel_shape),
                        **{f'FUSE_{k.upper()}': getattr(v, 'name', v) for k, v in model_config.items()})
            if isinstance(corr_filename, (str, os.PathLike)):
                self._save(corr_filename, corr, self._transform, nodata, meta)
            if want_params and isinstance(param_filename, (str, os.PathLike)):
                param_tf = self._ref_transform if (self._proc_crs == ProcCrs.ref and not self._same_grid) else self._transform
                self._save(param_filename, params, param_tf, float('nan'), meta)
        return corr, params

    @staticmethod
    def _pin(ctx, arrays) -> List[np.ndarray]:
        """ hipHostRegister the arrays that can be (C-contiguous, not page-locked yet); returns those that were. """
        done = []
        for arr in arrays:
            if arr is None or not isinstance(arr, np.ndarray) or not arr.flags['C_CONTIGUOUS'] or arr.nbytes == 0:
                continue
            try:
                ctx.pin(arr)   # counted per address range (Context.pin): concurrent process() calls share a registration
                done.append(arr)
            except DeviceError:
                pass  # not lockable (ulimit -l, foreign memory): the copies of this array stay synchronous
        return done

    @staticmethod
    def _run_blocks(blocks, models, process_block, corr, params, nodata, block_config):
        if block_config['threads'] == 1 and len(models) == 1:
            for bp in blocks:
                process_block(bp, models[0], corr, params, nodata)
        else:
            workers = max(block_config['threads'], len(models))
            several = len({m.context.device for m in models}) > 1

            def on_device(bp, model):
                # several GPUs in one process: the worker that packs this block into a GPU's pinned staging ring runs on that GPU's
                # NUMA node (topology.bind_current_thread; one process per GPU binds the whole process instead: dist / bench.py)
                if several:
                    from homonim_amd import topology
                    topology.bind_current_thread(model.context.device)
                return process_block(bp, model, corr, params, nodata)

            with ThreadPoolExecutor(max_workers=workers) as ex:
                futures = [ex.submit(on_device, bp, models[i % len(models)]) for i, bp in enumerate(blocks)]
                for f in as_completed(futures):
                    f.result()  # re-raise worker exceptions (fuse.py:404-408)

    def _save(self, filename, array: np.ndarray, transform: Affine, nodata, metadata: Dict):
        """ ``.tif`` / ``.tiff``: tiled DEFLATE GeoTIFF like the reference's default output profile (no overviews);
        anything else: ``numpy.save``. """
        if os.fspath(filename).lower().endswith(('.tif', '.tiff')):
            from homonim_amd.tiff import write_tiff
            write_tiff(filename, array, transform, self._crs, nodata, {k: str(v) for k, v in metadata.items()})
        else:
            np.save(filename, array)


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
