"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden-vector generator; runs ONLY in the build container.

Executes the reference's own ``homonim/kernel_model.py`` (read in place from /root/reference, never copied) on
seeded inputs and writes small ``tests/golden/*.npz`` fixtures (inputs + expected outputs).  ``/root/reference``
does not exist on the GPU box, so nothing but the ``.npz`` data travels.

Third-party modules the reference imports but that are absent here (no network):

* ``cv2`` (opencv-python-headless>=4.5, pyproject.toml:8) -- stand-in below implementing exactly the four entry
  points the hot path calls (kernel_model.py:155-184,256-258,331-341,407-408) with OpenCV's documented semantics:
  zero-border, centre-anchored, un-normalised window sums accumulated in float64; ``boxFilter(ddepth=-1)`` returns
  the input depth, ``sqrBoxFilter(ddepth=-1)`` returns float64 for float input (OpenCV's own rule; confirmed
  against the reference's PARAM GeoTIFF, which is reproduced bit-for-bit only with that depth).
  This is a RESTATEMENT of the published algorithm, so the goldens pin the reference's *Python* (mask logic, numpy
  operation order, dtype promotion) bit-for-bit, and the OpenCV boundary only by definition.  DESIGN.md says so.
* ``rasterio`` (>=1.1, pyproject.toml:7) -- placeholder names only (``Affine``, ``CRS``, ``Window``, enums, ...);
  base-class ``KernelModel.fit/apply`` never calls into GDAL.  ``fillnodata`` is the identity and COUNTS calls that
  had holes inside the valid area, so no golden silently depends on in-painting.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.npz + manifest.json)
"""
import importlib.util
import json
import os
import sys
import types
from collections import namedtuple
from enum import Enum, IntEnum

import numpy as np

REF_ROOT = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')

sys.path.insert(0, REPO)


# ----------------------------------------------------------------------------------------------------------------------
# stand-in cv2
def _win_sum(src, ksize, square):
    kw, kh = int(ksize[0]), int(ksize[1])
    rh, rw = kh // 2, kw // 2
    h, w = src.shape
    x = src.astype(np.float64)
    if square:
        x = x * x
    pad = np.zeros((h + 2 * rh, w + 2 * rw), np.float64)
    pad[rh:rh + h, rw:rw + w] = x
    acc = np.zeros((h, w), np.float64)
    for dy in range(kh):  # direct 2-D accumulation (deliberately a different order to oracle_np.box_sum)
        for dx in range(kw):
            acc += pad[dy:dy + h, dx:dx + w]
    if square:
        # cv::sqrBoxFilter: ``if (ddepth < 0) ddepth = sdepth < CV_32F ? CV_32F : CV_64F`` -- float32 in, float64 OUT.
        # Pinned by the reference's own PARAM GeoTIFF: only this depth reproduces it (tests/test_oracle_golden.py).
        return acc
    out_dtype = src.dtype if src.dtype in (np.float32, np.float64) else np.float64
    return acc.astype(out_dtype)


def _make_cv2():
    cv2 = types.ModuleType('cv2')
    cv2.BORDER_CONSTANT = 0
    cv2.MORPH_RECT = 0

    def boxFilter(src, ddepth, ksize, normalize=True, borderType=None, **kw):
        assert ddepth == -1 and normalize is False and borderType == cv2.BORDER_CONSTANT
        return _win_sum(src, ksize, False)

    def sqrBoxFilter(src, ddepth, ksize, normalize=True, borderType=None, **kw):
        assert ddepth == -1 and normalize is False and borderType == cv2.BORDER_CONSTANT
        return _win_sum(src, ksize, True)

    def getStructuringElement(shape, ksize):
        return np.ones((int(ksize[1]), int(ksize[0])), np.uint8)

    def erode(src, se, borderType=None, borderValue=0, **kw):
        kh, kw_ = se.shape
        rh, rw = kh // 2, kw_ // 2
        h, w = src.shape
        pad = np.full((h + 2 * rh, w + 2 * rw), borderValue, src.dtype)
        pad[rh:rh + h, rw:rw + w] = src
        out = np.full((h, w), np.iinfo(src.dtype).max, src.dtype)
        for dy in range(kh):
            for dx in range(kw_):
                out = np.minimum(out, pad[dy:dy + h, dx:dx + w])
        return out

    cv2.boxFilter, cv2.sqrBoxFilter = boxFilter, sqrBoxFilter
    cv2.getStructuringElement, cv2.erode = getStructuringElement, erode
    return cv2


# ----------------------------------------------------------------------------------------------------------------------
# stand-in rasterio (names only)
FILL_CALLS = dict(total=0, with_holes=0)


def _make_rasterio():
    rio = types.ModuleType('rasterio')

    class Affine(namedtuple('Affine', 'a b c d e f')):
        pass

    class CRS:
        def __init__(self, name='EPSG:3857'):
            self.name = name

        def __eq__(self, other):
            return isinstance(other, CRS) and other.name == self.name

        def __hash__(self):
            return hash(self.name)

    class Window(namedtuple('Window', 'col_off row_off width height')):
        def toranges(self):
            return ((self.row_off, self.row_off + self.height), (self.col_off, self.col_off + self.width))

    class _Placeholder:
        pass

    Resampling = IntEnum(
        'Resampling', 'nearest bilinear cubic cubic_spline lanczos average mode gauss max min med q1 q3 sum rms',
        start=0
    )
    MaskFlags = Enum('MaskFlags', 'all_valid per_dataset alpha nodata')
    ColorInterp = Enum('ColorInterp', 'undefined gray palette red green blue alpha')

    def fillnodata(image, mask=None, **kw):
        FILL_CALLS['total'] += 1
        if np.any(~np.asarray(mask, bool) & ~np.isnan(image)):
            FILL_CALLS['with_holes'] += 1
        return image

    def _sub(name, **attrs):
        m = types.ModuleType(f'rasterio.{name}')
        for k, v in attrs.items():
            setattr(m, k, v)
        setattr(rio, name, m)
        sys.modules[f'rasterio.{name}'] = m
        return m

    rio.Affine = Affine
    rio.DatasetReader = _Placeholder
    rio.uint8, rio.float32 = 'uint8', 'float32'
    _sub('enums', Resampling=Resampling, MaskFlags=MaskFlags, ColorInterp=ColorInterp)
    _sub('fill', fillnodata=fillnodata)
    _sub('crs', CRS=CRS)
    _sub('transform', TransformMethodsMixin=type('TransformMethodsMixin', (), {}))
    _sub(
        'windows', Window=Window, WindowMethodsMixin=type('WindowMethodsMixin', (), {}),
        transform=lambda window, transform: transform
    )
    def reproject(source, destination=None, src_crs=None, src_transform=None, src_nodata=None, dst_crs=None,
                  dst_transform=None, dst_nodata=None, num_threads=1, resampling=None, **kw):
        # GDAL warp stand-in for IDENTICAL grids only: average / nearest re-sampling onto the same grid is the identity
        # (source nodata -> destination nodata).  Anything else is outside what these goldens may depend on.
        assert dst_transform is None or tuple(dst_transform) == tuple(src_transform), 'stand-in reproject: grids differ'
        assert destination.shape[-2:] == source.shape[-2:], 'stand-in reproject: shapes differ'
        assert resampling in (Resampling.average, Resampling.nearest), resampling
        src = np.asarray(source)
        out = src.astype(destination.dtype)
        if src_nodata is not None and dst_nodata is not None:
            bad = np.isnan(src) if (isinstance(src_nodata, float) and np.isnan(src_nodata)) else (src == src_nodata)
            out[bad] = dst_nodata
        destination[...] = out
        return destination, src_transform

    _sub('warp', reproject=reproject, Resampling=Resampling)
    _sub('errors', NotGeoreferencedWarning=type('NotGeoreferencedWarning', (UserWarning, ), {}))
    _sub('vrt', WarpedVRT=_Placeholder)
    _sub('io', DatasetWriter=_Placeholder)
    _sub('dtypes', can_cast_dtype=lambda v, d: True)
    _sub('drivers', raster_driver_extensions=lambda: {'tif': 'GTiff'})
    return rio


def load_reference():
    """ Load homonim/{enums,errors,utils,raster_array,kernel_model}.py by path under a synthetic package. """
    sys.modules['cv2'] = _make_cv2()
    sys.modules['rasterio'] = _make_rasterio()
    pkg = types.ModuleType('homonim')
    pkg.__path__ = [os.path.join(REF_ROOT, 'homonim')]
    sys.modules['homonim'] = pkg
    mods = {}
    for name in ('enums', 'errors', 'utils', 'raster_array', 'kernel_model'):
        spec = importlib.util.spec_from_file_location(f'homonim.{name}', os.path.join(REF_ROOT, 'homonim', f'{name}.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f'homonim.{name}'] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
        mods[name] = mod
    return mods


# ----------------------------------------------------------------------------------------------------------------------
def make_inputs(h, w, seed, variant):
    """ Seeded src/ref pairs with the nodata settings SURVEY.md section 8c lists. """
    from oracle.oracle_np import synth_pair
    if variant == 'nan_frame_holes':
        src, ref = synth_pair(h, w, seed, 'frame+holes')
        return src, np.nan, ref, np.nan
    if variant == 'no_nodata':
        src, ref = synth_pair(h, w, seed, 'none')
        return src, None, ref, None
    if variant == 'nan_nodata_clean':
        src, ref = synth_pair(h, w, seed, 'none')
        return src, np.nan, ref, np.nan
    if variant == 'numeric_nodata':
        # byte-like DN data with nodata 0 in src (a 2-px frame + scattered zeros) and nodata None in ref
        rng = np.random.default_rng(1000 + seed)
        src = np.round(rng.uniform(1, 255, (h, w))).astype(np.float32)
        ref = np.round(0.8 * src + 20 + rng.normal(0, 4, (h, w))).astype(np.float32)
        src[:2], src[-2:], src[:, :2], src[:, -2:] = 0, 0, 0, 0
        src[rng.random((h, w)) < 0.01] = 0
        return src, 0., ref, None
    if variant == 'dn_like':
        src, ref = synth_pair(h, w, seed, 'frame+holes', dn_like=True)
        return src, np.nan, ref, np.nan
    if variant == 'all_masked':
        src, ref = synth_pair(h, w, seed, 'none')
        src[:] = np.nan
        return src, np.nan, ref, np.nan
    raise ValueError(variant)


def run_reference(mods, model, kernel_shape, find_r2, thresh, src, src_nodata, ref, ref_nodata):
    """ KernelModel.fit on copies, then KernelModel.apply on the ORIGINAL source (the reference never re-uses the
    fitted src_ra: SURVEY.md section 8b 'Trap'). """
    import warnings
    km, ra_mod, rio = mods['kernel_model'], mods['raster_array'], sys.modules['rasterio']
    crs, tf = sys.modules['rasterio.crs'].CRS(), rio.Affine(1., 0., 0., 0., -1., 0.)
    RasterArray = ra_mod.RasterArray
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model_obj = km.KernelModel(model, kernel_shape, find_r2=find_r2, r2_inpaint_thresh=thresh)
        src_fit = RasterArray(src.copy(), crs, tf, nodata=src_nodata)
        ref_fit = RasterArray(ref.copy(), crs, tf, nodata=ref_nodata)
        norm = None
        if model == 'gain-blk-offset':
            norm = km.KernelModel._fit_block_norm(
                RasterArray(src.copy(), crs, tf, nodata=src_nodata), RasterArray(ref.copy(), crs, tf, nodata=ref_nodata)
            )
        with np.errstate(all='ignore'):
            param_ra = model_obj.fit(src_fit, ref_fit)
            src_apply = RasterArray(src.copy(), crs, tf, nodata=src_nodata)
            corr_ra = model_obj.apply(src_apply, param_ra)
    return param_ra.array, corr_ra.array, norm


CASES = []
for _model in ('gain', 'gain-blk-offset', 'gain-offset'):
    for _k in ((1, 1), (3, 3), (5, 5), (5, 7), (15, 15)):
        if _model == 'gain-offset' and _k == (1, 1):
            continue  # ValueError by design (utils.py:123-125)
        for _r2 in (False, True):
            # the full nodata cross only at the default 5x5 kernel; other shapes on the NaN frame+holes input
            variants = ('nan_frame_holes', 'no_nodata', 'numeric_nodata') if _k == (5, 5) else ('nan_frame_holes', )
            for _variant in variants:
                threshes = (None, 0.25) if _model == 'gain-offset' else (0.25, )
                for _thresh in threshes:
                    CASES.append(dict(model=_model, k=_k, find_r2=_r2, variant=_variant, thresh=_thresh))
# a few extra: harsher DN-like distribution, all-masked block, taller-than-wide kernel
CASES += [
    dict(model='gain-offset', k=(5, 5), find_r2=True, variant='dn_like', thresh=None),
    dict(model='gain-blk-offset', k=(5, 5), find_r2=True, variant='dn_like', thresh=0.25),
    dict(model='gain-blk-offset', k=(5, 5), find_r2=False, variant='all_masked', thresh=0.25),
    dict(model='gain', k=(7, 3), find_r2=True, variant='nan_frame_holes', thresh=0.25),
    dict(model='gain-offset', k=(9, 9), find_r2=True, variant='nan_frame_holes', thresh=0.25),
]


class _FakeDataset:
    """ North-up dataset (pixel size `res`, upper-left corner `origin`): just enough of rasterio.DatasetReader for
    RasterPairReader.open()'s window set-up and block_pairs(). """
    closed = False

    def __init__(self, height, width, res=(1., 1.), origin=(0., 0.)):
        self.height, self.width, self.res, self.origin = height, width, res, origin
        self.shape = (height, width)

    @property
    def bounds(self):
        Window = sys.modules['rasterio.windows'].Window
        return self.window_bounds(Window(0, 0, self.width, self.height))

    def window_bounds(self, win):
        (rx, ry), (x0, y0) = self.res, self.origin
        return (x0 + win.col_off * rx, y0 - (win.row_off + win.height) * ry, x0 + (win.col_off + win.width) * rx,
                y0 - win.row_off * ry)

    def window(self, left, bottom, right, top):
        (rx, ry), (x0, y0) = self.res, self.origin
        Window = sys.modules['rasterio.windows'].Window
        return Window((left - x0) / rx, (y0 - top) / ry, (right - left) / rx, (top - bottom) / ry)


def gen_block_goldens():
    """ The reference's own block partition (raster_pair.py:227-269,342-428) on same-grid rasters -> JSON table. """
    import warnings
    spec = importlib.util.spec_from_file_location('homonim.raster_pair', os.path.join(REF_ROOT, 'homonim', 'raster_pair.py'))
    rp = importlib.util.module_from_spec(spec)
    sys.modules['homonim.raster_pair'] = rp
    spec.loader.exec_module(rp)
    Window = sys.modules['rasterio.windows'].Window
    ProcCrs = sys.modules['homonim.enums'].ProcCrs
    table = []
    cases = [  # (height, width, n_bands, kernel_shape, max_block_mem MB)
        (16384, 16384, 4, (5, 5), 100), (16384, 16384, 8, (15, 15), 100), (8192, 8192, 4, (5, 5), 100),
        (4096, 4096, 4, (5, 5), 100), (1421, 805, 3, (5, 5), 1), (1000, 3000, 2, (3, 7), 2), (777, 333, 1, (9, 9), 0.5),
        (2048, 2048, 1, (1, 1), 4), (100, 100, 2, (5, 5), float('inf')),
    ]
    for (h, w, nb, k, mem) in cases:
        for proc in (ProcCrs.ref, ProcCrs.src):
            rdr = object.__new__(rp.RasterPairReader)
            rdr._src_im, rdr._ref_im = _FakeDataset(h, w), _FakeDataset(h, w)
            rdr._src_bands = tuple(range(1, nb + 1))
            rdr._ref_bands = tuple(range(1, nb + 1))
            rdr._src_win = rdr._ref_win = Window(0, 0, w, h)
            rdr._proc_crs = proc
            rdr._src_filename = rdr._ref_filename = 'fake.tif'
            overlap = sys.modules['homonim.utils'].overlap_for_kernel(k)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                block_shape = rdr._auto_block_shape(max_block_mem=mem)
                bps = list(rdr.block_pairs(overlap=overlap, max_block_mem=mem))
            wins = [[bp.band_i, *[int(v) for v in bp.src_in_block], *[int(v) for v in bp.src_out_block],
                     *[int(v) for v in bp.ref_in_block], *[int(v) for v in bp.ref_out_block], bool(bp.outer)] for bp in bps]
            table.append(dict(height=h, width=w, n_bands=nb, kernel_shape=list(k), overlap=[int(o) for o in overlap],
                              max_block_mem=(None if mem == float('inf') else mem), proc_crs=proc.value,
                              block_shape=[int(b) for b in block_shape], n_blocks=len(wins),
                              # [band_i, src_in(col_off,row_off,w,h), src_out(...), ref_in(...), ref_out(...), outer]
                              block_pairs=wins if len(wins) <= 70 else wins[:35] + wins[-35:]))
    with open(os.path.join(GOLDEN_DIR, 'block_pairs.json'), 'w') as f:
        json.dump(dict(note='reference raster_pair.py block partition on same-grid rasters; for > 70 blocks only the '
                            'first and last 35 are stored', cases=table), f)
    print(f'block goldens: {len(table)} cases')

    # pairs of different resolution / origin: RasterPairReader.open()'s windows (raster_pair.py:289-291) + block_pairs
    utils = sys.modules['homonim.utils']
    table = []
    multi = [  # (src h, w, res, origin), (ref h, w, res, origin), bands, kernel, max_block_mem
        ((1421, 805, 5., (-57129.449, -3723906.806)), (1361, 797, 10., (-60370., -3722700.)), 3, (5, 5), 1),
        ((560, 560, 5., (-56519.449, -3726056.806)), (289, 289, 10., (-56560., -3726010.)), 3, (5, 5), 0.25),
        ((400, 600, 0.5, (5., -5.)), (200, 300, 1., (5., -5.)), 2, (3, 3), 0.1),
        ((300, 500, 30., (1000., 9000.)), (900, 1500, 10., (1000., 9000.)), 1, (5, 5), 0.5),
        ((333, 517, 2., (11.3, 77.7)), (120, 200, 7., (-13., 101.)), 2, (7, 3), 0.2),
    ]
    for (sh, sw, sr, so), (rh, rw, rr, ro), nb, k, mem in multi:
        for proc in (ProcCrs.ref, ProcCrs.src):
            rdr = object.__new__(rp.RasterPairReader)
            rdr._src_im, rdr._ref_im = _FakeDataset(sh, sw, (sr, sr), so), _FakeDataset(rh, rw, (rr, rr), ro)
            rdr._src_bands = rdr._ref_bands = tuple(range(1, nb + 1))
            rdr._ref_win = utils.expand_window_to_grid(rdr._ref_im.window(*rdr._src_im.bounds))
            rdr._src_win = utils.expand_window_to_grid(rdr._src_im.window(*rdr._ref_im.window_bounds(rdr._ref_win)))
            rdr._proc_crs = proc
            rdr._src_filename = rdr._ref_filename = 'fake.tif'
            overlap = utils.overlap_for_kernel(k)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                block_shape = rdr._auto_block_shape(max_block_mem=mem)
                bps = list(rdr.block_pairs(overlap=overlap, max_block_mem=mem))
            wins = [[bp.band_i, *[int(v) for v in bp.src_in_block], *[int(v) for v in bp.src_out_block],
                     *[int(v) for v in bp.ref_in_block], *[int(v) for v in bp.ref_out_block], bool(bp.outer)] for bp in bps]
            table.append(dict(src=dict(height=sh, width=sw, res=sr, origin=list(so)),
                              ref=dict(height=rh, width=rw, res=rr, origin=list(ro)), n_bands=nb, kernel_shape=list(k),
                              overlap=[int(o) for o in overlap], max_block_mem=mem, proc_crs=proc.value,
                              src_win=[int(v) for v in rdr._src_win], ref_win=[int(v) for v in rdr._ref_win],
                              block_shape=[int(b) for b in block_shape], n_blocks=len(wins),
                              block_pairs=wins if len(wins) <= 60 else wins[:30] + wins[-30:]))
    with open(os.path.join(GOLDEN_DIR, 'block_pairs_multires.json'), 'w') as f:
        json.dump(dict(note='reference raster_pair.py windows + block partition for source / reference pairs of different '
                            'resolution and origin (north-up, same CRS); > 60 blocks: first and last 30', cases=table), f)
    print(f'multi-resolution block goldens: {len(table)} cases')


def gen_convert_goldens(mods):
    """ RasterArray._convert_array_dtype (raster_array.py:353-387) of the REFERENCE on a float32 block with NaN nodata. """
    rio = sys.modules['rasterio']
    RasterArray = mods['raster_array'].RasterArray
    rng = np.random.default_rng(99)
    a = rng.uniform(-40, 300, (24, 40)).astype(np.float32)
    a[0, :8] = [-0.5, 0.5, 1.5, 2.5, 254.5, 255.5, -1e9, 1e9]
    a[1, :8] = [65534.5, 65535.5, 70000, -32768.5, -32769, 32767.5, 3e9, -3e9]
    a[2, :6] = [4294967295., 4294967040., 2147483520., -2147483648., 16777217., 0.49999997]
    a[rng.random(a.shape) < 0.1] = np.nan
    out = dict(input=a)
    crs, tf = sys.modules['rasterio.crs'].CRS(), rio.Affine(1., 0., 0., 0., -1., 0.)
    for dtype, nodata in (('uint8', 0), ('uint8', 255), ('uint16', 0), ('int16', -9999), ('int32', -2147483648),
                          ('uint32', 0), ('float32', -9999.0), ('float32', float('nan')), ('float64', -1.5)):
        ra = RasterArray(a.copy(), crs, tf, nodata=float('nan'))
        with np.errstate(all='ignore'):
            res = ra._convert_array_dtype(dtype, nodata=nodata)
        key = f"{dtype}_{'nan' if (isinstance(nodata, float) and np.isnan(nodata)) else nodata}"
        out[key] = res
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'convert_dtype.npz'), **out)
    print('convert goldens:', [k for k in out if k != 'input'])


def gen_mask_partial_goldens(mods):
    """ RefSpaceModel.apply / SrcSpaceModel.fit with mask_partial=True on a SHARED grid (kernel_model.py:375-409,
    484-503, 516-535), run through the reference's own classes. """
    import warnings
    km, ra_mod, rio = mods['kernel_model'], mods['raster_array'], sys.modules['rasterio']
    # re-bind the freshly installed stand-in (raster_array imported `reproject` by name)
    ra_mod.reproject = sys.modules['rasterio.warp'].reproject
    RasterArray = ra_mod.RasterArray
    crs, tf = sys.modules['rasterio.crs'].CRS(), rio.Affine(1., 0., 0., 0., -1., 0.)
    src, snd, ref, rnd = make_inputs(40, 56, 3, 'nan_frame_holes')
    out = dict(src=src, ref=ref)
    cases = []
    for model, k in (('gain-blk-offset', (1, 1)), ('gain-blk-offset', (3, 3)), ('gain-blk-offset', (3, 5)),
                     ('gain-blk-offset', (5, 5)), ('gain-offset', (5, 5)), ('gain', (7, 3))):
        for space in ('ref', 'src'):
            cls = km.RefSpaceModel if space == 'ref' else km.SrcSpaceModel
            with warnings.catch_warnings(), np.errstate(all='ignore'):
                warnings.simplefilter('ignore')
                m = cls(model, k, find_r2=True, mask_partial=True, r2_inpaint_thresh=None)
                src_ra = RasterArray(src.copy(), crs, tf, nodata=snd)
                ref_ra = RasterArray(ref.copy(), crs, tf, nodata=rnd)
                param_ra = m.fit(src_ra, ref_ra)
                corr_ra = m.apply(RasterArray(src.copy(), crs, tf, nodata=snd), param_ra)
            name = f'{space}_{model}_{k[0]}x{k[1]}'
            out[name + '_params'] = param_ra.array
            out[name + '_corr'] = corr_ra.array
            cases.append(dict(name=name, space=space, model=model, kernel_shape=list(k)))
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'mask_partial.npz'), **out)
    with open(os.path.join(GOLDEN_DIR, 'mask_partial.json'), 'w') as f:
        json.dump(cases, f)
    print(f'mask_partial goldens: {len(cases)} cases')


def gen_compare_goldens(mods):
    """ The reference's own RasterCompare.process (compare.py:212-278) on in-memory same-grid band stacks: the block
    reader of RasterPairReader is replaced by array slicing (no GDAL), everything from get_block_sums to
    _get_image_stats is the reference's code.  -> tests/golden/compare.npz (inputs + per-band r2 / RMSE / rRMSE / N). """
    import warnings
    for name in ('matched_pair', 'compare'):
        spec = importlib.util.spec_from_file_location(f'homonim.{name}', os.path.join(REF_ROOT, 'homonim', f'{name}.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f'homonim.{name}'] = mod
        spec.loader.exec_module(mod)
    cmp_mod, ra_mod, rio = sys.modules['homonim.compare'], mods['raster_array'], sys.modules['rasterio']
    ra_mod.reproject = sys.modules['rasterio.warp'].reproject
    Window = sys.modules['rasterio.windows'].Window
    ProcCrs = sys.modules['homonim.enums'].ProcCrs
    crs = sys.modules['rasterio.crs'].CRS()

    class ArrayCompare(cmp_mod.RasterCompare):
        def __init__(self, src, src_nodata, ref, ref_nodata, proc_crs):
            nb, h, w = src.shape
            self._src, self._ref, self._snd, self._rnd = src, ref, src_nodata, ref_nodata
            self._src_im, self._ref_im = _FakeDataset(h, w), _FakeDataset(h, w)
            self._src_im.descriptions = self._ref_im.descriptions = (None, ) * nb
            self._src_bands = self._ref_bands = tuple(range(1, nb + 1))
            self._src_win = self._ref_win = Window(0, 0, w, h)
            self._proc_crs = proc_crs
            self._src_filename = self._ref_filename = 'memory.tif'

        def read(self, bp):
            def cut(stack, nodata, win):
                (r0, r1), (c0, c1) = win.toranges()
                tf = rio.Affine(1., 0., float(c0), 0., -1., -float(r0))
                return ra_mod.RasterArray(stack[bp.band_i, r0:r1, c0:c1].copy(), crs, tf, nodata=nodata)
            return cut(self._src, self._snd, bp.src_in_block), cut(self._ref, self._rnd, bp.ref_in_block)

    arrays, cases = {}, []
    specs = [  # (name, h, w, bands, variant of make_inputs, proc_crs, max_block_mem MB)
        ('nan_holes', 300, 420, 3, 'nan_frame_holes', ProcCrs.ref, 0.1),
        ('no_nodata', 257, 511, 2, 'no_nodata', ProcCrs.src, 0.25),
        ('numeric_nodata', 300, 420, 2, 'numeric_nodata', ProcCrs.ref, 512),
    ]
    for (name, h, w, nb, variant, proc, mem) in specs:
        srcs, refs = [], []
        for b in range(nb):
            s_, snd, r_, rnd = make_inputs(h, w, 100 + 7 * b + len(cases), variant)
            srcs.append(s_), refs.append(r_)
        src, ref = np.stack(srcs), np.stack(refs)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            stats = ArrayCompare(src, snd, ref, rnd, proc).process(threads=1, max_block_mem=mem)
        arrays[f'{name}_src'], arrays[f'{name}_ref'] = src, ref
        keys = list(stats.keys())
        arrays[f'{name}_stats'] = np.array([[float(stats[k]['r2']), float(stats[k]['rmse']), float(stats[k]['rrmse']),
                                             float(stats[k]['n'])] for k in keys], dtype=np.float64)
        cases.append(dict(name=name, bands=keys, proc_crs=proc.value, max_block_mem=mem,
                          src_nodata=(None if snd is None else ('nan' if np.isnan(snd) else float(snd))),
                          ref_nodata=(None if rnd is None else ('nan' if np.isnan(rnd) else float(rnd)))))
    arrays['cases_json'] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'compare.npz'), **arrays)
    print(f'compare goldens: {len(cases)} cases', {c['name']: arrays[c['name'] + '_stats'][-1].tolist() for c in cases})


def main():
    mods = load_reference()
    gen_block_goldens()
    gen_compare_goldens(mods)
    gen_mask_partial_goldens(mods)
    gen_convert_goldens(mods)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    h, w = 36, 52
    manifest = dict(
        numpy=np.__version__, reference='leftfield-geospatial/homonim v0.4.3 (/root/reference)',
        shape=[h, w], note='cv2/rasterio are stand-ins (see oracle/gen_golden.py header); gain-blk-offset goldens are '
        'the NumPy>=2 flavour (float64 normalised source)', cases=[]
    )
    arrays = {}
    inputs_done = {}
    for ci, case in enumerate(CASES):
        seed = 7 + len(inputs_done)
        vkey = case['variant']
        if vkey not in inputs_done:
            src, snd, ref, rnd = make_inputs(h, w, seed, vkey)
            inputs_done[vkey] = (src, snd, ref, rnd)
            arrays[f'in_{vkey}_src'] = src
            arrays[f'in_{vkey}_ref'] = ref
        src, snd, ref, rnd = inputs_done[vkey]
        holes_before = FILL_CALLS['with_holes']
        params, corr, norm = run_reference(
            mods, case['model'], case['k'], case['find_r2'], case['thresh'], src, snd, ref, rnd
        )
        inpainted = FILL_CALLS['with_holes'] > holes_before
        name = f'c{ci:03d}'
        arrays[f'{name}_params'] = params
        arrays[f'{name}_corr'] = corr
        if norm is not None:
            arrays[f'{name}_norm'] = norm
        manifest['cases'].append(
            dict(
                name=name, model=case['model'], kernel_shape=list(case['k']), find_r2=case['find_r2'],
                r2_inpaint_thresh=case['thresh'], variant=vkey,
                src_nodata=(None if snd is None else ('nan' if np.isnan(snd) else float(snd))),
                ref_nodata=(None if rnd is None else ('nan' if np.isnan(rnd) else float(rnd))),
                inpaint_had_holes=bool(inpainted), param_dtype=str(params.dtype), corr_dtype=str(corr.dtype)
            )
        )
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'kernel_model_goldens.npz'), **arrays)

    # reference conftest arrays (tests/conftest.py:74-89) run through gain-offset 5x5 -- the known-answer behind
    # tests/data/parameter/*PARAM*.tif
    a100 = np.array(range(1, 201), dtype='float32').reshape(20, 10)
    a100[:, [0, -1]] = np.nan
    a100[[0, -1], :] = np.nan
    params, corr, _ = run_reference(mods, 'gain-offset', (5, 5), True, 0.25, a100, np.nan, a100.copy(), np.nan)
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'conftest_100cm_gain_offset_k5.npz'), src=a100, params=params, corr=corr)

    manifest['fillnodata_calls'] = FILL_CALLS
    with open(os.path.join(GOLDEN_DIR, 'manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1)
    n_holes = sum(c['inpaint_had_holes'] for c in manifest['cases'])
    print(f'{len(CASES)} cases written; fillnodata calls: {FILL_CALLS}; cases with real in-paint holes: {n_holes}')
    size = os.path.getsize(os.path.join(GOLDEN_DIR, 'kernel_model_goldens.npz'))
    print(f'kernel_model_goldens.npz: {size / 1e6:.2f} MB')


if __name__ == '__main__':
    main()
