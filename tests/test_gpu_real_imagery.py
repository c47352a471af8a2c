"""
BASELINE.json configs[0] in miniature, on the GPU: a crop of the reference's own test rasters (NGI 5 m aerial RGB, uint8,
nodata 0, vs Sentinel-2 10 m; grids offset by a fraction of a pixel -- tests/golden/real_imagery_crop.npz,
oracle/extract_real_imagery.py) through the full RefSpaceModel / SrcSpaceModel pipeline: average down-sampling, kernel
model fit, cubic-spline up-sampling of the parameters (or of the reference), apply.

`test_reference_rasters_fuse_then_compare` runs the full-size pair itself, file to file (tests/golden/rasters: the two
data files, read and written by homonim_amd/tiff.py), as the reference's integration test does.

GDAL is not available, so the acceptance bar is the reference's own integration criterion
(tests/integration.py:79-83): the corrected image agrees better with the reference than the source did -- r2 up,
RMSE and rRMSE down, per band -- plus the mask rules of :85-103.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from homonim_amd import Affine, CRS, Model, RasterArray, RefSpaceModel, Resampling, SrcSpaceModel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pair():
    g = np.load(os.path.join(GOLDEN_DIR, 'real_imagery_crop.npz'))
    src_tf, ref_tf = Affine(*g['src_transform']), Affine(*g['ref_transform'])
    return g['src'], src_tf, float(g['src_nodata']), g['ref'], ref_tf


def _compare(img_ra: RasterArray, ref_ra: RasterArray):
    """ RasterCompare in a nutshell (homonim/compare.py:232-256,142-186): bring the finer image to the reference grid
    with `average`, then r2 / RMSE / rRMSE over the jointly valid pixels. """
    ds = img_ra.reproject(**ref_ra.proj_profile, resampling=Resampling.average)
    m = ds.mask & ref_ra.mask
    x, y = ds.array[m].astype(np.float64), ref_ra.array[m].astype(np.float64)
    r2 = np.corrcoef(x, y)[0, 1] ** 2
    rmse = np.sqrt(np.mean((x - y) ** 2))
    return dict(r2=r2, rmse=rmse, rrmse=rmse / y.mean(), n=int(m.sum()))


@pytest.mark.parametrize('cls, model, kernel_shape, mask_partial', [
    (RefSpaceModel, Model.gain_blk_offset, (5, 5), False),
    (RefSpaceModel, Model.gain, (1, 1), False),
    (RefSpaceModel, Model.gain_offset, (15, 15), False),
    (RefSpaceModel, Model.gain_blk_offset, (5, 5), True),
    (SrcSpaceModel, Model.gain_blk_offset, (31, 31), False),
])
def test_correction_improves_agreement_with_reference(pair, cls, model, kernel_shape, mask_partial):
    src, src_tf, src_nodata, ref, ref_tf = pair
    crs = CRS('EPSG:32735')
    km = cls(model, kernel_shape, mask_partial=mask_partial)
    for band in range(src.shape[0]):
        src_ra = RasterArray(src[band].astype(np.float32), crs, src_tf, nodata=src_nodata)
        ref_ra = RasterArray(ref[band].astype(np.float32), crs, ref_tf, nodata=None)
        param_ra = km.fit(src_ra, ref_ra)
        corr_ra = km.apply(src_ra, param_ra)
        assert corr_ra.shape == src_ra.shape and corr_ra.transform == src_ra.transform
        before, after = _compare(src_ra, ref_ra), _compare(corr_ra, ref_ra)
        assert after['r2'] > before['r2'], (band, before, after)
        assert after['rmse'] < before['rmse'], (band, before, after)
        assert after['rrmse'] < before['rrmse'], (band, before, after)
        if not mask_partial:
            assert (corr_ra.mask == src_ra.mask).all()
        else:
            assert 0 < corr_ra.mask.sum() < src_ra.mask.sum() and src_ra.mask[corr_ra.mask].all()


@pytest.mark.parametrize('model, kernel_shape, proc_crs, threads', [
    ('gain-blk-offset', (5, 5), 'auto', 4), ('gain', (3, 3), 'ref', 1), ('gain-offset', (5, 5), 'src', 4),
])
def test_raster_fuse_process_across_resolutions(pair, model, kernel_shape, proc_crs, threads):
    """ RasterFuse.process on the 5 m / 10 m pair: the reference's block partition across the two grids, re-sampling
    and kernel models on the device per block; integration criteria per band + the mask rule. """
    import warnings
    from homonim_amd.fuse import RasterFuse
    src, src_tf, src_nodata, ref, ref_tf = pair
    crs = CRS('EPSG:32735')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        rf = RasterFuse(src, ref, src_nodata=src_nodata, ref_nodata=None, proc_crs=proc_crs, crs=crs, transform=src_tf,
                        ref_transform=ref_tf)
        corr, params = rf.process(None, model, kernel_shape, param_filename=True,
                                  block_config=dict(threads=threads, max_block_mem=0.3))
        n_blocks = len(list(rf.block_pairs(utils_overlap(kernel_shape), 0.3)))
    assert n_blocks >= 3 * 4 and corr.shape == src.shape and corr.dtype == np.float32
    assert params.shape[0] == (3 if model == 'gain-offset' else 3) * src.shape[0]
    for band in range(src.shape[0]):
        src_ra = RasterArray(src[band].astype(np.float32), crs, src_tf, nodata=src_nodata)
        ref_ra = RasterArray(ref[band].astype(np.float32), crs, ref_tf, nodata=None)
        corr_ra = RasterArray(corr[band], crs, src_tf)
        before, after = _compare(src_ra, ref_ra), _compare(corr_ra, ref_ra)
        assert after['r2'] > before['r2'] and after['rmse'] < before['rmse'] and after['rrmse'] < before['rrmse']
        assert (corr_ra.mask == src_ra.mask).all()
    if model == 'gain':
        # gain is block-partition invariant on one grid; across grids the reference's overlap of ceil(k/2) processing
        # pixels leaves the cubic-spline support at block seams / image edges slightly different from a whole-image pass
        # (inherent to its design), so the bulk of the pixels -- not all -- agree with one whole-image RefSpaceModel pass
        km = RefSpaceModel(Model.gain, kernel_shape)
        src_ra = RasterArray(src[0].astype(np.float32), crs, src_tf, nodata=src_nodata)
        ref_ra = RasterArray(ref[0].astype(np.float32), crs, ref_tf, nodata=None)
        whole = km.apply(src_ra, km.fit(src_ra, ref_ra)).array
        ok = ~np.isnan(whole) & ~np.isnan(corr[0])
        assert np.percentile(np.abs(whole[ok] - corr[0][ok]), 85) < 1e-3


@pytest.mark.parametrize('proc_crs, max_block_mem', [('ref', 512), ('ref', 0.3), ('src', 0.3)])
def test_raster_compare_across_resolutions(pair, proc_crs, max_block_mem):
    """ RasterCompare.process on the 5 m / 10 m pair (the reference's acceptance metric, compare.py:212-278): blocks
    cut on the processing grid, the other raster re-sampled onto it on the device, masked sums on the device.  In one
    block on the reference grid it must agree with the numpy statistics of `_compare`; block-wise the re-sampling supports
    differ slightly at the seams (as in the reference), and on the source grid the reference is up-sampled instead. """
    import warnings
    from homonim_amd.compare import RasterCompare
    src, src_tf, src_nodata, ref, ref_tf = pair
    crs = CRS('EPSG:32735')
    names = ['red', 'green', 'blue'][:src.shape[0]]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with RasterCompare(src.astype(np.float32), ref.astype(np.float32), src_nodata=src_nodata, ref_nodata=None,
                           proc_crs=proc_crs, crs=crs, transform=src_tf, ref_transform=ref_tf, band_names=names) as cmp:
            stats = cmp.process(threads=2, max_block_mem=max_block_mem)
    assert list(stats.keys()) == names + ['Mean']
    for band, name in enumerate(names):
        src_ra = RasterArray(src[band].astype(np.float32), crs, src_tf, nodata=src_nodata)
        ref_ra = RasterArray(ref[band].astype(np.float32), crs, ref_tf, nodata=None)
        exp, got = _compare(src_ra, ref_ra), stats[name]
        if proc_crs == 'ref' and max_block_mem == 512:
            # `_compare` averages the whole source raster, whose clamped kernel support reaches one reference column
            # beyond the source extent; the reference's boundless source window (nodata outside) does not
            assert 0 <= exp['n'] - got['n'] <= ref.shape[-1]
            for k in ('r2', 'rmse', 'rrmse'):
                assert got[k] == pytest.approx(exp[k], rel=2e-3), (name, k)
        else:
            assert 0 < got['r2'] <= 1 and got['n'] > 0
            if proc_crs == 'ref':
                assert abs(got['n'] - exp['n']) <= 0.01 * exp['n']
                for k in ('r2', 'rmse', 'rrmse'):
                    assert got[k] == pytest.approx(exp[k], rel=0.05), (name, k)
            else:
                # on the (finer) source grid: ~4x the pixels, and the 5 m detail the up-sampled reference cannot
                # explain lowers the correlation / raises the error somewhat
                assert got['n'] > 3 * exp['n']
                assert 0.8 * exp['r2'] < got['r2'] < exp['r2']
                assert exp['rmse'] < got['rmse'] < 1.25 * exp['rmse']
    assert stats['Mean']['r2'] == pytest.approx(np.mean([stats[n]['r2'] for n in names]), rel=1e-12)


def test_geotiff_in_geotiff_out(pair, tmp_path):
    """ The reference's file-level workflow (fuse.py:321-408, compare.py:212-278) on GeoTIFFs: RasterFuse(src.tif,
    ref.tif).process(corr.tif, ..., param_filename=param.tif) equals the in-memory run, the outputs carry the
    geo-referencing / nodata / FUSE_* tags, and RasterCompare on the files shows the improvement. """
    import warnings
    from homonim_amd.compare import RasterCompare
    from homonim_amd.fuse import RasterFuse
    from homonim_amd.tiff import read_tiff, write_tiff
    src, src_tf, src_nodata, ref, ref_tf = pair
    crs = CRS('EPSG:32735')
    src_path, ref_path = tmp_path / 'src.tif', tmp_path / 'ref.tif'
    corr_path, param_path = tmp_path / 'corr.tif', tmp_path / 'param.tif'
    write_tiff(src_path, src, src_tf, crs, src_nodata, tile=256)
    write_tiff(ref_path, ref, ref_tf, crs, None, tile=256)
    kw = dict(model='gain-blk-offset', kernel_shape=(5, 5), out_profile=dict(dtype='uint8', nodata=0),
              block_config=dict(threads=2, max_block_mem=0.3))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with RasterFuse(src_path, ref_path) as rf:
            assert rf.proc_crs.name == 'ref'
            corr, params = rf.process(corr_path, param_filename=param_path, **kw)
        with RasterFuse(src, ref, src_nodata=src_nodata, ref_nodata=None, crs=crs, transform=src_tf, ref_transform=ref_tf) as rf:
            corr_mem, params_mem = rf.process(None, param_filename=True, **kw)
    assert corr.dtype == np.uint8 and np.array_equal(corr, corr_mem) and np.array_equal(params, params_mem, equal_nan=True)
    out = read_tiff(corr_path)
    assert np.array_equal(out.array, corr) and out.nodata == 0 and out.transform == src_tf and out.crs == crs
    assert out.metadata['FUSE_MODEL'] == 'gain_blk_offset' and out.metadata['FUSE_PROC_CRS'] == 'ref'
    assert out.metadata['FUSE_SRC_FILE'] == 'src.tif' and out.metadata['FUSE_KERNEL_SHAPE'] == '(5, 5)'
    par = read_tiff(param_path)
    assert par.array.dtype == np.float32 and par.array.shape == (3 * src.shape[0], *ref.shape[-2:])
    assert np.array_equal(par.array, params, equal_nan=True) and np.isnan(par.nodata) and par.transform == ref_tf
    with pytest.raises(FileExistsError):
        with RasterFuse(src_path, ref_path) as rf:
            rf.process(corr_path, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with RasterCompare(src_path, ref_path) as cmp:
            before = cmp.process(threads=1)
        with RasterCompare(corr_path, ref_path) as cmp:
            after = cmp.process(threads=1)
    assert after['Mean']['r2'] > before['Mean']['r2'] and after['Mean']['rmse'] < before['Mean']['rmse']
    assert after['Mean']['rrmse'] < before['Mean']['rrmse'] and after['Mean']['n'] == before['Mean']['n']


RASTER_DIR = os.path.join(GOLDEN_DIR, 'rasters')  # two data files of the reference's own tests (tests/data)


@pytest.mark.parametrize('model, kernel_shape, proc_crs, mask_partial, exp_proc_crs, out_dtype', [
    ('gain-blk-offset', (5, 5), 'auto', False, 'ref', 'float32'),     # BASELINE.json configs[0]
    ('gain-offset', (15, 15), 'auto', False, 'ref', 'float32'), ('gain-offset', (31, 31), 'src', False, 'src', 'float32'),
    ('gain', (1, 1), 'auto', False, 'ref', 'float32'), ('gain-blk-offset', (5, 5), 'auto', True, 'ref', 'float32'),
    ('gain-blk-offset', (5, 5), 'auto', False, 'ref', 'uint8'),
])
def test_reference_rasters_fuse_then_compare(tmp_path, model, kernel_shape, proc_crs, mask_partial, exp_proc_crs, out_dtype):
    """ The reference's integration test (tests/integration.py:30-110) on its own rasters, file to file: the 5 m NGI
    aerial tile ngi_rgb_byte_1.tif (3 x 1421 x 805 uint8, nodata 0, 256-px tiles) fused with sentinel2_b432_byte.tif
    (10 m, RGB-interleaved strips, grid origin off by a fraction of a pixel), max_block_mem = 1 MB as there; the corrected
    GeoTIFF must agree better with the reference than the source did in every band (r2 up, RMSE and rRMSE down), and
    keep the source mask (or shrink it to one blob's worth of interior pixels with mask_partial). """
    import warnings
    from homonim_amd.compare import RasterCompare
    from homonim_amd.fuse import RasterFuse
    from homonim_amd.tiff import read_tiff
    src_path = os.path.join(RASTER_DIR, 'ngi_rgb_byte_1.tif')
    ref_path = os.path.join(RASTER_DIR, 'sentinel2_b432_byte.tif')
    corr_path = tmp_path / 'corr.tif'
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with RasterFuse(src_path, ref_path, proc_crs=proc_crs) as rf:
            assert rf.proc_crs.name == exp_proc_crs and rf.shape == (1421, 805)
            rf.process(corr_path, model, kernel_shape, model_config=dict(mask_partial=mask_partial),
                       out_profile=dict(dtype='uint8', nodata=0) if out_dtype == 'uint8' else None,
                       block_config=dict(threads=2, max_block_mem=1))
        with RasterCompare(src_path, ref_path, proc_crs=proc_crs) as cmp:
            src_res = cmp.process()
        with RasterCompare(corr_path, ref_path, proc_crs=proc_crs) as cmp:
            corr_res = cmp.process()
    for band, src_dict in src_res.items():
        corr_dict = corr_res[band]
        assert corr_dict['r2'] > src_dict['r2'], (band, src_dict, corr_dict)
        assert corr_dict['rmse'] < src_dict['rmse'], (band, src_dict, corr_dict)
        assert corr_dict['rrmse'] < src_dict['rrmse'], (band, src_dict, corr_dict)
    src, corr = read_tiff(src_path), read_tiff(corr_path)
    assert corr.array.dtype.name == out_dtype and corr.array.shape == src.array.shape
    assert corr.transform == src.transform and corr.crs == src.crs
    src_mask = (src.array != 0).any(axis=0)                                             # dataset masks
    if out_dtype == 'uint8':
        # (the reference's default output is float32 / NaN; in a uint8 file with nodata 0 a valid pixel that is corrected to
        # 0 in every band reads back as nodata -- in the reference too -- so the masks agree up to a handful of pixels)
        assert corr.nodata == 0
        corr_mask = (corr.array != 0).any(axis=0)
        assert not (corr_mask & ~src_mask).any() and (src_mask & ~corr_mask).sum() <= 1e-4 * src_mask.sum()
        assert 0 <= src_res['Mean']['n'] - corr_res['Mean']['n'] <= 1e-4 * src_res['Mean']['n']
        return
    assert np.isnan(corr.nodata)
    corr_mask = (~np.isnan(corr.array)).any(axis=0)
    if not mask_partial:
        assert corr_res['Mean']['n'] == src_res['Mean']['n']
        assert (corr_mask == src_mask).all()
    else:
        assert corr_res['Mean']['n'] < src_res['Mean']['n']
        assert 0 < corr_mask.sum() < src_mask.sum() and src_mask[corr_mask].all()


def utils_overlap(kernel_shape):
    from homonim_amd import utils
    return utils.overlap_for_kernel(kernel_shape)


@pytest.mark.parametrize('model, kernel_shape, mask_partial', [
    (Model.gain_blk_offset, (5, 5), False), (Model.gain_offset, (5, 5), False), (Model.gain, (3, 3), True),
    (Model.gain_offset, (15, 15), True),
])
def test_fused_refspace_pipeline_equals_step_by_step(pair, model, kernel_shape, mask_partial):
    """ hk_refspace_fit_apply keeps the re-sampled source, the parameters and the masks in HBM; it must reproduce the
    step-by-step RefSpaceModel.fit -> apply (four host round trips) bit for bit -- float32 and uint8 in / out. """
    import warnings
    from homonim_amd.fuse import convert_dtype
    src, src_tf, src_nodata, ref, ref_tf = pair
    crs = CRS('EPSG:32735')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        km = RefSpaceModel(model, kernel_shape, find_r2=True, mask_partial=mask_partial)
    src_ra = RasterArray(src[1].astype(np.float32), crs, src_tf, nodata=src_nodata)
    ref_ra = RasterArray(ref[1].astype(np.float32), crs, ref_tf, nodata=None)
    param_ra = km.fit(src_ra, ref_ra)
    corr_ra = km.apply(src_ra, param_ra)
    f_corr, f_param = km.fit_apply(src_ra, ref_ra, want_params=True)
    same = lambda a, b: bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
    assert same(f_param.array, param_ra.array) and f_param.transform == ref_ra.transform
    assert same(f_corr.array, corr_ra.array) and f_corr.transform == src_ra.transform
    # typed: uint8 rasters in, uint8 corrected block out, converted on the device
    b_corr, _ = km.fit_apply(RasterArray(src[1], crs, src_tf, nodata=src_nodata), RasterArray(ref[1], crs, ref_tf, nodata=None),
                             out_dtype='uint8', out_nodata=0)
    assert b_corr.array.dtype == np.uint8
    np.testing.assert_array_equal(b_corr.array, convert_dtype(corr_ra.array, 'uint8', 0))


@pytest.mark.oracle
@pytest.mark.parametrize('model, kernel_shape, upsampling, mask_partial', [
    ('gain-offset', (5, 5), 'cubic_spline', False), ('gain', (3, 3), 'bilinear', False), ('gain-offset', (3, 3), 'nearest', False),
    ('gain', (3, 3), 'cubic_spline', True), ('gain-offset', (5, 5), 'average', True), ('gain-blk-offset', (5, 5), 'cubic_spline', False),
])
def test_fused_refspace_pipeline_vs_the_oracle_step_by_step(pair, model, kernel_shape, upsampling, mask_partial):
    """ hk_refspace_fit_apply against the ORACLE's composition of the same steps (kernel_model.py:476-503): `average` down-sampling
    of the source to the reference grid, the fit there, gain and offset brought back with the up-sampling method -- cubic spline and
    bilinear fused into the apply kernel, the others through the general re-sampler + the re-masking apply --, mask_partial's
    eroded full-coverage mask brought back with `nearest`.  Bit for bit (gain-blk-offset: the block statistics are the GPU's exact
    float64 ones against numpy's float32 pairwise ones, 1e-5 relative). """
    import warnings
    from homonim_amd.geo import grid_mapping
    from oracle import oracle_np as onp
    src, src_tf, src_nodata, ref, ref_tf = pair
    crs = CRS('EPSG:32735')
    s, r = src[1].astype(np.float32), ref[1].astype(np.float32)
    down, up = grid_mapping(src_tf, ref_tf), grid_mapping(ref_tf, src_tf)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ds = onp.reproject(s, src_nodata, down, r.shape, dst_nodata=np.nan, resampling='average')
        norm = onp.fit_block_norm(ds, np.nan, r, None) if model == 'gain-blk-offset' else None
        params, _ = onp.fit(model, ds, np.nan, r, None, kernel_shape, False, None, norm_model=norm)
        p_us = np.stack([onp.reproject(params[b], np.nan, up, s.shape, dst_nodata=np.nan, resampling=upsampling) for b in range(2)])
        valid = onp.mask_of(s, src_nodata)
        if mask_partial:
            cover = onp.reproject(valid.astype(np.float32), None, down, r.shape, dst_nodata=None, resampling='average')
            keep = onp.full_coverage_mask(cover >= 1, params[:2], kernel_shape)
            valid = onp.reproject(keep.astype(np.float32), None, up, s.shape, dst_nodata=0, resampling='nearest').astype(bool)
        exp = np.where(valid, p_us[0] * s + p_us[1], np.float32(np.nan)).astype(np.float32)
        km = RefSpaceModel(model, kernel_shape, find_r2=False, mask_partial=mask_partial, r2_inpaint_thresh=None, upsampling=getattr(Resampling, upsampling))
        corr_ra, param_ra = km.fit_apply(RasterArray(s, crs, src_tf, nodata=src_nodata), RasterArray(r, crs, ref_tf, nodata=None),
                                         want_params=True)
    got = corr_ra.array
    assert (np.isnan(got) == np.isnan(exp)).all()
    ok = ~np.isnan(exp)
    assert ok.sum() > 0.3 * ok.size
    if model == 'gain-blk-offset':
        assert np.max(np.abs(got[ok] - exp[ok]) / np.maximum(np.abs(exp[ok]), 1e-6)) < 1e-5
    else:
        bad = int((got[ok] != exp[ok]).sum())
        assert bad <= max(2, 1e-5 * int(ok.sum())), f'{bad} of {int(ok.sum())} corrected pixels differ from the oracle composition'
        pe, pg = params[:2], param_ra.array[:2]
        assert (((pe == pg) | (np.isnan(pe) & np.isnan(pg))).mean()) > 1 - 1e-5


@pytest.mark.oracle
@pytest.mark.parametrize('image', [1, 2, 3, 4])
def test_published_accuracy_table_of_the_real_stack(image):
    """ The reference PUBLISHES, for its own test rasters, what the real homonim + OpenCV + GDAL stack prints for
    `homonim fuse -m gain-blk-offset -k 5 5` (NGI aerial tile x Sentinel-2) followed by `homonim compare` against a
    Landsat-8 image (docs/cli.rst:47-72, docs/api.rst:33-36; tests/golden/docs_table.json, oracle/gen_docs_table_fixture.py).
    The SOURCE rows pin GDAL's `average` down-sampling (5 m -> 30 m, nodata handling, grid alignment) and the comparison
    sums; the CORRECTED rows pin the whole RefSpace chain on real imagery -- average down-sampling onto the 10 m grid,
    block normalisation, the gain-blk-offset fit, cubic-spline up-sampling of the parameters, apply.  N must agree
    exactly, r2 / RMSE / rRMSE to every printed digit. """
    import json
    import warnings
    from homonim_amd.compare import RasterCompare
    from homonim_amd.fuse import RasterFuse
    from homonim_amd.tiff import read_tiff
    with open(os.path.join(GOLDEN_DIR, 'docs_table.json')) as f:
        fx = json.load(f)
    name = f'ngi_rgb_byte_{image}.tif'
    src_path = os.path.join(RASTER_DIR, name)
    l8 = read_tiff(os.path.join(RASTER_DIR, fx['compare']['reference']))
    bands = [b - 1 for b in fx['compare']['ref_bands_1based']]        # wavelength pairing of homonim/matched_pair.py
    l8_ra = RasterArray(np.ascontiguousarray(l8.array[bands]), l8.crs, l8.transform, nodata=l8.nodata)
    src = read_tiff(src_path)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with RasterCompare(RasterArray(src.array, src.crs, src.transform, nodata=src.nodata), l8_ra) as cmp:
            src_stats = cmp.process()['Mean']
        with RasterFuse(src_path, os.path.join(RASTER_DIR, fx['fuse']['reference'])) as rf:
            corr, _ = rf.process(None, fx['fuse']['model'], tuple(fx['fuse']['kernel_shape']))
        with RasterCompare(RasterArray(corr, src.crs, src.transform, nodata=float('nan')), l8_ra) as cmp:
            corr_stats = cmp.process()['Mean']
    for row, got in ((name, src_stats), (name.replace('.tif', '_FUSE_cREF_mGAIN-BLK-OFFSET_k5_5.tif'), corr_stats)):
        exp = fx['rows'][row]
        assert got['n'] == exp['n'], (row, got, exp)
        for key in ('r2', 'rmse', 'rrmse'):
            assert f'{got[key]:.3f}' == exp[key], (row, key, got, exp)


def _vrt_mosaic(tiles, layout, fill):
    """ gdal.BuildVRT mosaic of `tiles` (list of (bands, h, w) arrays in the VRT's source order) on the VRT's grid, as GDAL
    reads it (VRTSimpleSource::GetSrcDstWindow + the nearest-neighbour RasterIO behind it): a source whose DstRect offset has
    the fractional part f lands at column floor(off) + (1 if f > 0.5 else 0); the destination rectangle covers one pixel more
    than the source (floor(off) .. ceil(off + size)), which is filled by repeating the source's first (f > 0.5) or last
    (f <= 0.5) column / row; later sources overwrite earlier ones except where they hold nodata (NaN for float rasters). """
    nb = tiles[0].shape[0]
    out = np.full((nb, layout['height'], layout['width']), fill, tiles[0].dtype)

    def axis_index(off, n_src, n_dst_total):
        o0 = int(np.floor(off + 0.001))
        o1 = int(np.ceil(off + n_src - 0.001))
        j = np.arange(o1 - o0)
        src = np.floor(j + 0.5 - (off - o0) + 1e-10).astype(np.int64).clip(0, n_src - 1)   # GDAL's nearest pixel, clamped
        keep = (o0 + j >= 0) & (o0 + j < n_dst_total)
        return o0 + j[keep], src[keep]

    for tile, t in zip(tiles, layout['tiles']):
        cols, sc = axis_index(t['x_off'], t['width'], layout['width'])
        rows, sr = axis_index(t['y_off'], t['height'], layout['height'])
        assert tile.shape[1:] == (t['height'], t['width'])
        placed = tile[:, sr][:, :, sc]
        valid = ~np.isnan(placed) if np.isnan(fill) else (placed != t['nodata'])
        view = out[:, rows[0]:rows[-1] + 1, cols[0]:cols[-1] + 1]
        view[valid] = placed[valid]
    return out


@pytest.mark.oracle
def test_published_mosaic_table_of_the_real_stack():
    """ The reference's tutorial notebook prints a SECOND accuracy table of the real homonim + OpenCV + GDAL stack
    (docs/tutorials/basic_correction.ipynb:299-334; tests/golden/notebook_table.json, oracle/gen_notebook_table_fixture.py):
    the four NGI tiles fused with Sentinel-2 (gain-blk-offset 5x5), mosaicked (gdal.BuildVRT) and compared PER BAND with
    the Landsat-8 image -- source mosaic and corrected mosaic, N = 76 143.  All eight band rows + the two mean rows: N
    exactly, r2 / RMSE / rRMSE to every printed digit.  On top of the per-tile table this pins per-band (not only mean)
    statistics, the `average` re-sampling over a mosaic with interior nodata seams, and the whole RefSpace chain on all four
    tiles at once. """
    import json
    import warnings
    from homonim_amd.compare import RasterCompare
    from homonim_amd.fuse import RasterFuse
    from homonim_amd.tiff import read_tiff
    with open(os.path.join(GOLDEN_DIR, 'notebook_table.json')) as f:
        fx = json.load(f)
    with open(os.path.join(GOLDEN_DIR, 'docs_table.json')) as f:
        bands = [b - 1 for b in json.load(f)['compare']['ref_bands_1based']]   # wavelength pairing of homonim/matched_pair.py
    lay = fx['mosaic']
    gt = lay['geotransform']
    mosaic_tf = Affine(gt[1], gt[2], gt[0], gt[4], gt[5], gt[3])
    l8 = read_tiff(os.path.join(RASTER_DIR, fx['compare']['reference']))
    l8_ra = RasterArray(np.ascontiguousarray(l8.array[bands]), l8.crs, l8.transform, nodata=l8.nodata)
    src_tiles, corr_tiles = [], []
    crs = None
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for t in lay['tiles']:
            path = os.path.join(RASTER_DIR, t['file'])
            tif = read_tiff(path)
            crs = tif.crs
            src_tiles.append(tif.array)
            with RasterFuse(path, os.path.join(RASTER_DIR, fx['fuse']['reference'])) as rf:
                corr, _ = rf.process(None, fx['fuse']['model'], tuple(fx['fuse']['kernel_shape']))
            corr_tiles.append(corr)
        results = {}
        for label, tiles, fill in (('Source', src_tiles, lay['nodata']), ('Corrected', corr_tiles, np.nan)):
            mosaic = _vrt_mosaic(tiles, lay, np.float32(fill) if label == 'Corrected' else src_tiles[0].dtype.type(fill))
            ra = RasterArray(mosaic, crs, mosaic_tf, nodata=float(fill))
            with RasterCompare(ra, l8_ra) as cmp:
                results[label] = list(cmp.process().values())
    for label, rows in fx['tables'].items():
        for got, (name, exp) in zip(results[label], rows.items()):
            assert got['n'] == exp['n'], (label, name, got, exp)
            for key in ('r2', 'rmse', 'rrmse'):
                assert f'{got[key]:.3f}' == exp[key], (label, name, key, got, exp)
