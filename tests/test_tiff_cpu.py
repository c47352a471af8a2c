"""
CPU tests of the GeoTIFF subset reader / writer (homonim_amd/tiff.py) against fixtures decoded by an independent
implementation (tifffile, oracle/gen_tiff_fixtures.py): three data files of the reference's own tests and synthetic
files in the layouts its larger rasters use.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from homonim_amd import Affine, CRS
from homonim_amd.errors import IoError
from homonim_amd.tiff import read_tiff, write_tiff

TIFF_DIR = os.path.join(GOLDEN_DIR, 'tiff')
FILES = sorted(glob.glob(os.path.join(TIFF_DIR, '*.tif')))


@pytest.mark.parametrize('path', FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_reader_decodes_like_tifffile(path):
    decoded = np.load(os.path.join(TIFF_DIR, 'decoded.npz'))
    exp = decoded[os.path.basename(path)[:-4]]
    got = read_tiff(path)
    assert got.array.dtype == exp.dtype and got.array.shape == exp.shape
    assert np.array_equal(got.array, exp, equal_nan=exp.dtype.kind == 'f')


def test_georeferencing_nodata_and_metadata_of_reference_files():
    assert len(FILES) == 9
    modis = read_tiff(os.path.join(TIFF_DIR, 'modis_nbar.tif'))
    assert modis.nodata == -32768 and modis.array.shape == (15, 31, 18)
    assert tuple(modis.transform) == pytest.approx((463.312716528, 0, -60693.965865168, 0, -463.312716528, -3722254.364585952))
    assert 'LICENSE' in modis.metadata
    assert '3072=32767' in modis.crs.to_string()          # user-defined projection: labelled by its full definition
    param = read_tiff(os.path.join(TIFF_DIR, 'float_100cm_rgb_FUSE_cREF_mGAIN-OFFSET_k5_5_PARAM.tif'))
    assert param.crs == CRS('EPSG:3857') and np.isnan(param.nodata)
    assert param.metadata['FUSE_DOWNSAMPLING'] == 'average' and param.metadata['FUSE_KERNEL_SHAPE'] == '(5, 5)'
    strips = read_tiff(os.path.join(TIFF_DIR, 'float_100cm_rgb_FUSE_cREF_mGAIN-OFFSET_k5_5_PARAM_tile_10x20.tif'))
    assert np.array_equal(strips.array, param.array, equal_nan=True)     # same parameters, strips vs tiles
    synth = read_tiff(os.path.join(TIFF_DIR, 'synth_contig_strips_u8.tif'))
    assert synth.crs == CRS('EPSG:32735') and synth.nodata == 0
    assert synth.transform == Affine(10., 0., -60370., 0., -10., -3722700.)


@pytest.mark.parametrize('dtype, nodata', [('uint8', 0), ('uint16', 65535), ('int16', -32768), ('float32', float('nan')),
                                           ('float32', None), ('float64', -9999.0)])
@pytest.mark.parametrize('shape, tile', [((3, 100, 130), 64), ((1, 17, 5), 512), ((2, 512, 512), 512)])
def test_write_read_round_trip(tmp_path, dtype, nodata, shape, tile):
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 200, shape).astype(dtype)
    if dtype.startswith('float'):
        a[0, :3] = np.nan
    tf = Affine(5., 0., -57129.5, 0., -5., -3723906.75)
    meta = dict(FUSE_MODEL='gain_blk_offset', FUSE_KERNEL_SHAPE='(5, 5)', note='a <b> & "c"')
    path = tmp_path / 'out.tif'
    write_tiff(path, a, tf, CRS('EPSG:32735'), nodata, meta, tile=tile)
    r = read_tiff(path)
    assert r.array.dtype == a.dtype and np.array_equal(r.array, a, equal_nan=True)
    assert r.transform == tf and r.crs == CRS('EPSG:32735') and r.metadata == meta
    assert (r.nodata is None) if nodata is None else (np.isnan(r.nodata) if np.isnan(nodata) else r.nodata == nodata)
    # a CRS without an EPSG code travels as its label; 2-D input is one band; uncompressed works too
    write_tiff(path, a[0], tf, CRS('my local grid'), nodata, compress=False)
    r = read_tiff(path)
    assert r.array.shape == (1, *shape[1:]) and 'my local grid' in r.crs.to_string() and r.metadata == {}


def test_unsupported_files_raise(tmp_path):
    p = tmp_path / 'x.tif'
    p.write_bytes(b'not a tiff at all')
    with pytest.raises(IoError):
        read_tiff(p)
    with pytest.raises(IoError):
        write_tiff(p, np.zeros((4, 4), np.complex64), Affine.identity())
    with pytest.raises(IoError):
        write_tiff(p, np.zeros((4, 4), np.float32), Affine(1., 0.5, 0., 0., -1., 0.))
