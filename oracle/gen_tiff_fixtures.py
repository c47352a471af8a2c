"""
Small TIFF fixtures for tests/test_tiff_cpu.py, written and decoded by an independent implementation (tifffile under
/opt/conda/bin/python3.9 -- this container only; nothing here runs on the GPU box):

* tests/golden/tiff/{*PARAM*.tif, modis_nbar.tif}: copied data files of the reference's own tests (tests/data);
* tests/golden/tiff/synth_*.tif: layouts the reference's larger rasters use (contig + strips as sentinel2_b432_byte.tif
  and landsat8_byte.tif, separate + 256-px tiles as the ngi_rgb_byte_*.tif) plus predictor 2, big-endian and BigTIFF;
* tests/golden/tiff/decoded.npz: every fixture decoded by tifffile.

Run:  /opt/conda/bin/python3.9 oracle/gen_tiff_fixtures.py
"""
import glob
import os

import numpy as np
import tifffile

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'tiff')
rng = np.random.default_rng(5)


def smooth(shape, dtype, lo, hi, noise=0.01):
    b, h, w = shape
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([np.sin(x / (7 + k)) * np.cos(y / (5 + k)) for k in range(b)]) * 0.5 + 0.5
    a = lo + (hi - lo) * a + rng.normal(0, (hi - lo) * noise, a.shape)
    return np.clip(a, lo, hi).astype(dtype)


geo = [(33550, 'd', 3, (10.0, 10.0, 0.0)), (33922, 'd', 6, (0.0, 0.0, 0.0, -60370.0, -3722700.0, 0.0)),
       (34735, 'H', 16, (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32735)), (42113, 's', 0, '0')]
cases = {
    'synth_contig_strips_u8': dict(data=np.moveaxis(smooth((3, 61, 47), 'u1', 0, 255), 0, -1), photometric='rgb',
                                   planarconfig='contig', compression='zlib', rowsperstrip=3),
    'synth_separate_tiles_u8': dict(data=smooth((3, 300, 270), 'u1', 0, 255, noise=0), photometric='minisblack',
                                    planarconfig='separate', compression='zlib', tile=(256, 256)),
    'synth_pred2_u16_strips': dict(data=smooth((2, 40, 33), 'u2', 0, 60000), photometric='minisblack',
                                   planarconfig='separate', compression='zlib', predictor=True, rowsperstrip=16),
    'synth_pred2_contig_i16': dict(data=np.moveaxis(smooth((4, 35, 50), 'i2', -3000, 3000), 0, -1), photometric='minisblack',
                                   planarconfig='contig', compression='zlib', predictor=True, tile=(16, 32), extrasamples=[0, 0, 0]),
    'synth_bigendian_f32': dict(data=smooth((1, 30, 20), 'f4', -1, 1)[0], photometric='minisblack', byteorder='>',
                                compression=None, rowsperstrip=8),
    'synth_bigtiff_f64_tiles': dict(data=np.round(smooth((2, 70, 90), 'f8', 0, 1, noise=0), 3), photometric='minisblack', planarconfig='separate',
                                    compression='zlib', tile=(32, 48), bigtiff=True),
}
for name, kw in cases.items():
    data = kw.pop('data')
    tifffile.imwrite(os.path.join(OUT, name + '.tif'), data, extratags=geo, metadata=None, **kw)

decoded = {}
for path in sorted(glob.glob(os.path.join(OUT, '*.tif'))):
    with tifffile.TiffFile(path) as t:
        page = t.pages[0]
        arr = page.asarray()
        if arr.ndim == 2:
            arr = arr[None]
        elif page.planarconfig == 1:      # contig: (h, w, samples) -> bands first
            arr = np.moveaxis(arr, -1, 0)
        decoded[os.path.basename(path)[:-4]] = np.ascontiguousarray(arr.astype(arr.dtype.newbyteorder('=')))
np.savez_compressed(os.path.join(OUT, 'decoded.npz'), **decoded)
print({k: (v.shape, str(v.dtype)) for k, v in decoded.items()})
