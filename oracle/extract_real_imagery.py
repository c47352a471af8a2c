"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference and tifffile, importable
under /opt/conda/bin/python3.9):

    /opt/conda/bin/python3.9 oracle/extract_real_imagery.py

Crops the reference's own test rasters (DATA its tests hold: tests/data/source/ngi_rgb_byte_1.tif, 5 m NGI aerial RGB,
nodata 0; tests/data/reference/sentinel2_b432_byte.tif, 10 m Sentinel-2 B4/B3/B2) to a small co-located window and
stores pixels + geo-transforms in tests/golden/real_imagery_crop.npz -- BASELINE.json configs[0] in miniature, used as
an acceptance test with the reference's integration criteria (tests/integration.py:79-83).
"""
import os

import numpy as np
import tifffile

REF = '/root/reference/tests/data'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'real_imagery_crop.npz')


def read(path):
    with tifffile.TiffFile(path) as tif:
        arr = tif.asarray()
        page = tif.pages[0]
        scale = page.tags['ModelPixelScaleTag'].value
        tie = page.tags['ModelTiepointTag'].value
        nodata = page.tags['GDAL_NODATA'].value if 'GDAL_NODATA' in page.tags else None
        geokeys = page.tags['GeoKeyDirectoryTag'].value if 'GeoKeyDirectoryTag' in page.tags else None
    if arr.ndim == 3 and arr.shape[-1] <= 4:
        arr = np.moveaxis(arr, -1, 0)
    # north-up: x = tie_x + (col - tie_i) * sx ; y = tie_y - (row - tie_j) * sy
    a, e = float(scale[0]), -float(scale[1])
    c = float(tie[3]) - float(tie[0]) * a
    f = float(tie[4]) - float(tie[1]) * e
    return arr, (a, 0.0, c, 0.0, e, f), nodata, geokeys


src, src_tf, src_nd, src_keys = read(os.path.join(REF, 'source', 'ngi_rgb_byte_1.tif'))
ref, ref_tf, ref_nd, ref_keys = read(os.path.join(REF, 'reference', 'sentinel2_b432_byte.tif'))
print('src', src.shape, src.dtype, src_tf, src_nd)
print('ref', ref.shape, ref.dtype, ref_tf, ref_nd)
print('geokeys equal:', src_keys == ref_keys)

# source window: 560 x 560 px around the middle of the tile
h, w = src.shape[-2:]
r0, c0, n = h // 2 - 280, w // 2 - 280, 560
src_win = src[:, r0:r0 + n, c0:c0 + n]
src_win_tf = (src_tf[0], 0.0, src_tf[2] + c0 * src_tf[0], 0.0, src_tf[4], src_tf[5] + r0 * src_tf[4])
# reference window covering it with a 4-px margin
x0, y0 = src_win_tf[2], src_win_tf[5]
x1, y1 = x0 + n * src_tf[0], y0 + n * src_tf[4]
rc0 = int(np.floor((x0 - ref_tf[2]) / ref_tf[0])) - 4
rc1 = int(np.ceil((x1 - ref_tf[2]) / ref_tf[0])) + 4
rr0 = int(np.floor((y0 - ref_tf[5]) / ref_tf[4])) - 4
rr1 = int(np.ceil((y1 - ref_tf[5]) / ref_tf[4])) + 4
assert rc0 >= 0 and rr0 >= 0 and rc1 <= ref.shape[-1] and rr1 <= ref.shape[-2], (rc0, rc1, rr0, rr1, ref.shape)
ref_win = ref[:, rr0:rr1, rc0:rc1]
ref_win_tf = (ref_tf[0], 0.0, ref_tf[2] + rc0 * ref_tf[0], 0.0, ref_tf[4], ref_tf[5] + rr0 * ref_tf[4])
print('src window', src_win.shape, src_win_tf, 'ref window', ref_win.shape, ref_win_tf)
print('fractional grid offset (ref px):', ((src_win_tf[2] - ref_win_tf[2]) / ref_tf[0]) % 1, ((src_win_tf[5] - ref_win_tf[5]) / ref_tf[4]) % 1)
np.savez_compressed(OUT, src=src_win, src_transform=np.array(src_win_tf), src_nodata=np.array(float(src_nd) if src_nd is not None else np.nan),
                    ref=ref_win, ref_transform=np.array(ref_win_tf), ref_has_nodata=np.array(ref_nd is not None))
print(os.path.getsize(OUT) / 1e6, 'MB')
