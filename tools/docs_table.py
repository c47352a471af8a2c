"""Reproduce the accuracy table the reference publishes for its own test rasters (docs/cli.rst:61-72, docs/api.rst:35-36):
`homonim fuse -m gain-blk-offset -k 5 5` of ngi_rgb_byte_1.tif with sentinel2_b432_byte.tif, then `homonim compare`
of the source and of the corrected image against landsat8_byte.tif.  Run on the GPU box from the repo root."""
import json
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homonim_amd import RasterArray
from homonim_amd.compare import RasterCompare
from homonim_amd.fuse import RasterFuse
from homonim_amd.tiff import read_tiff

R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'rasters')
# band pairing of homonim/matched_pair.py:95-179: an RGB image without wavelength tags gets 0.650 / 0.560 / 0.480 um, which
# match Landsat-8 SR_B4 (0.655), SR_B3 (0.562), SR_B2 (0.482) = file bands 4, 3, 2 (landsat8_byte.vrt)
L8_BANDS = [3, 2, 1]


def main():
    warnings.simplefilter('ignore')
    src, s2, l8 = (read_tiff(os.path.join(R, n)) for n in ('ngi_rgb_byte_1.tif', 'sentinel2_b432_byte.tif', 'landsat8_byte.tif'))
    l8_ra = RasterArray(np.ascontiguousarray(l8.array[L8_BANDS]), l8.crs, l8.transform, nodata=l8.nodata)
    src_ra = RasterArray(src.array, src.crs, src.transform, nodata=src.nodata)
    out = {}
    with RasterCompare(src_ra, l8_ra) as cmp:
        out['source'] = cmp.process()
    with RasterFuse(os.path.join(R, 'ngi_rgb_byte_1.tif'), os.path.join(R, 'sentinel2_b432_byte.tif')) as rf:
        corr = rf.process(None, 'gain-blk-offset', (5, 5))
    corr_arr = corr[0] if isinstance(corr, tuple) else corr
    corr_ra = RasterArray(np.asarray(corr_arr), src.crs, src.transform, nodata=float('nan'))
    with RasterCompare(corr_ra, l8_ra) as cmp:
        out['corrected'] = cmp.process()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
