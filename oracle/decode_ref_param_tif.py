"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference and tifffile, which is
only importable under /opt/conda/bin/python3.9 here):

    /opt/conda/bin/python3.9 oracle/decode_ref_param_tif.py

Decodes the reference's own golden data file
``tests/data/parameter/float_100cm_rgb_FUSE_cREF_mGAIN-OFFSET_k5_5_PARAM.tif`` (9 bands x 20 x 10 float32: gain,
offset, R2 for each of 3 source bands = i * ``array_100cm_float``, reference tests/conftest.py:351-374) into
``tests/golden/ref_param_tif.npz``.  This is DATA the reference's tests hold (tests/test_stats.py:36-51), produced
by the real homonim + OpenCV + GDAL stack -- the only artefact here that pins the OpenCV float64-accumulation
boundary (142/144 valid px exactly (1, 0, 1); two px off by <=1.3e-3 in offset, 66 ulp in gain).
"""
import os

import numpy as np
import tifffile

REF = '/root/reference/tests/data/parameter/float_100cm_rgb_FUSE_cREF_mGAIN-OFFSET_k5_5_PARAM.tif'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'ref_param_tif.npz')

with tifffile.TiffFile(REF) as tif:
    arr = tif.asarray()
    page = tif.pages[0]
    tags = {t.name: str(t.value)[:2000] for t in page.tags.values() if t.name in ('GDAL_METADATA', 'GDAL_NODATA')}
arr = np.asarray(arr, dtype=np.float32)
if arr.ndim == 3 and arr.shape[-1] == 9:
    arr = np.moveaxis(arr, -1, 0)
print(arr.shape, arr.dtype, {k: v[:200] for k, v in tags.items()})
np.savez_compressed(OUT, params=arr, gdal_metadata=tags.get('GDAL_METADATA', ''), gdal_nodata=tags.get('GDAL_NODATA', ''))
