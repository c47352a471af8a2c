""" Debug: event-timed hk_block_norm_dev over a 16384^2 x B raster (one launch), for A/B of libraries. """
import sys, time
import numpy as np
sys.path.insert(0, '.')
from homonim_amd import _hk
H = W = 16384
B = 4
ctx = _hk.get_context(0)
stride = W
band_stride = stride * H
desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, None, None)
src, ref = ctx.dev_alloc(4 * band_stride * B), ctx.dev_alloc(4 * band_stride * B)
norm = ctx.dev_alloc(16 * B)
ctx.synth_fill_dev(src, ref, B, H, W, stride, band_stride, seed=1234, nodata_variant=0, stream=0)
job = _hk.DevJob()
job.src, job.ref, job.corr, job.norm = src, ref, None, norm
job.n_bands, job.height, job.width, job.stride, job.band_stride = B, H, W, stride, band_stride
job.seg_rows, job.stream = 0, 0
for rep in range(3):
    ctx.stream_sync(0)
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.block_norm_dev(desc, job, norm)
    ctx.stream_sync(0)
    t = (time.perf_counter() - t0) * 100
nm = np.zeros((B, 2)); ctx.d2h(nm, norm)
print('%.3f ms per 4-band block norm' % t, nm[0])
