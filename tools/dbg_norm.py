""" Debug: wall time and result of hk_block_norm_dev for every block position of bench config 3, one stream. """
import sys, time
import numpy as np
sys.path.insert(0, '.')
from homonim_amd import _hk, utils
from homonim_amd.fuse import block_pairs

H = W = 16384
B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 15
ctx = _hk.get_context(0)
overlap = utils.overlap_for_kernel((k, k))
positions = [bp for bp in block_pairs((H, W), 8, overlap, 100) if bp.band_i == 0]
stride = (W + 63) // 64 * 64
band_stride = stride * H
desc = _hk.make_desc('gain-blk-offset', (k, k), False, None, None, None)
src, ref = ctx.dev_alloc(4 * band_stride * B), ctx.dev_alloc(4 * band_stride * B)
norm = ctx.dev_alloc(16 * B)
ctx.synth_fill_dev(src, ref, B, H, W, stride, band_stride, seed=1234, nodata_variant=0, stream=0)
ctx.stream_sync(0)
for i, bp in enumerate(positions):
    wi = bp.src_in_block
    off = 4 * (wi.row_off * stride + wi.col_off)
    job = _hk.DevJob()
    job.src, job.ref, job.corr = src + off, ref + off, None
    job.norm = norm
    job.n_bands, job.height, job.width, job.stride, job.band_stride = B, wi.height, wi.width, stride, band_stride
    job.seg_rows, job.stream = 0, 0
    ts, res = [], []
    for rep in range(3):
        ctx.stream_sync(0)
        t0 = time.perf_counter()
        ctx.block_norm_dev(desc, job, norm)
        ctx.stream_sync(0)
        ts.append((time.perf_counter() - t0) * 1e3)
        nm = np.zeros((B, 2))
        ctx.d2h(nm, norm)
        res.append(nm.copy())
    same = all((r == res[0]).all() for r in res)
    print(i, (wi.row_off, wi.col_off, wi.height, wi.width), ' '.join('%.3f' % t for t in ts), 'same' if same else 'DIFF', res[0][0])
