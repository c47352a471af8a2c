""" configs[3], one (band, block) at a time: is the fit of a 4104 x 4104 in-block faster when the block's statistics pass has
just streamed its src / ref planes (135 MB) through the 256 MB Infinity Cache?  

    python tools/c3_mall_probe.py [seg_rows ...]

For each segment height: the fit of one block of one band, timed with HIP events on its stream,
  cold : after a 2 x 1.6 GB flat stream over other buffers (the block's lines are gone from every cache)
  warm : right after hk_block_norm_dev of the same block (temporal loads: what a statistics -> fit chain would see)
  hot  : right after the same fit (src, ref and corr lines all recently used)
and the statistics pass itself, cold and hot. """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homonim_amd import _hk  # noqa: E402

ctx = _hk.Context(0, n_streams=2)
H = W = 16384
stride, plane = W, W * H
nbands = int(os.environ.get('PROBE_BANDS', '1'))
bufs = {k: ctx.dev_alloc(4 * plane * nbands) for k in ('src', 'ref', 'corr')}
junk = [ctx.dev_alloc(1600 << 20) for _ in range(3)]
norm = ctx.dev_alloc(16 * nbands)
ctx.synth_fill_dev(bufs['src'], bufs['ref'], nbands, H, W, stride, plane, seed=1, nodata_variant=0, stream=0)
ctx.stream_sync(0)
desc = _hk.make_desc('gain-blk-offset', (15, 15), False, None, None, None)
ev = [ctx.event() for _ in range(4)]


def job_for(seg_rows, row_off=4088, col_off=4088, h=4112, w=4112):
    job = _hk.DevJob()
    off = 4 * (row_off * stride + col_off)
    job.src, job.ref, job.corr = bufs['src'] + off, bufs['ref'] + off, bufs['corr'] + off
    job.gain = job.offset = job.r2 = job.fail_count = None
    job.norm = norm
    job.n_bands, job.height, job.width, job.stride, job.band_stride = nbands, h, w, stride, plane
    job.seg_rows, job.stream = seg_rows, 0
    job.out_row0, job.out_col0, job.out_rows, job.out_cols = 8, 8, 4096, 4096
    return job


def flush():
    ctx.stream_probe_dev(junk[0], junk[1], junk[2], 1600 << 20, 0)


def timed(fn, before):
    ts = []
    for _ in range(12):
        before()
        ctx.event_record(ev[0], 0)
        fn()
        ctx.event_record(ev[1], 0)
        ctx.stream_sync(0)
        ts.append(ctx.event_elapsed_ms(ev[0], ev[1]))
    return float(np.median(ts[2:])) * 1e3


segs = [int(a) for a in sys.argv[1:]] or [0, 128, 64, 32]
px = 4096 * 4096 * nbands
print(f'# one block position, {nbands} band(s): in-block 4112 x 4112 = {8 * 4112 * 4112 * nbands / 1e6:.0f} MB of src + ref, corr {4 * px / 1e6:.0f} MB; microseconds, medians of 10')
job = job_for(0)
t_stats_cold = timed(lambda: ctx.block_norm_dev(desc, job, norm), flush)
t_stats_hot = timed(lambda: ctx.block_norm_dev(desc, job, norm), lambda: ctx.block_norm_dev(desc, job, norm))
print(f'statistics pass (all its launches): cold {t_stats_cold:8.1f}  hot {t_stats_hot:8.1f}   ({8 * 4112 * 4112 * nbands / t_stats_cold / 1e6:.2f} / {8 * 4112 * 4112 * nbands / t_stats_hot / 1e6:.2f} TB/s)')
for sr in segs:
    job = job_for(sr)
    fit = lambda: ctx.fit_apply_dev(desc, job)  # noqa: E731
    cold = timed(fit, flush)
    warm = timed(fit, lambda: (flush(), ctx.block_norm_dev(desc, job, norm)))
    hot = timed(fit, fit)
    print(f'fit seg_rows {sr:4d}: cold {cold:8.1f}  warm (after its statistics) {warm:8.1f}  hot (after itself) {hot:8.1f}'
          f'   -> {12 * px / cold / 1e6:.2f} / {12 * px / warm / 1e6:.2f} / {12 * px / hot / 1e6:.2f} TB/s algorithmic')
