#!/usr/bin/env python3
""" Device-resident timing of the comparison-sums reduction (hk_compare_sums_dev): 8 B read per pixel*band. """
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homonim_amd import _hk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=16384)
    ap.add_argument('--bands', type=int, default=4)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--nodata', type=int, default=1)
    a = ap.parse_args()
    ctx = _hk.Context(0, n_streams=1)
    H = W = a.size
    stride = (W + 63) // 64 * 64
    band_stride = stride * H
    d = {k: ctx.dev_alloc(4 * band_stride * a.bands) for k in ('src', 'ref')}
    d['sums'] = ctx.dev_alloc(56 * a.bands)
    ctx.synth_fill_dev(d['src'], d['ref'], a.bands, H, W, stride, band_stride, seed=1, nodata_variant=a.nodata, stream=0)
    job = _hk.DevJob()
    job.src, job.ref = d['src'], d['ref']
    job.corr = job.gain = job.offset = job.r2 = job.norm = job.fail_count = None
    job.n_bands, job.height, job.width, job.stride, job.band_stride = a.bands, H, W, stride, band_stride
    job.seg_rows, job.stream = 0, 0
    nd = np.nan if a.nodata in (1, 2) else None
    for _ in range(3):
        ctx.compare_sums_dev(job, nd, nd, d['sums'])
    ev = [(ctx.event(), ctx.event()) for _ in range(a.steps)]
    for e0, e1 in ev:
        ctx.event_record(e0, 0)
        ctx.compare_sums_dev(job, nd, nd, d['sums'])
        ctx.event_record(e1, 0)
    ctx.stream_sync(0)
    ms = float(np.median([ctx.event_elapsed_ms(e0, e1) for e0, e1 in ev]))
    sums = np.zeros((a.bands, 7))
    ctx.d2h(sums, d['sums'])
    gb = 8.0 * H * W * a.bands / 1e9
    print(json.dumps(dict(kernel='compare_sums', size=a.size, bands=a.bands, ms=round(ms, 4), GBps=round(gb / ms * 1e3, 1),
                          frac_of_8TBps=round(gb / ms * 1e3 / 8000, 4), n_valid=int(sums[:, 6].sum()))))


if __name__ == '__main__':
    main()
