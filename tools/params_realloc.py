#!/usr/bin/env python3
""" tools/params_realloc.py: does the bimodal timing of the six-stream launch (`bench.py --params`: src, ref -> corr, gain, offset, r2)
follow the ALLOCATION?  One process allocates the planes, times the launch, frees everything, allocates again (with a spacer of
varying size in between) and times again. """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from homonim_amd import _hk  # noqa: E402


def main():
    params = '--no-params' not in sys.argv   # --no-params: the headline's three planes only
    ctx = _hk.Context(0, n_streams=1)
    n, B = 16384, 4
    plane = n * n
    desc = _hk.make_desc('gain-offset', (5, 5), params, 0.25, None, None)
    for trial in range(8):
        spacer = ctx.dev_alloc((trial * 37 + 1) << 20) if trial else None
        bufs = {k: ctx.dev_alloc(4 * plane * B) for k in (('src', 'ref', 'corr', 'gain', 'offset', 'r2') if params else ('src', 'ref', 'corr'))}
        fail = ctx.dev_alloc(8 * B)
        ctx.memset(fail, 0, 8 * B)
        ctx.synth_fill_dev(bufs['src'], bufs['ref'], B, n, n, n, plane, seed=1234, nodata_variant=0, stream=0)
        job = _hk.DevJob()
        job.src, job.ref, job.corr = bufs['src'], bufs['ref'], bufs['corr']
        job.gain, job.offset, job.r2 = (bufs['gain'], bufs['offset'], bufs['r2']) if params else (None, None, None)
        job.fail_count, job.norm = fail, None
        job.n_bands, job.height, job.width, job.stride, job.band_stride = B, n, n, n, plane
        job.seg_rows, job.stream = 0, 0
        for _ in range(3):
            ctx.fit_apply_dev(desc, job)
        ctx.stream_sync(0)
        e0, e1 = ctx.event(), ctx.event()
        ctx.event_record(e0, 0)
        for _ in range(20):
            ctx.fit_apply_dev(desc, job)
        ctx.event_record(e1, 0)
        ms = ctx.event_elapsed_ms(e0, e1) / 20
        print(f'trial {trial}: {ms:.3f} ms per launch; src at {bufs["src"]:#x}, corr at {bufs["corr"]:#x}')
        ctx.event_destroy(e0), ctx.event_destroy(e1)
        for p in bufs.values():
            ctx.dev_free(p)
        ctx.dev_free(fail)
        if spacer:
            ctx.dev_free(spacer)
    ctx.close()


if __name__ == '__main__':
    main()
