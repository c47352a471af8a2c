#!/usr/bin/env python3
""" tools/stage_stamps.py [bench-like args]: where a wave of the fused kernel spends its time.  Needs a library whose hk_kernels.hip
was built with -DHK_STAMPS (tools/mkvariant.sh st -DHK_STAMPS [-DHK_DEV_SUBSET15]; HOMONIM_AMD_LIB=_ab/lib_st.so): every wave
accumulates shader-clock cycles per stage of its row iterations (s_memtime; a wait for memory lands in the stage that needs the
data).  Prints the mean cycles per iteration and stage.  The stamps cost ~10 % of the kernel's time. """
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from homonim_amd import _hk  # noqa: E402

STAGES = ['requests issued, entering row arrived + classified', 'next row requested, ring traffic, leaving row arrived + classified',
          'column sums updated', 'centre row, horizontal sums, window counts', 'pointwise stages + stores',
          'loop bookkeeping (+ set-up in the first iteration)']


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--model', default='gain-blk-offset')
    p.add_argument('--kernel', type=int, default=15)
    p.add_argument('--size', type=int, default=16384)
    p.add_argument('--bands', type=int, default=8)
    p.add_argument('--nodata', type=int, default=0)
    p.add_argument('--no-thresh', action='store_true')
    p.add_argument('--steps', type=int, default=3)
    a = p.parse_args()
    ctx = _hk.Context(int(os.environ.get('LOCAL_RANK', '0')), n_streams=1)
    n, B = a.size, a.bands
    plane = n * n
    bufs = {k: ctx.dev_alloc(4 * plane * B) for k in ('src', 'ref', 'corr')}
    norm, fail = ctx.dev_alloc(16 * B), ctx.dev_alloc(8 * B)
    ctx.memset(fail, 0, 8 * B)
    ctx.synth_fill_dev(bufs['src'], bufs['ref'], B, n, n, n, plane, seed=1234, nodata_variant=a.nodata, stream=0)
    nd = np.nan if a.nodata in (1, 2) else None
    thresh = 0.25 if (a.model == 'gain-offset' and not a.no_thresh) else None
    desc = _hk.make_desc(a.model, (a.kernel, a.kernel), False, thresh, nd, nd)
    job = _hk.DevJob()
    job.src, job.ref, job.corr = bufs['src'], bufs['ref'], bufs['corr']
    job.gain = job.offset = job.r2 = None
    job.fail_count = fail if thresh is not None else None
    job.norm = norm if a.model == 'gain-blk-offset' else None
    job.n_bands, job.height, job.width, job.stride, job.band_stride = B, n, n, n, plane
    job.seg_rows, job.stream = 0, 0
    if a.model == 'gain-blk-offset':
        ctx.block_norm_dev(desc, job, norm)
    ctx.fit_apply_dev(desc, job)   # warm-up
    ctx.stream_sync(0)
    ctx.debug_stage_stamps(reset=True)
    ev0, ev1 = ctx.event(), ctx.event()
    ctx.event_record(ev0, 0)
    for _ in range(a.steps):
        ctx.fit_apply_dev(desc, job)
    ctx.event_record(ev1, 0)
    ms = ctx.event_elapsed_ms(ev0, ev1) / a.steps
    st = ctx.debug_stage_stamps(reset=True).astype(np.float64)
    iters, waves = st[15], st[14]
    if iters == 0:
        raise SystemExit('no stamps: the loaded library was not built with -DHK_STAMPS')
    print(f'{a.model} {a.kernel}x{a.kernel}, {B} x {n}^2, nodata variant {a.nodata}: {ms:.3f} ms per launch (with stamps), '
          f'{waves / a.steps:.0f} waves of {iters / waves:.1f} row iterations')
    tot = st[:6].sum()
    for k in (5, 0, 1, 2, 3, 4):
        print(f'  {st[k] / iters:9.0f} cycles per iteration  {100 * st[k] / tot:5.1f} %   {STAGES[k]}')
    print(f'  {tot / iters:9.0f} cycles per iteration in all')
    ctx.close()


if __name__ == '__main__':
    main()
