""" HK_GUARD_ALLOC (hk_api.hip: dev_malloc): every device allocation of the library in its own mapping between unmapped address
ranges, flush with the lower (lo) or the upper (hi) end, contents poisoned.  The switch is how the whole GPU suite is checked for
out-of-range accesses of the kernels (GPU AddressSanitizer is not available on every box); this test keeps the switch itself
working: the same calls under lo / hi / poison give the bytes of a plain run. """
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

_SCRIPT = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from homonim_amd import _hk
from oracle import oracle_np as onp
ctx = _hk.Context(0, n_streams=2)
md = hashlib.sha256()
for (h, w), model, k, thresh in [((300, 1003), 'gain', 5, None), ((260, 520), 'gain-blk-offset', 7, None),
                                 ((200, 300), 'gain-offset', 5, 0.9), ((517, 640), 'gain-offset', 15, None)]:
    src, ref = onp.synth_pair(h, w, seed=h + w, nodata_variant='frame+holes')
    desc = _hk.make_desc(model, (k, k), False, thresh, np.nan, np.nan)
    params, corr, norm, n_fail = ctx.fit_apply(desc, src, ref, 3 if thresh is not None else 2, want_params=True, want_corr=True)
    md.update(params.tobytes()), md.update(corr.tobytes()), md.update(norm.tobytes()), md.update(str(n_fail).encode())
# device-resident jobs: separate allocations per plane, statistics workspace re-allocated to fit every job exactly
n, B = 768, 2
bufs = {name: ctx.dev_alloc(4 * n * n * B) for name in ('src', 'ref', 'corr')}
nrm = ctx.dev_alloc(16 * B)
ctx.synth_fill_dev(bufs['src'], bufs['ref'], B, n, n, n, n * n, seed=5, nodata_variant=1, stream=0)
ctx.stream_sync(0)
desc = _hk.make_desc('gain-blk-offset', (15, 15), False, None, np.nan, np.nan)
for rows in (n, n - 40, n - 8):
    job = _hk.DevJob()
    job.src, job.ref, job.corr = bufs['src'], bufs['ref'], bufs['corr']
    job.gain = job.offset = job.r2 = job.fail_count = None
    job.norm = nrm
    job.n_bands, job.height, job.width, job.stride, job.band_stride = B, rows, n, n, n * n
    job.seg_rows, job.stream = 0, 1
    job.out_row0, job.out_col0, job.out_rows, job.out_cols = 0, 0, rows, n
    ctx.block_norm_dev(desc, job, nrm)
    ctx.fit_apply_dev(desc, job)
    ctx.stream_sync(1)
    out = np.empty((B, n, n), np.float32)
    ctx.d2h(out, bufs['corr'])
    md.update(out[:, :rows].tobytes())
for p in list(bufs.values()) + [nrm]:
    ctx.dev_free(p)
ctx.close()
print('digest', md.hexdigest())
''' % REPO


def _run(mode):
    env = dict(os.environ, PYTHONPATH=REPO)
    env.pop('HK_GUARD_ALLOC', None)
    if mode:
        env['HK_GUARD_ALLOC'] = mode
    run = subprocess.run([sys.executable, '-c', _SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, f'HK_GUARD_ALLOC={mode}: exit {run.returncode}\n{run.stderr[-2000:]}'
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith('digest ')]
    assert len(lines) == 1
    return lines[0]


def test_guarded_and_poisoned_allocations_change_nothing():
    plain = _run(None)
    for mode in ('lo', 'hi', 'poison'):
        assert _run(mode) == plain, f'HK_GUARD_ALLOC={mode} changed the results'
