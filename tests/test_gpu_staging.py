"""
The library's pinned host staging (hk_api.hip stage_h2d / stage_d2h): pageable caller memory is packed into page-locked chunks
and moved with contiguous hipMemcpyAsync copies; the HIP runtime never sees a pageable pointer.  The chunked paths (blocks
larger than a chunk, rows split over chunks, the ring wrapping round) must give the same bytes as a single chunk.
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from homonim_amd import _hk
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, 'tests', '_staging_worker.py')


def _run_worker(tmp_path, name, chunk_kb=None):
    env = dict(os.environ)
    env.pop('HK_STAGE_CHUNK_KB', None)
    if chunk_kb is not None:
        env['HK_STAGE_CHUNK_KB'] = str(chunk_kb)
    out = str(tmp_path / f'{name}.npz')
    res = subprocess.run([sys.executable, WORKER, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stdout
    return np.load(out)


def test_chunked_staging_gives_the_same_bytes(tmp_path):
    """ 8 MB chunks (everything in one) against 4 KB and 64 KB chunks (one or a few device rows per chunk). """
    ref = _run_worker(tmp_path, 'default')
    assert int(ref['blob_equal'][0]) == 1
    for kb in (4, 64):
        got = _run_worker(tmp_path, f'chunk{kb}', kb)
        assert sorted(got.files) == sorted(ref.files)
        for key in ref.files:
            a, b = ref[key], got[key]
            assert a.dtype == b.dtype and a.shape == b.shape, key
            assert np.array_equal(a, b, equal_nan=(a.dtype.kind == 'f')), f'{key} differs with {kb} KB chunks'


def test_staged_results_match_the_oracle(tmp_path):
    """ ... and they are the right bytes: the staged gain-offset call against the oracle, bit for bit. """
    got = _run_worker(tmp_path, 'oracle', 16)
    src, ref = onp.synth_pair(333, 517, 9, 'frame+holes')
    exp_params, n_fail = onp.fit_gain_offset(src, np.nan, ref, np.nan, (5, 5), True, 0.25)
    assert int(got['go_nf'][0]) == n_fail
    assert np.array_equal(got['go_params'], exp_params, equal_nan=True)
    assert np.array_equal(got['go_corr'], onp.apply(src, exp_params), equal_nan=True)


def test_small_mask_rows_many_times():
    """ The call the round-3 driver run died in -- hk_partial_mask with the uint8 mask on 20 x 10 / 40 x 20 rasters, fresh
    pageable numpy arrays every time -- a few thousand times, from four threads. """
    ctx = _hk.default_context()
    errors = []

    def loop(seed):
        rng = np.random.default_rng(seed)
        try:
            for i in range(1500):
                h, w = ((20, 10), (40, 20))[i & 1]
                a = rng.random((h, w), dtype=np.float32)
                a[rng.random((h, w)) < 0.05] = np.nan
                par = rng.random((2, h, w), dtype=np.float32)
                _, _, m = ctx.partial_mask(a, np.nan, par, (3, 3), want_mask=True)
                valid = ~np.isnan(a)
                # erosion by 5 x 5 with a zero border
                pad = np.zeros((h + 4, w + 4), bool)
                pad[2:-2, 2:-2] = valid
                exp = np.ones((h, w), bool)
                for dy in range(5):
                    for dx in range(5):
                        exp &= pad[dy:dy + h, dx:dx + w]
                if not np.array_equal(m.astype(bool), exp):
                    errors.append((seed, i))
                    return
        except Exception as ex:  # noqa: BLE001
            errors.append((seed, repr(ex)))

    threads = [threading.Thread(target=loop, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_unpinned_count_buffers_are_refused():
    """ hk_fail_counts_async writes into caller memory asynchronously: it insists on page-locked memory. """
    ctx = _hk.default_context()
    h, w, stride = 64, 64, 64
    d = ctx.dev_alloc(4 * h * stride * 4 + 64)
    try:
        job = _hk.DevJob()
        job.src, job.ref, job.corr = d, d + h * stride * 4, d + 2 * h * stride * 4
        job.fail_count = d + 4 * h * stride * 4
        job.n_bands, job.height, job.width, job.stride, job.band_stride, job.stream = 1, h, w, stride, 0, 0
        ev = ctx.event()
        with pytest.raises(ValueError, match='page-locked'):
            ctx.fail_counts_async(job, np.zeros(1, np.uint64), ev)
        ctx.event_destroy(ev)
    finally:
        ctx.dev_free(d)
