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


@pytest.mark.oracle
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


def test_a_view_over_two_registered_ranges_with_a_pageable_gap_is_staged():
    """ Direct copies are for arrays that ONE page-locked range of the library holds entirely.  A strided view whose first rows lie
    in one registered range and whose last rows lie in another, with pageable memory between them, used to pass (the first and the
    last byte were asked about) and reached hipMemcpy2DAsync as "pinned" -- the runtime then pins the gap on the fly, the path of
    the round-3 abort.  It must travel through the staging ring, and give the same bytes; the same view inside one registered
    range is copied directly.  (The suite runs with HK_ASSERT_PINNED=1: a direct copy of a pageable page fails its call.) """
    assert os.environ.get('HK_ASSERT_PINNED') == '1'
    ctx = _hk.default_context()
    page, part = 4096, 64 * 1024
    buf = np.zeros(3 * part + 2 * page, np.uint8)
    a0 = (-buf.ctypes.data) % page                        # first page boundary inside the buffer
    h, w, pitch = 48, 500, 4096                           # 16 rows per 64 KB part
    src_full, ref_full = onp.synth_pair(h, w, 77, 'frame+holes')
    view = np.ndarray((h, w), np.float32, buffer=buf, offset=a0, strides=(pitch, 4))
    view[:] = src_full
    assert view.ctypes.data % page == 0 and view.strides == (pitch, 4)
    desc = _hk.make_desc('gain-offset', (5, 5), False, 0.25, np.nan, np.nan)
    _, exp, _, _ = ctx.fit_apply(desc, np.ascontiguousarray(view), ref_full, 2, want_params=False, want_corr=True)
    first, last = buf[a0:a0 + part], buf[a0 + 2 * part:a0 + 3 * part]
    ctx.pin(first), ctx.pin(last)
    try:
        _hk.staging_counters(reset=True)
        _, got, _, _ = ctx.fit_apply(desc, view, ref_full, 2, want_params=False, want_corr=True)
        direct, staged = _hk.staging_counters(reset=True)
        assert direct == 0 and staged > 0, (direct, staged)      # src (the gap view) and ref, corr (plain numpy) all staged
        assert np.array_equal(got, exp, equal_nan=True)
        # the rows inside ONE registered range: copied directly
        _, got16, _, _ = ctx.fit_apply(desc, view[:16], ref_full[:16], 2, want_params=False, want_corr=True)
        direct, staged = _hk.staging_counters(reset=True)
        assert direct >= 1, (direct, staged)
        _, exp16, _, _ = ctx.fit_apply(desc, np.ascontiguousarray(view[:16]), ref_full[:16], 2, want_params=False, want_corr=True)
        assert np.array_equal(got16, exp16, equal_nan=True)
    finally:
        ctx.unpin(first), ctx.unpin(last)
    # unregistered again: nothing of the buffer is copied directly any more
    _hk.staging_counters(reset=True)
    ctx.fit_apply(desc, view[:16], ref_full[:16], 2, want_params=False, want_corr=True)
    assert _hk.staging_counters()[0] == 0


def test_an_error_between_the_copies_leaves_no_pointer_behind(monkeypatch):
    """ A host-pointer call that fails after it queued results for unpacking must not leave the caller's output pointer in the
    staging ring: the next call on that stream would unpack up to 8 MB into memory the caller has usually freed by then (round-4
    advisor finding).  hk_debug_fail_after_d2h makes hk_fit_apply fail exactly there; afterwards the failed call's output array
    must stay as the caller left it, whatever runs on the context's streams. """
    ctx = _hk.Context(0, n_streams=1)   # one stream: the next call takes the failed call's slot
    try:
        src, ref = onp.synth_pair(300, 500, 3, 'none')
        desc = _hk.make_desc('gain', (5, 5), False, None, None, None)
        _, exp, _, _ = ctx.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True)
        victim = np.full(src.shape, -1.0, np.float32)
        _hk.debug_fail_after_d2h(True)
        try:
            with pytest.raises(Exception, match='hk_debug_fail_after_d2h'):
                ctx.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True, out_corr=victim)
        finally:
            _hk.debug_fail_after_d2h(False)
        victim[:] = -2.0   # "freed and re-used"
        for _ in range(6):   # more calls than the ring has chunks
            _, got, _, _ = ctx.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True)
            assert np.array_equal(got, exp, equal_nan=True)
        assert (victim == -2.0).all(), 'a later call unpacked the failed call\'s result into its output array'
    finally:
        ctx.close()


def test_an_error_behind_direct_copies_drains_them_before_the_call_returns(monkeypatch):
    """ The same failure with PAGE-LOCKED caller arrays (RasterFuse's default: rasters registered for the block loop), whose copies
    are queued straight on the caller's memory: when the failing call returns, no DMA may still be in flight on those arrays -- the
    caller unregisters and frees them next (round-5 advisor finding: stage_abandon() only looked at staged chunks).  Observable:
    the result copy was queued before the injected failure, so once the call has returned the output array holds the complete
    result -- its LAST row included, checked first, at once -- and nothing writes it afterwards. """
    ctx = _hk.Context(0, n_streams=1)
    try:
        h, w = 4096, 4096                              # 64 MB per plane: the result copy alone takes more than a millisecond
        src, ref = onp.synth_pair(h, w, 3, 'none')
        desc = _hk.make_desc('gain', (5, 5), False, None, None, None)
        _, exp, _, _ = ctx.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True)
        ps, pr, victim = ctx.pinned_empty((h, w)), ctx.pinned_empty((h, w)), ctx.pinned_empty((h, w))
        ps[:], pr[:], victim[:] = src, ref, -1.0
        _hk.staging_counters(reset=True)
        _hk.debug_fail_after_d2h(True)
        try:
            with pytest.raises(Exception, match='hk_debug_fail_after_d2h'):
                ctx.fit_apply(desc, ps, pr, 2, want_params=False, want_corr=True, out_corr=victim)
            last_row_done = np.array_equal(np.array(victim[-1]), exp[-1], equal_nan=True)
        finally:
            _hk.debug_fail_after_d2h(False)
        direct, staged = _hk.staging_counters()
        assert direct == 3 and staged == 0, (direct, staged)     # the three arrays went the direct way
        assert last_row_done, 'the failed call returned while its result copy was still in flight on the caller\'s array'
        assert np.array_equal(np.array(victim), exp, equal_nan=True)
        victim[:] = -2.0
        for _ in range(3):
            _, got, _, _ = ctx.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True)
        assert (np.array(victim) == -2.0).all()
    finally:
        ctx.close()
