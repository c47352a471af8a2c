""" Batched device launches (hk_block_norm_batch_dev / hk_fit_apply_batch_dev / hk_fail_counts_batch_async): many device-resident
jobs as one launch per kernel stage.  The contract is "bit-identical to the per-job calls" -- a job's statistics and wave units do
not depend on the launch it travels in -- so every test runs the same jobs both ways and compares bytes; one of them also goes to
the oracle. """
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from homonim_amd import _hk  # noqa: E402


@pytest.fixture(scope='module')
def ctx():
    c = _hk.Context(0, n_streams=4)
    yield c
    c.close()


@pytest.fixture(scope='module')
def oracle():
    from oracle import oracle_c
    return oracle_c


def _blocks_of(n, B, k, max_mem):
    from homonim_amd import utils
    from homonim_amd.fuse import block_pairs
    overlap = utils.overlap_for_kernel((k, k))
    return [bp for bp in block_pairs((n, n), B, overlap, max_mem) if bp.band_i == 0]


def _block_jobs(positions, bufs, norm, n, plane, B, stream):
    jobs = []
    for i, bp in enumerate(positions):
        wi, wo = bp.src_in_block, bp.src_out_block
        off = 4 * (wi.row_off * n + wi.col_off)
        job = _hk.DevJob()
        job.src, job.ref, job.corr = bufs['src'] + off, bufs['ref'] + off, bufs['corr'] + off
        job.gain = job.offset = job.r2 = job.fail_count = None
        job.norm = norm + 16 * B * i
        job.n_bands, job.height, job.width, job.stride, job.band_stride = B, wi.height, wi.width, n, plane
        job.seg_rows, job.stream = 0, stream
        job.out_row0, job.out_col0 = wo.row_off - wi.row_off, wo.col_off - wi.col_off
        job.out_rows, job.out_cols = wo.height, wo.width
        jobs.append(job)
    return jobs


# kernels whose halo (utils.overlap_for_kernel) is a multiple of 4 pixels: a block processed in place starts 16-byte aligned
@pytest.mark.parametrize('model,k,nodata_variant', [('gain-blk-offset', 15, 0), ('gain-blk-offset', 7, 1), ('gain', 7, 0),
                                                    ('gain', 15, 0), ('gain-offset', 7, 2), ('gain-blk-offset', 23, 1), ('gain-blk-offset', 31, 0)])   # (23 / 31: the batched builds of kernels wider than 15)
@pytest.mark.oracle
def test_blocks_of_a_raster_in_one_launch_equal_one_launch_per_block(ctx, oracle, model, k, nodata_variant):
    """ A 3-band 1536 x 1536 raster cut into the reference's blocks (raster_pair.py:342-428; edge blocks differ in shape and
    store window from interior ones): statistics and corrected planes of the batched launches == those of per-block launches,
    byte for byte; one window against the oracle. """
    n, B = 1536, 3
    plane = n * n
    positions = _blocks_of(n, B, k, 1)   # 1 MB blocks: 4 x 4 positions; corner, edge and interior blocks differ in shape
    assert len(positions) == 16
    shapes = {(bp.src_in_block.height, bp.src_in_block.width) for bp in positions}
    assert len(shapes) >= 2, 'the partition should mix block shapes'
    bufs = {name: ctx.dev_alloc(4 * plane * B) for name in ('src', 'ref', 'corr', 'corr2')}
    norm = ctx.dev_alloc(16 * B * len(positions))
    norm2 = ctx.dev_alloc(16 * B * len(positions))
    try:
        nd = np.nan if nodata_variant in (1, 2) else None
        ctx.synth_fill_dev(bufs['src'], bufs['ref'], B, n, n, n, plane, seed=77, nodata_variant=nodata_variant, stream=0)
        ctx.memset(bufs['corr'], 0xff, 4 * plane * B)
        ctx.memset(bufs['corr2'], 0xff, 4 * plane * B)
        ctx.stream_sync(0)
        desc = _hk.make_desc(model, (k, k), False, None, nd, nd)
        one = _block_jobs(positions, bufs, norm, n, plane, B, stream=1)
        for job in one:
            if model == 'gain-blk-offset':
                ctx.block_norm_dev(desc, job, job.norm)
            ctx.fit_apply_dev(desc, job)
        ctx.stream_sync(1)
        bufs_b = dict(bufs, corr=bufs['corr2'])
        many = _block_jobs(positions, bufs_b, norm2, n, plane, B, stream=2)
        arr = ctx.job_array(many)
        for _ in range(2):   # the second round re-uses the stream's table ring
            if model == 'gain-blk-offset':
                ctx.block_norm_batch_dev(desc, arr, norm2)
            ctx.fit_apply_batch_dev(desc, arr)
        ctx.stream_sync(2)
        a, b = np.empty((B, n, n), np.float32), np.empty((B, n, n), np.float32)
        ctx.d2h(a, bufs['corr']), ctx.d2h(b, bufs['corr2'])
        assert not np.isnan(a).all()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), 'batched corrected planes differ from per-job launches'
        if model == 'gain-blk-offset':
            na, nb = np.zeros((len(positions), B, 2)), np.zeros((len(positions), B, 2))
            ctx.d2h(na, norm), ctx.d2h(nb, norm2)
            assert np.array_equal(na.view(np.uint64), nb.view(np.uint64)), 'batched statistics differ from per-job launches'
        # one interior window of the last block position against the oracle (the block is a stand-alone raster to the reference)
        pi, band = len(positions) - 1, B - 1
        wi, wo = positions[pi].src_in_block, positions[pi].src_out_block
        s_all, r_all = np.empty((n, n), np.float32), np.empty((n, n), np.float32)
        ctx.d2h(s_all, bufs['src'] + 4 * plane * band), ctx.d2h(r_all, bufs['ref'] + 4 * plane * band)
        blk = (slice(wi.row_off, wi.row_off + wi.height), slice(wi.col_off, wi.col_off + wi.width))
        s, t = np.ascontiguousarray(s_all[blk]), np.ascontiguousarray(r_all[blk])
        nm = nb[pi, band] if model == 'gain-blk-offset' else None
        _, exp, _ = oracle.fit_apply(model, s, nd, t, nd, (k, k), False, None, norm_model=nm, want_params=False)
        oy, ox = wo.row_off - wi.row_off, wo.col_off - wi.col_off
        got = b[band][blk][oy:oy + wo.height, ox:ox + wo.width]
        e = exp[oy:oy + wo.height, ox:ox + wo.width]
        same = (e.view(np.uint32) == got.view(np.uint32)) | (np.isnan(e) & np.isnan(got))
        assert same.mean() > 1 - 1e-5, (model, k, 1 - same.mean())
    finally:
        for ptr in bufs.values():
            ctx.dev_free(ptr)
        ctx.dev_free(norm), ctx.dev_free(norm2)


def test_tiles_with_r2_threshold_in_one_launch(ctx):
    """ Independent tiles, gain-offset with r2_inpaint_thresh (configs[4]'s shape of work): one batched launch + one batched counter
    fetch + per-tile in-painting == per-tile launches.  Tile 2 has a noisy reference (pixels fail the r2 mask and are
    in-painted); the others are clean. """
    n, B, T, k = 512, 2, 4, 5
    plane = n * n
    desc = _hk.make_desc('gain-offset', (k, k), False, 0.25, None, None)
    fail = ctx.dev_alloc(8 * B * T)
    ctx.memset(fail, 0, 8 * B * T)
    tiles = []
    try:
        for t in range(T):
            d = {name: ctx.dev_alloc(4 * plane * B) for name in ('src', 'ref', 'corr', 'corr2')}
            ctx.synth_fill_dev(d['src'], d['ref'], B, n, n, n, plane, seed=900 + t, nodata_variant=3 if t == 2 else 0, stream=0)
            tiles.append(d)
        ctx.stream_sync(0)

        def jobs_for(out_name, stream):
            jobs = []
            for t, d in enumerate(tiles):
                job = _hk.DevJob()
                job.src, job.ref, job.corr = d['src'], d['ref'], d[out_name]
                job.gain = job.offset = job.r2 = job.norm = None
                job.fail_count = fail + 8 * B * t
                job.n_bands, job.height, job.width, job.stride, job.band_stride = B, n, n, n, plane
                job.seg_rows, job.stream = 0, stream
                jobs.append(job)
            return jobs

        fails_one = 0
        for job in jobs_for('corr', 1):
            ctx.fit_apply_dev(desc, job)
            fails_one += ctx.inpaint_dev(desc, job)
        ctx.stream_sync(1)

        many = jobs_for('corr2', 2)
        arr = ctx.job_array(many)
        counts = ctx.pinned_empty((B * T,), np.uint64)
        ev = ctx.event()
        ctx.fit_apply_batch_dev(desc, arr)
        ctx.fail_counts_batch_async(arr, counts, ev)
        ctx.event_sync(ev)
        c_all = counts.copy()
        fails_many = 0
        for t, job in enumerate(many):
            c = c_all[B * t:B * (t + 1)]
            if ctx.counts_pending(c):
                fails_many += ctx.inpaint_dev_counts(desc, job, c)
        ctx.stream_sync(2)
        ctx.event_destroy(ev)
        assert fails_one == fails_many and fails_many > 0
        for t, d in enumerate(tiles):
            a, b = np.empty((B, n, n), np.float32), np.empty((B, n, n), np.float32)
            ctx.d2h(a, d['corr']), ctx.d2h(b, d['corr2'])
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f'tile {t}'
    finally:
        for d in tiles:
            for ptr in d.values():
                ctx.dev_free(ptr)
        ctx.dev_free(fail)


def test_batch_argument_checks(ctx):
    n, B = 256, 1
    plane = n * n
    d = {name: ctx.dev_alloc(4 * plane) for name in ('src', 'ref', 'corr')}
    try:
        def job(stream=0, corr=True):
            j = _hk.DevJob()
            j.src, j.ref, j.corr = d['src'], d['ref'], d['corr'] if corr else None
            j.gain = j.offset = j.r2 = j.fail_count = j.norm = None
            j.n_bands, j.height, j.width, j.stride, j.band_stride = B, n, n, n, plane
            j.seg_rows, j.stream = 0, stream
            return j
        desc = _hk.make_desc('gain', (5, 5), False, None, None, None)
        with pytest.raises(ValueError, match='share one stream'):
            ctx.fit_apply_batch_dev(desc, [job(0), job(1)])
        with pytest.raises(ValueError, match='same set of outputs'):
            ctx.fit_apply_batch_dev(desc, [job(0), job(0, corr=False)])
        blk = _hk.make_desc('gain-blk-offset', (5, 5), False, None, None, None)
        with pytest.raises(ValueError, match='needs job->norm'):
            ctx.fit_apply_batch_dev(blk, [job(0)])
        with pytest.raises(ValueError):
            ctx.fit_apply_batch_dev(desc, (_hk.DevJob * 0)())
        ctx.fit_apply_batch_dev(desc, [job(0)])   # a batch of one is fine
        ctx.stream_sync(0)
    finally:
        for ptr in d.values():
            ctx.dev_free(ptr)


def test_stream_wait_event_orders_two_streams(ctx):
    """ hk_stream_wait_event: a job on stream 2 that reads what a job on stream 1 writes, ordered on the device by an event
    (no host wait in between), equals the same two jobs on one stream. """
    n, B = 1024, 2
    plane = n * n
    d = {name: ctx.dev_alloc(4 * plane * B) for name in ('src', 'ref', 'mid', 'out', 'mid1', 'out1')}
    try:
        ctx.synth_fill_dev(d['src'], d['ref'], B, n, n, n, plane, seed=5, nodata_variant=0, stream=0)
        ctx.stream_sync(0)
        desc = _hk.make_desc('gain', (5, 5), False, None, None, None)

        def job(src, out, stream):
            j = _hk.DevJob()
            j.src, j.ref, j.corr = src, d['ref'], out
            j.gain = j.offset = j.r2 = j.fail_count = j.norm = None
            j.n_bands, j.height, j.width, j.stride, j.band_stride = B, n, n, n, plane
            j.seg_rows, j.stream = 0, stream
            return j
        ctx.fit_apply_dev(desc, job(d['src'], d['mid1'], 3))
        ctx.fit_apply_dev(desc, job(d['mid1'], d['out1'], 3))
        ctx.stream_sync(3)
        ev = ctx.event()
        for _ in range(3):
            ctx.fit_apply_dev(desc, job(d['src'], d['mid'], 1))
            ctx.event_record(ev, 1)
            ctx.stream_wait_event(2, ev)
            ctx.fit_apply_dev(desc, job(d['mid'], d['out'], 2))
        ctx.stream_sync(2)
        ctx.event_destroy(ev)
        a, b = np.empty((B, n, n), np.float32), np.empty((B, n, n), np.float32)
        ctx.d2h(a, d['out1']), ctx.d2h(b, d['out'])
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        for ptr in d.values():
            ctx.dev_free(ptr)


def test_jobs_of_different_band_counts_and_shapes_in_one_launch(ctx):
    """ Three unrelated rasters (1, 3 and 2 bands; different heights / widths / strides, one narrower than a strip) as one
    gain-blk-offset batch == three separate calls: statistics and corrected planes byte for byte. """
    shapes = [(1, 300, 1003, 1004), (3, 517, 640, 640), (2, 64, 100, 128)]   # bands, height, width, stride
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, np.nan, np.nan)
    total_bands = sum(s[0] for s in shapes)
    norm_a, norm_b = ctx.dev_alloc(16 * total_bands), ctx.dev_alloc(16 * total_bands)
    rasters = []
    try:
        b0 = 0
        one, many = [], []
        for i, (B, h, w, stride) in enumerate(shapes):
            plane = h * stride
            d = {name: ctx.dev_alloc(4 * plane * B) for name in ('src', 'ref', 'corr_a', 'corr_b')}
            rasters.append((d, B, h, w, stride))
            ctx.synth_fill_dev(d['src'], d['ref'], B, h, w, stride, plane, seed=40 + i, nodata_variant=1, stream=0)
            ctx.memset(d['corr_a'], 0xff, 4 * plane * B), ctx.memset(d['corr_b'], 0xff, 4 * plane * B)
            for lst, out, norm, stream in ((one, 'corr_a', norm_a, 1), (many, 'corr_b', norm_b, 2)):
                j = _hk.DevJob()
                j.src, j.ref, j.corr = d['src'], d['ref'], d[out]
                j.gain = j.offset = j.r2 = j.fail_count = None
                j.norm = norm + 16 * b0
                j.n_bands, j.height, j.width, j.stride, j.band_stride = B, h, w, stride, plane
                j.seg_rows, j.stream = 0, stream
                lst.append(j)
            b0 += B
        ctx.stream_sync(0)
        for j in one:
            ctx.block_norm_dev(desc, j, j.norm)
            ctx.fit_apply_dev(desc, j)
        ctx.stream_sync(1)
        arr = ctx.job_array(many)
        ctx.block_norm_batch_dev(desc, arr, norm_b)
        ctx.fit_apply_batch_dev(desc, arr)
        ctx.stream_sync(2)
        na, nb = np.zeros((total_bands, 2)), np.zeros((total_bands, 2))
        ctx.d2h(na, norm_a), ctx.d2h(nb, norm_b)
        assert np.isfinite(na).all() and np.array_equal(na.view(np.uint64), nb.view(np.uint64))
        for d, B, h, w, stride in rasters:
            a, b = np.empty((B, h, stride), np.float32), np.empty((B, h, stride), np.float32)
            ctx.d2h(a, d['corr_a']), ctx.d2h(b, d['corr_b'])
            assert np.isfinite(a[:, 8:-8, 8:w - 8]).any()
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (B, h, w)
    finally:
        for d, *_ in rasters:
            for ptr in d.values():
                ctx.dev_free(ptr)
        ctx.dev_free(norm_a), ctx.dev_free(norm_b)


@pytest.mark.oracle
def test_statistics_batch_with_a_smaller_plane_that_has_more_chunks(ctx, oracle):
    """ The streaming pass of the statistics strides every plane by its OWN wave count, which follows the number of 1 KB
    chunks (rows x ceil(ceil(w / 4) / 64)), not the number of pixels: 1090 x 1025 has fewer pixels than 1100 x 1024 but 5
    instead of 4 chunks per row.  The launch's grid must cover the plane with the most chunks (round-3 advice: it was sized
    from the plane with the most pixels, and the other plane's last waves never ran). """
    shapes = [(1100, 1024), (1090, 1025), (64, 4000)]
    strides = [(w + 3) // 4 * 4 for _, w in shapes]
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, None, None)
    bufs, jobs_one, jobs_many = [], [], []
    norm1, norm2 = ctx.dev_alloc(16 * len(shapes)), ctx.dev_alloc(16 * len(shapes))
    host = []
    try:
        for i, ((h, w), st) in enumerate(zip(shapes, strides)):
            s, r = ctx.dev_alloc(4 * h * st), ctx.dev_alloc(4 * h * st)
            bufs += [s, r]
            ctx.synth_fill_dev(s, r, 1, h, w, st, h * st, seed=5 + i, nodata_variant=0, stream=0)
            for jobs, norm, stream in ((jobs_one, norm1, 1), (jobs_many, norm2, 2)):
                job = _hk.DevJob()
                job.src, job.ref, job.norm = s, r, norm + 16 * i
                job.n_bands, job.height, job.width, job.stride, job.band_stride, job.stream = 1, h, w, st, h * st, stream
                jobs.append(job)
        ctx.stream_sync(0)
        for job in jobs_one:
            ctx.block_norm_dev(desc, job, job.norm)
        ctx.stream_sync(1)
        ctx.block_norm_batch_dev(desc, ctx.job_array(jobs_many), norm2)
        ctx.stream_sync(2)
        a, b = np.zeros((len(shapes), 2)), np.zeros((len(shapes), 2))
        ctx.d2h(a, norm1), ctx.d2h(b, norm2)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (a, b)
        for i, ((h, w), st) in enumerate(zip(shapes, strides)):   # ... and they are the oracle's statistics
            s, r = np.empty((h, st), np.float32), np.empty((h, st), np.float32)
            ctx.d2h(s, bufs[2 * i]), ctx.d2h(r, bufs[2 * i + 1])
            exp = oracle.fit_block_norm(np.ascontiguousarray(s[:, :w]), None, np.ascontiguousarray(r[:, :w]), None)
            assert np.allclose(b[i], exp, rtol=2e-6, atol=1e-9), (i, b[i], exp)
    finally:
        for p in bufs:
            ctx.dev_free(p)
        ctx.dev_free(norm1), ctx.dev_free(norm2)


@pytest.mark.parametrize('nodata_variant, thresh', [(0, 0.25), (2, 0.25), (3, 0.25), (0, None)])
def test_tiles_of_gain_offset_in_one_launch_equal_one_launch_per_tile(ctx, nodata_variant, thresh):
    """ hk_fit_apply_batch_dev for gain-offset with the R2 work (a mosaic of tiles, BASELINE configs[4]): six tiles of different shapes
    and band counts through the batched entry point -- certificate-only build first when nothing is expected to fail, the complete
    build with the in-painting's inputs in the jobs' scratch otherwise -- give the corrected planes of six separate calls, byte for
    byte; variant 3 (noisy reference: a third of the pixels fail) also runs the in-painting branch from both.  (The entry point runs
    this model as one launch per job: a build with the job-table look-up was measured in round 5 -- one launch for the 64 tiles of
    configs[4] 12.75-12.9 ms against 12.3 for 64 launches on four streams -- and not kept, HISTORY.md item 50.) """
    shapes = [(2, 300, 1003, 1004), (4, 517, 640, 640), (1, 64, 100, 128), (3, 256, 256, 256), (2, 33, 4099, 4100), (1, 700, 52, 64)]
    desc = _hk.make_desc('gain-offset', (5, 5), False, thresh, np.nan if nodata_variant == 2 else None, np.nan if nodata_variant == 2 else None)
    total_bands = sum(sh[0] for sh in shapes)
    fail_a, fail_b = ctx.dev_alloc(8 * total_bands), ctx.dev_alloc(8 * total_bands)
    ctx.memset(fail_a, 0, 8 * total_bands), ctx.memset(fail_b, 0, 8 * total_bands)
    rasters = []
    try:
        b0, one, many = 0, [], []
        for i, (B, h, w, stride) in enumerate(shapes):
            plane = h * stride
            d = {name: ctx.dev_alloc(4 * plane * B) for name in ('src', 'ref', 'corr_a', 'corr_b')}
            rasters.append((d, B, h, w, stride))
            ctx.synth_fill_dev(d['src'], d['ref'], B, h, w, stride, plane, seed=90 + i, nodata_variant=nodata_variant, stream=0)
            ctx.memset(d['corr_a'], 0xff, 4 * plane * B), ctx.memset(d['corr_b'], 0xff, 4 * plane * B)
            for lst, out, fail, stream in ((one, 'corr_a', fail_a, 1), (many, 'corr_b', fail_b, 2)):
                j = _hk.DevJob()
                j.src, j.ref, j.corr = d['src'], d['ref'], d[out]
                j.gain = j.offset = j.r2 = j.norm = None
                j.fail_count = fail + 8 * b0 if thresh is not None else None
                j.n_bands, j.height, j.width, j.stride, j.band_stride = B, h, w, stride, plane
                j.seg_rows, j.stream = 0, stream
                if thresh is not None:
                    j.scratch_bytes = ctx.job_scratch_bytes(j)
                    d[f'scratch_{out}'] = ctx.dev_alloc(j.scratch_bytes)
                    j.scratch = d[f'scratch_{out}']
                lst.append(j)
            b0 += B
        ctx.stream_sync(0)
        for rnd in range(2):   # the second round starts from what the first one learnt (certificate-only or not)
            for j in one:
                ctx.fit_apply_dev(desc, j)
            arr = ctx.job_array(many)
            ctx.fit_apply_batch_dev(desc, arr)
            ctx.stream_sync(1), ctx.stream_sync(2)
            counts = []
            for fail, jobs in ((fail_a, one), (fail_b, many)):
                c = np.zeros(total_bands, np.uint64)
                ctx.d2h(c, fail)
                counts.append(c.copy())
                if thresh is not None:   # the host's look at the counters: re-runs of the lighter build, the in-painting branch
                    k = 0
                    for j in jobs:
                        cj = c[k:k + j.n_bands].copy()
                        if ctx.counts_pending(cj):
                            ctx.inpaint_dev_counts(desc, j, cj)
                        k += j.n_bands
                    ctx.stream_sync(jobs[0].stream)
                ctx.memset(fail, 0, 8 * total_bands)
            if thresh is not None:
                # a band's count is void where its history asked for the re-run (HK_COUNT_RETRY: certificate-only build); wherever
                # BOTH histories counted, they counted the same pixels
                both = ((counts[0] | counts[1]) >> np.uint64(63)) == 0
                assert np.array_equal(counts[0][both], counts[1][both]), (rnd, counts)
            for d, B, h, w, stride in rasters:
                a, b = np.empty((B, h, stride), np.float32), np.empty((B, h, stride), np.float32)
                ctx.d2h(a, d['corr_a']), ctx.d2h(b, d['corr_b'])
                assert np.array_equal(a[:, :, :w].view(np.uint32), b[:, :, :w].view(np.uint32)), (rnd, B, h, w)
                assert not np.isnan(a[:, :, :w]).all()
    finally:
        for d, *_ in rasters:
            for ptr in d.values():
                ctx.dev_free(ptr)
        ctx.dev_free(fail_a), ctx.dev_free(fail_b)
