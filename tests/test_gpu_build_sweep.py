"""
Every build of the fused kernel against the oracle (round 6; the launch ledger of tests/conftest.py showed that the standing suite
had launched 178 of the 332 builds).

The fused kernel `hk::fit_apply_kernel<MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB, BATCH>` exists in one build per model x R2
quantity set x kernel half-width (0..7 at compile time, four residues beyond) x nodata specialisation x ring mode x
certificate-only x lock-step workgroup x batched launch; which one a call gets is decided by hk_api.hip `fill_args` from the model
configuration, the kernel shape and the raster's nodata (thresholds measured on MI355X, moved every round) and by `launch_build` from
the LDS the shape needs.  The tests here walk that decision space -- every model configuration over a grid of kernel heights and
widths on either side of each threshold, through the three entry points that differ in builds (parameters out; corrected block
only = the RasterFuse path, certificate-only first; batched jobs) -- on a raster of two strips, and hold EVERY launch to the C
oracle.  The ring modes and segment policies a shape does not get by default are forced through the library's testing switches
(HK_USE_RING, HK_FORCE_GENERAL) in a second pass, so that builds only they reach are checked as well.
Reference shapes: homonim/utils.py:104-133 (any odd shape), tests/integration.py:32-43 (1x1, 5x5, 15x15, 31x31).
"""
import warnings

import numpy as np
import pytest

from homonim_amd import _hk
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu

H, W = 44, 300          # two strips (248 / 224 / ... output columns each), one row segment
HEIGHTS = (1, 3, 5, 7, 9, 11, 13, 15, 17, 39, 41)          # either side of: full ring <= 5 (7, 15), split ring 9-15, centre ring <= 39
WIDTHS = (1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23)       # compile-time half-widths 0..7, then every residue of kw // 2 mod 4


@pytest.fixture(scope='module')
def ctx():
    return _hk.default_context()


@pytest.fixture(scope='module')
def oc():
    from homonim_amd import build
    build.build_oracle(verbose=False)
    from oracle import oracle_c
    return oracle_c


def _same(got, exp, what):
    """ bit-exact float32 incl. the NaN pattern; <= 2 ulp on <= 1e-5 of the pixels (min. 2) for the order of the float64 sum of squares """
    assert got.shape == exp.shape, what
    nan_g, nan_e = np.isnan(got), np.isnan(exp)
    assert (nan_g == nan_e).all(), f'{what}: NaN pattern differs at {np.argwhere(nan_g != nan_e)[:3].tolist()}'
    ok = ~nan_e
    diff = ok & (got != exp)
    n = int(diff.sum())
    if n:
        g, e = got[diff], exp[diff]
        inf_same = np.isinf(g) & np.isinf(e) & (g == e)
        ulps = np.abs(g.view(np.int32).astype(np.int64) - e.view(np.int32).astype(np.int64))[~inf_same]
        worst = int(ulps.max()) if ulps.size else 0
        assert worst <= 2 and n <= max(2, 1e-5 * int(ok.sum())), f'{what}: {n} of {int(ok.sum())} differ, worst {worst} ulp'


def _pair(seed, nodata, numeric=None):
    src, ref = onp.synth_pair(H, W, seed=seed, nodata_variant='none' if nodata is None else 'frame+holes')
    if numeric is not None:   # the same holes, spelt as a number
        src, ref = np.where(np.isnan(src), np.float32(numeric), src), np.where(np.isnan(ref), np.float32(numeric), ref)
    return src, ref


CONFIGS = [  # (model, find_r2, thresh, nodata)
    ('gain', False, None, None), ('gain', False, None, np.nan), ('gain', True, None, None), ('gain', True, None, np.nan),
    ('gain-blk-offset', False, None, np.nan), ('gain-blk-offset', True, None, np.nan),
    ('gain-offset', False, None, None), ('gain-offset', False, None, np.nan), ('gain-offset', True, None, None),
    ('gain-offset', True, None, np.nan), ('gain-offset', False, 0.25, None), ('gain-offset', False, 0.25, np.nan),
    ('gain-offset', True, 0.25, None), ('gain-offset', True, 0.25, np.nan),
]


def _dev_job_corrected_only(ctx, desc, src, ref):
    """ hk_fit_apply_dev + hk_inpaint_dev of a device-resident job that keeps nothing but the corrected block and carries no scratch:
    always the certificate build + its list launch (hk_api.hip launch_fit) -> (raw failure count, corrected block) """
    h, w = src.shape
    stride = (w + 3) // 4 * 4
    pad = lambda a: np.ascontiguousarray(np.pad(a, ((0, 0), (0, stride - w))).astype(np.float32))  # noqa: E731
    d = {k: ctx.dev_alloc(4 * stride * h) for k in ('src', 'ref', 'corr')}
    d['fail'] = ctx.dev_alloc(8)
    try:
        ctx.h2d(d['src'], pad(src)), ctx.h2d(d['ref'], pad(ref))
        ctx.memset(d['fail'], 0, 8)
        job = _hk.DevJob()
        job.src, job.ref, job.corr, job.fail_count = d['src'], d['ref'], d['corr'], d['fail']
        job.gain = job.offset = job.r2 = job.norm = None
        job.n_bands, job.height, job.width, job.stride, job.band_stride = 1, h, w, stride, stride * h
        job.seg_rows, job.stream = 0, 0
        ctx.fit_apply_dev(desc, job)
        ctx.stream_sync(0)
        raw = np.zeros(1, np.uint64)
        ctx.d2h(raw, d['fail'])
        ctx.inpaint_dev(desc, job)
        ctx.stream_sync(0)
        corr = np.empty((h, stride), np.float32)
        ctx.d2h(corr, d['corr'])
        return int(raw[0]), corr[:, :w].copy()
    finally:
        for v in d.values():
            ctx.dev_free(v)


def _check_shape(ctx, oc, model, find_r2, thresh, nodata, kshape, src, ref, what):
    norm = oc.fit_block_norm(src, nodata, ref, nodata) if model == 'gain-blk-offset' else None
    if thresh is not None and kshape[0] in (3, 5, 9):
        # with the r2 mask (one height per ring mode: the in-painting branch is the expensive part of this sweep): a patch the fit
        # cannot explain, so that wave-rows stay open for the list launch, pixels fail and the in-painting branch runs -- in EVERY
        # width's certificate / list / complete build, not only on the shapes the other tests happen to use
        # (uncorrelated values, not a constant: a flat window has sstot = 0 exactly and its R2 is inf or NaN by the last bit of ssres)
        ref = ref.copy()
        ref[18:23, 90:131] = np.random.default_rng(5).uniform(-3, -1, (5, 41)).astype(np.float32)
    exp_p, exp_c, exp_fail = oc.fit_apply(model, src, nodata, ref, nodata, kshape, find_r2, thresh, norm_model=norm)
    desc = _hk.make_desc(model, kshape, find_r2, thresh, nodata, nodata)
    params, corr, _, n_fail = ctx.fit_apply(desc, src, ref, exp_p.shape[0], want_params=True, want_corr=True, norm_in=norm)
    _same(params, exp_p, f'{what}: params')
    _same(corr, exp_c, f'{what}: corrected')
    assert n_fail == (exp_fail if thresh is not None else 0), what
    if not find_r2:
        # corrected block only (RasterFuse without a parameter file): with the r2 mask the certificate-only build runs first
        _, corr_f, _, n_fail_f = ctx.fit_apply(desc, src, ref, exp_p.shape[0], want_params=False, want_corr=True, norm_in=norm)
        _same(corr_f, exp_c, f'{what}: corrected, fused')
        assert n_fail_f == (exp_fail if thresh is not None else 0), what
        if thresh is not None and kshape[0] in (3, 5, 9):
            assert exp_fail > 0, what
            raw, corr_d = _dev_job_corrected_only(ctx, desc, src, ref)      # certificate build + list launch, then the in-painting
            assert raw == exp_fail, (what, raw, exp_fail)
            _same(corr_d, exp_c, f'{what}: corrected, device job (certificate build + list launch)')


@pytest.mark.oracle
@pytest.mark.parametrize('model, find_r2, thresh, nodata', CONFIGS)
def test_every_kernel_shape_of_a_configuration_vs_oracle(ctx, oc, model, find_r2, thresh, nodata):
    """ The builds `fill_args` picks by itself: 11 heights x 12 widths per model configuration. """
    src, ref = _pair(7, nodata)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for kh in HEIGHTS:
            for kw in WIDTHS:
                if model == 'gain-offset' and kh * kw < 2:
                    continue
                _check_shape(ctx, oc, model, find_r2, thresh, nodata, (kh, kw), src, ref, f'{model} r2={find_r2} thresh={thresh} nodata={nodata} {kh}x{kw}')


@pytest.mark.oracle
@pytest.mark.parametrize('model, find_r2, thresh, nodata', CONFIGS)
def test_paired_builds_of_the_wide_kernels_vs_oracle(ctx, oc, model, find_r2, thresh, nodata):
    """ Kernels 31 and 33 - 37 wide (odd counts of whole lanes either side, hk_fit_kernel.h wide_pairs) run builds of their own on the
    centre ring, whose horizontal sums add the whole lanes as pairs (RW = -5 .. -8): every model configuration, every residue of
    kw // 2 mod 4, certificate-only / list / complete, dense and NaN-aware, with open wave-rows and failing pixels at 3 and 9 rows. """
    src, ref = _pair(17, nodata)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for kh in (3, 9, 17):
            for kw in (31, 33, 35, 37):
                _check_shape(ctx, oc, model, find_r2, thresh, nodata, (kh, kw), src, ref, f'paired: {model} r2={find_r2} thresh={thresh} nodata={nodata} {kh}x{kw}')


@pytest.mark.oracle
@pytest.mark.parametrize('ring', ['0', '1', '2', '3'])
@pytest.mark.parametrize('model, find_r2, thresh, nodata', CONFIGS)
def test_forced_ring_modes_of_every_width_vs_oracle(ctx, oc, model, find_r2, thresh, nodata, ring, monkeypatch):
    """ The ring modes a shape does not get by default (HK_USE_RING; the library ignores the switch where the mode has no build):
    full ring on tall kernels (one wave per workgroup: its LDS no longer fits four), centre ring on short ones, everything
    re-loaded from 9 wide, split ring from 7 rows -- every width, numeric nodata on the general builds. """
    numeric = -3.0 if (nodata is not None and model != 'gain-blk-offset') else None
    src, ref = _pair(11, nodata, numeric)
    nd = nodata if numeric is None else numeric
    heights = {'0': (5, 17), '1': (3, 11, 17), '2': (1, 5, 65), '3': (7, 11, 15)}[ring]
    monkeypatch.setenv('HK_USE_RING', ring)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for kh in heights:
            for kw in WIDTHS:
                if model == 'gain-offset' and kh * kw < 2:
                    continue
                _check_shape(ctx, oc, model, find_r2, thresh, nd, (kh, kw), src, ref, f'ring {ring}: {model} r2={find_r2} thresh={thresh} nodata={nd} {kh}x{kw}')


@pytest.mark.oracle
@pytest.mark.parametrize('model, find_r2, thresh', [('gain', False, None), ('gain', True, None), ('gain-offset', False, None),
                                                    ('gain-offset', True, None), ('gain-offset', False, 0.25), ('gain-offset', True, 0.25)])
def test_general_builds_on_rasters_without_nodata_vs_oracle(ctx, oc, model, find_r2, thresh, monkeypatch):
    """ HK_FORCE_GENERAL: the NaN-aware builds on rasters whose nodata is None (every row `clean`: their dense short cut all the way). """
    src, ref = _pair(13, None)
    monkeypatch.setenv('HK_FORCE_GENERAL', '1')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for kh in (3, 5, 9, 17):
            for kw in WIDTHS:
                _check_shape(ctx, oc, model, find_r2, thresh, None, (kh, kw), src, ref, f'general: {model} r2={find_r2} thresh={thresh} {kh}x{kw}')


@pytest.mark.oracle
@pytest.mark.parametrize('nodata_variant, ring', [(0, None), (1, None), (1, '1')])
def test_batched_builds_of_every_width_vs_oracle(oc, nodata_variant, ring, monkeypatch):
    """ hk_fit_apply_batch_dev: the job-table builds (gain-blk-offset without R2, the block loop of a resident mosaic) of every kernel
    width and ring mode the policy gives them -- and, forced (HK_USE_RING=1), the full ring on kernels up to 15 wide, four strips per
    workgroup while the LDS holds them (5 rows) and one beyond (11 rows) --, two jobs of different shapes per launch, every job
    against the oracle. """
    if ring is not None:
        monkeypatch.setenv('HK_USE_RING', ring)
    ctx = _hk.Context(0, n_streams=2)
    B = 2
    shapes = [(H, W), (H - 7, W - 36)]
    nd = np.nan if nodata_variant else None
    stride = (W + 3) // 4 * 4
    plane = stride * H
    bufs = [{k: ctx.dev_alloc(4 * plane * B) for k in ('src', 'ref', 'corr')} for _ in shapes]
    norm = ctx.dev_alloc(16 * B * len(shapes))
    try:
        host = []
        for j, (h, w) in enumerate(shapes):
            ctx.synth_fill_dev(bufs[j]['src'], bufs[j]['ref'], B, h, w, stride, plane, seed=300 + j, nodata_variant=nodata_variant, stream=0)
            ctx.stream_sync(0)
            s, r = np.empty((B, H, stride), np.float32), np.empty((B, H, stride), np.float32)
            ctx.d2h(s, bufs[j]['src']), ctx.d2h(r, bufs[j]['ref'])
            host.append((s, r))
        for kh in ((1, 5, 7, 9, 11, 15, 17, 41) if ring is None else (5, 11)):
            # (31 - 37 wide at two heights: the batched forms of the paired builds, test_paired_builds_of_the_wide_kernels_vs_oracle)
            for kw in WIDTHS + ((31, 33, 35, 37) if ring is None and kh in (5, 17) else ()):
                desc = _hk.make_desc('gain-blk-offset', (kh, kw), False, None, nd, nd)
                jobs = []
                for j, (h, w) in enumerate(shapes):
                    job = _hk.DevJob()
                    job.src, job.ref, job.corr = bufs[j]['src'], bufs[j]['ref'], bufs[j]['corr']
                    job.gain = job.offset = job.r2 = job.fail_count = None
                    job.norm = norm + 16 * B * j
                    job.n_bands, job.height, job.width, job.stride, job.band_stride = B, h, w, stride, plane
                    job.seg_rows, job.stream = 0, 1
                    ctx.memset(job.corr, 0, 4 * plane * B)
                    jobs.append(job)
                arr = ctx.job_array(jobs)
                ctx.block_norm_batch_dev(desc, arr, norm)
                ctx.fit_apply_batch_dev(desc, arr)
                ctx.stream_sync(1)
                norms = np.zeros((len(shapes), B, 2))
                ctx.d2h(norms, norm)
                for j, (h, w) in enumerate(shapes):
                    got = np.empty((B, H, stride), np.float32)
                    ctx.d2h(got, bufs[j]['corr'])
                    for b in range(B):
                        s, r = np.ascontiguousarray(host[j][0][b, :h, :w]), np.ascontiguousarray(host[j][1][b, :h, :w])
                        _, exp, _ = oc.fit_apply('gain-blk-offset', s, nd, r, nd, (kh, kw), False, None, norm_model=norms[j, b], want_params=False)
                        _same(np.ascontiguousarray(got[b, :h, :w]), exp, f'batched gain-blk-offset {kh}x{kw} job {j} band {b}')
    finally:
        for d in bufs:
            for p in d.values():
                ctx.dev_free(p)
        ctx.dev_free(norm)
        ctx.close()
