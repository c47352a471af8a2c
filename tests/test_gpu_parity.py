"""
GPU parity tests (run on the MI355X box: ``pytest -m gpu``).  Everything goes through the C ABI (ctypes ->
libhomonim_hk.so -> HIP kernels); the oracle is only the checker.

Bars (DESIGN.md "Numerics contract"):
* gain / gain-offset parameters, R2 and corrected output: BIT-EXACT float32 against the reference goldens and the
  numpy oracle, except that results which depend on the float64 sum of squares (whose 25..225-term float64 sum is
  order dependent in its last bit) may differ in <= 1e-5 of the pixels by <= 2 ulp.  The stated acceptance tolerance of
  the north star is 1e-5 relative; these tests hold the kernels to ~1e-7.
* gain-blk-offset: same, given the same block normalisation; the normalisation itself (exact float64 statistics on
  the GPU vs numpy's float32 pairwise ones) within 2e-6 relative.
"""
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN_CASES, assert_same_f32, case_id
from homonim_amd import Affine, CRS, KernelModel, Model, RasterArray, RefSpaceModel, SrcSpaceModel, _hk
from homonim_amd.enums import Resampling
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    c = _hk.default_context()
    c.selftest()
    return c


@pytest.fixture(scope='module')
def oc():
    """ the C oracle (bit-identical to oracle_np, tests/test_oracle_c.py) for inputs too large for Python loops """
    from homonim_amd import build
    build.build_oracle(verbose=False)
    from oracle import oracle_c
    return oracle_c


def _ra(arr, nodata):
    return RasterArray(arr, CRS(), Affine.identity(), nodata=nodata)


def assert_close_ulp(actual, expected, what, max_ulp=2, max_frac=1e-5):
    """ Bit-exact except for at most ``max_frac`` of the elements, which may be off by <= ``max_ulp`` float32 ulps. """
    actual, expected = np.asarray(actual), np.asarray(expected)
    assert actual.shape == expected.shape and actual.dtype == expected.dtype == np.float32, what
    nan_a, nan_e = np.isnan(actual), np.isnan(expected)
    assert (nan_a == nan_e).all(), f'{what}: NaN pattern differs at {np.argwhere(nan_a != nan_e)[:3].tolist()}'
    ok = ~nan_e
    diff = actual[ok] != expected[ok]
    n_diff = int(diff.sum())
    if n_diff:
        a, e = actual[ok][diff], expected[ok][diff]
        with np.errstate(all='ignore'):
            ulps = np.abs(a.view(np.int32).astype(np.int64) - e.view(np.int32).astype(np.int64))
        same_inf = np.isinf(a) & np.isinf(e) & (a == e)
        worst = int(ulps[~same_inf].max()) if (~same_inf).any() else 0
        frac = n_diff / max(1, int(ok.sum()))
        assert worst <= max_ulp and frac <= max(max_frac, 2.0 / max(1, int(ok.sum()))), (
            f'{what}: {n_diff} of {int(ok.sum())} valid elements differ (frac {frac:.2e}), worst {worst} ulp; '
            f'e.g. {a[:3]} vs {e[:3]}'
        )
    return n_diff


def _fit_via_abi(ctx, case_or_cfg, src, ref, norm_in=None, want_corr=True):
    c = case_or_cfg
    thresh = c['r2_inpaint_thresh'] if c['model'] == 'gain-offset' else None
    desc = _hk.make_desc(c['model'], c['kernel_shape'], c['find_r2'], thresh, c['src_nodata'], c['ref_nodata'])
    count = 3 if (c['find_r2'] or (c['model'] == 'gain-offset' and thresh is not None)) else 2
    return ctx.fit_apply(desc, src, ref, count, want_params=True, want_corr=want_corr, norm_in=norm_in)


def test_selftest(ctx):
    ctx.selftest()


@pytest.mark.oracle
@pytest.mark.parametrize('case', GOLDEN_CASES, ids=case_id)
def test_goldens_through_c_abi(ctx, goldens, case):
    """ Reference-generated golden vectors (oracle/gen_golden.py), fused fit+apply through the C ABI. """
    src = goldens[f"in_{case['variant']}_src"]
    ref = goldens[f"in_{case['variant']}_ref"]
    exp_params, exp_corr = goldens[f"{case['name']}_params"], goldens[f"{case['name']}_corr"]
    norm_in = goldens[f"{case['name']}_norm"] if case['model'] == 'gain-blk-offset' else None
    src_before, ref_before = src.copy(), ref.copy()
    params, corr, norm, n_fail = _fit_via_abi(ctx, case, src, ref, norm_in=norm_in)
    np.testing.assert_array_equal(src, src_before)  # inputs are never modified
    np.testing.assert_array_equal(ref, ref_before)
    assert n_fail == 0
    assert_close_ulp(params, exp_params, 'params')
    assert_close_ulp(corr, exp_corr, 'corrected')
    if norm_in is not None:
        np.testing.assert_array_equal(norm, norm_in)


@pytest.mark.oracle
@pytest.mark.parametrize('case', [c for c in GOLDEN_CASES if c['model'] == 'gain-blk-offset'], ids=case_id)
def test_block_norm_vs_goldens(ctx, goldens, case):
    src = goldens[f"in_{case['variant']}_src"]
    ref = goldens[f"in_{case['variant']}_ref"]
    exp = goldens[f"{case['name']}_norm"]
    desc = _hk.make_desc(case['model'], case['kernel_shape'], False, None, case['src_nodata'], case['ref_nodata'])
    norm = ctx.block_norm(desc, src, ref)
    if (exp == 0).all():
        assert (norm == 0).all()
    else:
        assert norm[0] == pytest.approx(exp[0], rel=2e-6)
        assert norm[1] == pytest.approx(exp[1], rel=2e-6, abs=2e-6 * abs(exp[0]))


@pytest.mark.parametrize('model, kernel_shape, find_r2, thresh', [
    ('gain', (5, 5), False, None),
    ('gain', (5, 5), True, None),
    ('gain', (1, 1), True, None),
    ('gain', (3, 3), False, None),
    ('gain-offset', (5, 5), False, None),
    ('gain-offset', (5, 5), True, 0.25),
    ('gain-offset', (3, 3), True, None),
    ('gain-offset', (5, 7), True, 0.25),
    ('gain-offset', (7, 5), False, 0.25),
    ('gain-offset', (7, 7), False, 0.25),   # window counts up to 49 from the lane-resident 1/N table; last strip: 3 live lanes
    ('gain-offset', (9, 7), True, 0.25),    # 63 = the table's last lane
    ('gain-offset', (9, 9), True, None),
    ('gain-offset', (15, 15), True, 0.25),
    ('gain-offset', (11, 13), True, None),
    ('gain-blk-offset', (5, 5), False, None),
    ('gain-blk-offset', (15, 15), True, None),
    ('gain-blk-offset', (1, 1), False, None),
])
@pytest.mark.oracle
@pytest.mark.parametrize('shape, variant', [((300, 1003), 'frame+holes'), ((517, 640), 'none')])
def test_multi_strip_multi_segment_vs_oracle(ctx, model, kernel_shape, find_r2, thresh, shape, variant):
    """ Rasters spanning several 248-column strips and 128-row segments, ragged widths, every kernel path
    (compile-time widths 0/1/2/3/7 and the run-time one) against the numpy oracle. """
    import warnings
    h, w = shape
    src, ref = onp.synth_pair(h, w, seed=h + w, nodata_variant=variant)
    nodata = np.nan if variant != 'none' else None
    cfg = dict(model=model, kernel_shape=kernel_shape, find_r2=find_r2, r2_inpaint_thresh=thresh, src_nodata=nodata,
               ref_nodata=nodata)
    norm_in = None
    if model == 'gain-blk-offset':
        norm_in = onp.fit_block_norm(src, nodata, ref, nodata)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        exp_params, aux = onp.fit(model, src, nodata, ref, nodata, kernel_shape, find_r2, thresh, norm_model=norm_in)
    exp_corr = onp.apply(src, exp_params)
    params, corr, norm, n_fail = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
    assert_close_ulp(params, exp_params, 'params')
    assert_close_ulp(corr, exp_corr, 'corrected')
    if model == 'gain-offset' and thresh is not None:
        assert n_fail == aux


@pytest.mark.parametrize('model, find_r2, thresh', [
    ('gain', False, None), ('gain', True, None), ('gain-blk-offset', False, None), ('gain-blk-offset', True, None),
    ('gain-offset', False, None), ('gain-offset', True, None), ('gain-offset', False, 0.25), ('gain-offset', True, 0.25),
])
@pytest.mark.oracle
@pytest.mark.parametrize('kernel_shape', [(17, 17), (21, 9), (31, 31), (9, 21), (5, 19), (33, 35), (3, 25), (45, 13), (7, 41)])
@pytest.mark.parametrize('variant', ['frame+holes', 'none'])
def test_kernels_wider_than_15_vs_oracle(ctx, oc, model, find_r2, thresh, kernel_shape, variant):
    """ Kernels wider than 15 (the reference's integration suite runs 31 x 31 -- tests/integration.py:36-37,42-43 -- and
    utils.validate_kernel_shape admits any odd shape, utils.py:104-133) take the builds that know kw // 2 mod 4 at compile time and
    kw // 8 at run time (hk_fit_kernel.h hsum_wide): every residue, one to four whole neighbour lanes per side, short / tall
    (centre ring; everything re-loaded beyond 39 rows), all three models, with and without R2 and the r2 mask, NaN holes and
    nodata None, on a raster of several strips (208 - 232 output columns each) and row segments, against the C oracle. """
    h, w = 290, 1003
    src, ref = onp.synth_pair(h, w, seed=kernel_shape[0] * 64 + kernel_shape[1], nodata_variant=variant)
    nodata = np.nan if variant != 'none' else None
    cfg = dict(model=model, kernel_shape=kernel_shape, find_r2=find_r2, r2_inpaint_thresh=thresh, src_nodata=nodata, ref_nodata=nodata)
    norm_in = onp.fit_block_norm(src, nodata, ref, nodata) if model == 'gain-blk-offset' else None
    exp_params, exp_corr, exp_fail = oc.fit_apply(model, src, nodata, ref, nodata, kernel_shape, find_r2, thresh, norm_model=norm_in)
    params, corr, norm, n_fail = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
    assert_close_ulp(params, exp_params, 'params')
    assert_close_ulp(corr, exp_corr, 'corrected')
    if model == 'gain-offset' and thresh is not None:
        assert n_fail == exp_fail
    if not find_r2:
        # the fused path of RasterFuse (corrected block only): with the r2 mask that is the certificate-only build of these widths
        desc = _hk.make_desc(model, kernel_shape, False, thresh, nodata, nodata)
        for _ in range(2):   # (the first call of a context may start with the complete build)
            _, corr_f, _, n_fail_f = ctx.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True, norm_in=norm_in)
            assert_close_ulp(corr_f, exp_corr, 'corrected (fused)')
            assert n_fail_f == (exp_fail if thresh is not None else 0)


@pytest.mark.parametrize('model, find_r2, thresh', [('gain', False, None), ('gain-blk-offset', True, None), ('gain-offset', False, 0.25),
                                                    ('gain-offset', True, None)])
@pytest.mark.oracle
@pytest.mark.parametrize('kernel_shape', [(63, 5), (129, 3), (35, 7), (255, 1), (33, 9), (61, 15), (5, 151), (9, 193), (3, 101), (1, 63), (41, 57)])
def test_tall_kernels_vs_oracle(ctx, oc, model, find_r2, thresh, kernel_shape):
    """ The extremes of the shape space: very wide kernels (up to the 193 columns a strip's overlap lanes allow: 24 whole neighbour lanes
    per side, window counts beyond the 1/N table), a single row, and kernels taller than the centre ring's default limit (39 rows): up to
    7 wide they keep the centre ring whatever their height (1 KB
    of LDS per wave and row of the half-height: 128 KB at 255 rows, one wave per CU), from 9 wide both rows are re-loaded (ring mode 0:
    the builds of hsum_wide, also for the 9 - 15 wide kernels that have compile-time builds otherwise).  utils.validate_kernel_shape
    admits any odd shape (utils.py:104-133). """
    h, w = 300, 700
    src, ref = onp.synth_pair(h, w, seed=kernel_shape[0] + kernel_shape[1], nodata_variant='frame+holes')
    cfg = dict(model=model, kernel_shape=kernel_shape, find_r2=find_r2, r2_inpaint_thresh=thresh, src_nodata=np.nan, ref_nodata=np.nan)
    norm_in = onp.fit_block_norm(src, np.nan, ref, np.nan) if model == 'gain-blk-offset' else None
    exp_params, exp_corr, exp_fail = oc.fit_apply(model, src, np.nan, ref, np.nan, kernel_shape, find_r2, thresh, norm_model=norm_in)
    params, corr, norm, n_fail = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
    assert_close_ulp(params, exp_params, 'params')
    assert_close_ulp(corr, exp_corr, 'corrected')
    if thresh is not None:
        assert n_fail == exp_fail


@pytest.mark.oracle
@pytest.mark.parametrize('seed', range(240))
def test_randomized_configurations_vs_oracle(ctx, oc, seed, monkeypatch):
    """ A seeded sweep over the configuration space (model, odd kernel shape up to 17 x 15 -- seeds from 120: 17 to 63 rows by 17 to
    55 columns, the builds of kernels wider than 15 --, R2 output, threshold, the three nodata kinds on either raster, raster shape
    from one pixel to a few strips / segments, fused vs parameter output): every draw must reproduce the C oracle (= the reference's
    whole fit branch incl. in-painting).  Seeds from 180 (round 6): rasters of up to 1500 rows -- several row segments per strip,
    rings of more than 64 / 128 rows in flight over a segment boundary -- under the two-size segment policy of large rasters, forced
    onto them (HK_WAVE_SLOTS / HK_SEG_BIG / HK_SEG_TAIL), any kernel shape of the first two groups.  The bar is the suite's: bit-exact
    but for <= 1e-5 of the pixels by <= 2 ulp (rounds 2 - 5 allowed this test 2e-3; nothing in it needs that). """
    import warnings
    # (HK_TEST_SEED_BASE: soak runs over other draws of the same space -- profiles/r06b_soak.txt; the suite's own draws are 1000 + seed)
    rng = np.random.default_rng(int(os.environ.get('HK_TEST_SEED_BASE', '1000')) + seed)
    model = ['gain', 'gain-blk-offset', 'gain-offset'][rng.integers(3)]
    if seed < 120 or (seed >= 180 and seed % 2):
        kshape = (int(rng.choice([1, 3, 5, 7, 9, 15, 17])), int(rng.choice([1, 3, 5, 7, 9, 13, 15])))
    else:
        kshape = (int(rng.choice([17, 19, 21, 23, 27, 31, 33, 45, 63])), int(rng.choice([17, 19, 21, 23, 25, 27, 29, 31, 35, 41, 55])))
    if model == 'gain-offset' and kshape[0] * kshape[1] < 2:
        kshape = (3, 3)
    find_r2 = bool(rng.integers(2))
    thresh = [None, 0.25, 0.6][rng.integers(3)] if model == 'gain-offset' else None
    h, w = int(rng.integers(1, 420)), int(rng.integers(1, 700))
    if seed >= 180:
        h, w = int(rng.integers(400, 1500)), int(rng.integers(200, 900))
        monkeypatch.setenv('HK_WAVE_SLOTS', '4')
        monkeypatch.setenv('HK_SEG_BIG', str(int(rng.choice([96, 160, 200, 300]))))
        monkeypatch.setenv('HK_SEG_TAIL', str(int(rng.choice([8, 16, 40]))))
    src = rng.uniform(0.05, 1, (h, w)).astype(np.float32)
    ref = ((0.6 + rng.random()) * src + 0.1 * rng.random() + rng.normal(0, 0.02 + 0.2 * rng.random(), (h, w))).astype(np.float32)
    nodata = {}
    for name, arr in (('src', src), ('ref', ref)):
        kind = rng.integers(3)
        holes = rng.random((h, w)) < [0.0, 0.002, 0.05][rng.integers(3)]
        if kind == 0:
            nodata[name] = None
        elif kind == 1:
            nodata[name] = np.nan
            arr[holes] = np.nan
        else:
            nodata[name] = -1.0
            arr[holes] = -1.0
    norm_in = oc.fit_block_norm(src, nodata['src'], ref, nodata['ref']) if model == 'gain-blk-offset' else None
    exp_params, exp_corr, exp_fail = oc.fit_apply(model, src, nodata['src'], ref, nodata['ref'], kshape, find_r2, thresh,
                                                  norm_in)
    cfg = dict(model=model, kernel_shape=kshape, find_r2=find_r2, r2_inpaint_thresh=thresh, src_nodata=nodata['src'],
               ref_nodata=nodata['ref'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        params, corr, _, n_fail = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
        desc = _hk.make_desc(model, kshape, find_r2, thresh, nodata['src'], nodata['ref'])
        _, corr_fused, _, n_fail_fused = ctx.fit_apply(desc, src, ref, exp_params.shape[0], want_params=False, want_corr=True,
                                                       norm_in=norm_in)
    what = f'{model} {kshape} r2={find_r2} thresh={thresh} {h}x{w} nodata={nodata}'
    if exp_params.shape[0] == 3 and kshape[0] * kshape[1] < 9:
        # R2 of a window of < 9 pixels is rounding noise wherever sstot = N*sum(r^2) - sum(r)^2 cancels (values like -1.8
        # or 9.0 come out of the reference itself): the last bit of the float64 sums decides it
        got_r2, exp_r2 = params[2], exp_params[2]
        assert (np.isnan(got_r2) == np.isnan(exp_r2)).all(), what
        ok = np.isfinite(exp_r2) & np.isfinite(got_r2)
        assert (np.isinf(got_r2) == np.isinf(exp_r2)).all(), what
        # (... and wherever the gain's own denominator N*sum(s^2) - sum(s)^2 cancels: two valid pixels with sources 0.10686 and 0.10691
        # make a gain of rounding noise and an R2 of 0.45 here against 0.97 in the oracle -- a soak over 960 other draws met three
        # such windows, profiles/r06b_soak.txt; a handful of pixels may differ, a defect moves thousands)
        n_bad = int(np.count_nonzero(~np.isclose(got_r2[ok], exp_r2[ok], rtol=1e-3, atol=1e-3)))
        assert n_bad <= max(3, int(1e-5 * ok.sum())), f'{what}: R2 of {n_bad} pixels'
        params, exp_params = params[:2], exp_params[:2]
    assert_close_ulp(params, exp_params, 'params: ' + what)
    assert_close_ulp(corr, exp_corr, 'corrected: ' + what)
    assert_close_ulp(corr_fused, exp_corr, 'corrected (fused, no parameter output): ' + what)
    if thresh is not None:
        assert n_fail == exp_fail == n_fail_fused, what


@pytest.mark.oracle
@pytest.mark.parametrize('shape', [(1, 1), (1, 7), (5, 1), (2, 3), (4, 250), (131, 5), (129, 249)])
@pytest.mark.parametrize('model, kernel_shape', [('gain', (3, 3)), ('gain-offset', (5, 5)), ('gain-offset', (15, 15))])
def test_tiny_and_ragged_shapes(ctx, shape, model, kernel_shape):
    """ Rasters smaller than the kernel / one lane / one strip: windows are partial everywhere (zero border). """
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    src = rng.uniform(0.1, 1, shape).astype(np.float32)
    ref = (1.3 * src + 0.1 + rng.normal(0, 0.01, shape)).astype(np.float32)
    cfg = dict(model=model, kernel_shape=kernel_shape, find_r2=True, r2_inpaint_thresh=None, src_nodata=np.nan,
               ref_nodata=np.nan)
    exp_params, _ = onp.fit(model, src, np.nan, ref, np.nan, kernel_shape, True, None)
    params, corr, _, _ = _fit_via_abi(ctx, cfg, src, ref)
    assert_close_ulp(params, exp_params, 'params', max_frac=1.0)  # tiny rasters: fraction is meaningless, ulps matter
    assert_close_ulp(corr, onp.apply(src, exp_params), 'corrected', max_frac=1.0)


def test_all_masked_block(ctx):
    src = np.full((40, 300), np.nan, np.float32)
    ref = np.ones((40, 300), np.float32)
    for model in ('gain', 'gain-blk-offset', 'gain-offset'):
        cfg = dict(model=model, kernel_shape=(5, 5), find_r2=True, r2_inpaint_thresh=0.25, src_nodata=np.nan,
                   ref_nodata=np.nan)
        params, corr, norm, n_fail = _fit_via_abi(ctx, cfg, src, ref)
        assert np.isnan(params).all() and np.isnan(corr).all() and n_fail == 0
        if model == 'gain-blk-offset':
            assert (norm == 0).all()  # kernel_model.py:225-226


@pytest.mark.oracle
def test_r2_fail_count_outlier(ctx):
    """ reference tests/test_kernel_model.py:166-203: one -100 outlier makes R2 < 0.5 in its k x k neighbourhood. """
    a = np.array(range(1, 201), dtype='float32').reshape(20, 10)
    a[:, [0, -1]] = np.nan
    a[[0, -1], :] = np.nan
    src = np.kron(a, np.ones((2, 2), np.float32)).astype(np.float32)
    ref = src.copy()
    loc = (src.shape[0] // 2, src.shape[1] // 2)
    ref[loc] = -100
    for k in ((5, 5), (5, 7), (9, 9)):
        cfg = dict(model='gain-offset', kernel_shape=k, find_r2=True, r2_inpaint_thresh=0.5, src_nodata=np.nan,
                   ref_nodata=np.nan)
        exp_params, exp_fail = onp.fit_gain_offset(src, np.nan, ref, np.nan, k, True, 0.5)
        params, _, _, n_fail = _fit_via_abi(ctx, cfg, src, ref, want_corr=False)
        assert n_fail == exp_fail and n_fail >= k[0] * k[1]
        assert_close_ulp(params, exp_params, 'params', max_frac=1.0)
        ul = (loc[0] - k[0] // 2, loc[1] - k[1] // 2)
        low = np.zeros(src.shape, bool)
        low[ul[0]:ul[0] + k[0], ul[1]:ul[1] + k[1]] = True
        mask = ~np.isnan(src)
        assert (params[2][low] < 0.5).all()
        assert params[2][~low & mask] == pytest.approx(1, abs=1e-3)


# -- through the reference-shaped Python classes -----------------------------------------------------------------------
@pytest.mark.parametrize('model, kernel_shape', [
    (Model.gain, (1, 1)), (Model.gain, (3, 3)), (Model.gain_blk_offset, (1, 1)), (Model.gain_blk_offset, (5, 5)),
    (Model.gain_offset, (5, 5)),
])
@pytest.mark.parametrize('cls', [RefSpaceModel, SrcSpaceModel, KernelModel])
def test_basic_fit_known_answer(ctx, cls, model, kernel_shape):
    """ reference tests/test_kernel_model.py:32-81 on a shared grid: src == ref  =>  gain ~ 1, offset ~ 0. """
    a = np.array(range(1, 201), dtype='float32').reshape(20, 10)
    a[:, [0, -1]] = np.nan
    a[[0, -1], :] = np.nan
    src_ra, ref_ra = _ra(a.copy(), np.nan), _ra(a.copy(), np.nan)
    km = cls(model, kernel_shape, mask_partial=False, r2_inpaint_thresh=0.25)
    param_ra = km.fit(src_ra, ref_ra)
    assert param_ra.shape == ref_ra.shape and param_ra.transform == ref_ra.transform
    assert (ref_ra.mask == param_ra.mask).all()
    assert param_ra.array[0, param_ra.mask] == pytest.approx(1, abs=1e-2)
    assert param_ra.array[1, param_ra.mask] == pytest.approx(0, abs=1e-2)
    np.testing.assert_array_equal(src_ra.array, a)  # not modified
    # apply with gain = offset = 1 (tests/test_kernel_model.py:84-117)
    ones = _ra(np.ones((2, *a.shape), np.float32), np.nan)
    ones.mask = src_ra.mask
    out_ra = km.apply(src_ra, ones)
    assert (src_ra.mask == out_ra.mask).all()
    assert out_ra.array[out_ra.mask] == pytest.approx(src_ra.array[out_ra.mask] + 1, abs=1e-2)


@pytest.mark.parametrize('model', list(Model))
@pytest.mark.parametrize('find_r2', [True, False])
def test_find_r2_band(ctx, model, find_r2):
    """ reference tests/test_kernel_model.py:120-140 """
    src, ref = onp.synth_pair(64, 80, 3, 'frame+holes')
    km = RefSpaceModel(model, (5, 5), find_r2=find_r2, r2_inpaint_thresh=None)
    param_ra = km.fit(_ra(src, np.nan), _ra(ref, np.nan))
    assert param_ra.count == (3 if find_r2 else 2)
    if find_r2:
        assert np.nanmax(param_ra.array[2]) <= 1


@pytest.mark.oracle
def test_reference_param_tif_on_gpu(ctx):
    """ The reference's own PARAM GeoTIFF (real homonim+OpenCV+GDAL output), reproduced by the HIP path. """
    import os
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, 'ref_param_tif.npz'))['params']
    a = np.array(range(1, 201), dtype='float32').reshape(20, 10)
    a[:, [0, -1]] = np.nan
    a[[0, -1], :] = np.nan
    km = RefSpaceModel(Model.gain_offset, (5, 5), find_r2=True, r2_inpaint_thresh=0.25)
    param_ra = km.fit(_ra(a.copy(), np.nan), _ra(a.copy(), np.nan))
    for pi in range(3):
        assert_same_f32(param_ra.array[pi], g[pi * 3], f'param band {pi}')


def test_fit_apply_fused_equals_two_calls(ctx):
    src, ref = onp.synth_pair(200, 520, 11, 'frame+holes')
    for model in Model:
        km = KernelModel(model, (5, 5), find_r2=True, r2_inpaint_thresh=None)
        src_ra, ref_ra = _ra(src.copy(), np.nan), _ra(ref.copy(), np.nan)
        corr_ra, param_ra = km.fit_apply(src_ra, ref_ra, want_params=True)
        param2 = km.fit(src_ra, ref_ra)
        corr2 = km.apply(src_ra, param2)
        assert_same_f32(param_ra.array, param2.array, 'params')
        assert_same_f32(corr_ra.array, corr2.array, 'corrected')
        corr3, none = km.fit_apply(src_ra, ref_ra)
        assert none is None
        assert_same_f32(corr3.array, corr2.array, 'corrected (no params)')


@pytest.mark.oracle
def test_concurrent_callers_share_one_model(ctx):
    """ homonim/fuse.py:396-401: many threads call fit/apply on ONE model object. """
    from concurrent.futures import ThreadPoolExecutor
    km = KernelModel(Model.gain_offset, (5, 5), r2_inpaint_thresh=None)
    blocks = [onp.synth_pair(128 + 8 * i, 300 + 16 * i, 20 + i, 'frame+holes') for i in range(8)]

    def work(b):
        corr_ra, _ = km.fit_apply(_ra(b[0], np.nan), _ra(b[1], np.nan))
        return corr_ra.array

    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(work, blocks))
    for (src, ref), corr in zip(blocks, got):
        exp, _ = onp.fit_gain_offset(src, np.nan, ref, np.nan, (5, 5), False, None)
        assert_close_ulp(corr, onp.apply(src, exp), 'corrected')


# -- RasterFuse.process block loop (homonim/fuse.py:321-408) ------------------------------------------------------------
def _oracle_process(src, ref, nodata, model, kernel_shape, max_block_mem, want_params, thresh):
    """ The reference's block loop restated with the oracle per block: read in-block, fit, apply, crop to out-block. """
    import warnings
    from homonim_amd import fuse, utils
    nb, h, w = src.shape
    find_r2 = want_params
    with_r2 = find_r2 or (model == 'gain-offset' and thresh is not None)
    corr = np.full(src.shape, np.nan, np.float32)
    params = np.full(((3 if with_r2 else 2) * nb, h, w), np.nan, np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        bps = list(fuse.block_pairs((h, w), nb, utils.overlap_for_kernel(kernel_shape), max_block_mem))
    for bp in bps:
        rs, cs = bp.src_in_block.toslices()
        s, r = src[bp.band_i][rs, cs], ref[bp.band_i][rs, cs]
        p, _ = onp.fit(model, s, nodata, r, nodata, kernel_shape, find_r2, thresh)
        c = onp.apply(s, p)
        crop = fuse.RasterFuse._crop(bp)
        ors, ocs = bp.src_out_block.toslices()
        corr[bp.band_i][ors, ocs] = c[crop]
        for pi in range(p.shape[0]):
            params[pi * nb + bp.band_i][ors, ocs] = p[pi][crop]
    return corr, params, len(bps)


@pytest.mark.parametrize('model, kernel_shape, thresh', [
    ('gain-blk-offset', (5, 5), 0.25), ('gain-offset', (5, 5), 0.25), ('gain', (3, 3), None),
    ('gain-blk-offset', (15, 15), None),
])
@pytest.mark.oracle
@pytest.mark.parametrize('threads', [1, 4])
def test_raster_fuse_process_multi_block(ctx, model, kernel_shape, thresh, threads):
    import warnings
    from homonim_amd.fuse import RasterFuse
    nb, h, w = 3, 700, 900
    pairs = [onp.synth_pair(h, w, 40 + b, 'frame+holes') for b in range(nb)]
    src = np.stack([p[0] for p in pairs])
    ref = np.stack([p[1] for p in pairs])
    mem = 0.5  # MB -> several blocks per band
    exp_corr, exp_params, n_blocks = _oracle_process(src, ref, np.nan, model, kernel_shape, mem, True, thresh)
    assert n_blocks >= 3 * 4
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with RasterFuse(src, ref) as rf:
            corr, params = rf.process(None, model, kernel_shape, param_filename=True,
                                      model_config=dict(r2_inpaint_thresh=thresh),
                                      block_config=dict(threads=threads, max_block_mem=mem))
    if model == 'gain-blk-offset':
        # block statistics: float64 on the GPU vs numpy float32 pairwise -> ~5e-7 relative in every parameter
        ok = ~np.isnan(exp_corr)
        assert (np.isnan(corr) == np.isnan(exp_corr)).all()
        assert np.max(np.abs(corr[ok] - exp_corr[ok]) / np.maximum(np.abs(exp_corr[ok]), 1e-6)) < 1e-5
        okp = ~np.isnan(exp_params[:2 * nb])
        assert np.max(np.abs(params[:2 * nb][okp] - exp_params[:2 * nb][okp]) /
                      np.maximum(np.abs(exp_params[:2 * nb][okp]), 1e-3)) < 1e-5
    else:
        assert_close_ulp(corr, exp_corr, 'corrected')
        assert_close_ulp(params, exp_params, 'params')
        # gain / gain-offset are partition invariant: the block loop equals one whole-image fit
        for b in range(nb):
            whole, _ = onp.fit(model, src[b], np.nan, ref[b], np.nan, kernel_shape, True, thresh)
            assert_close_ulp(corr[b], onp.apply(src[b], whole), 'corrected vs whole-image fit')


def test_raster_fuse_shards_are_disjoint_and_complete(ctx):
    """ rank sharding of process(): the union of two ranks' outputs is the single-rank result. """
    from homonim_amd.fuse import RasterFuse
    src, ref = onp.synth_pair(600, 500, 77, 'frame+holes')
    kw = dict(model='gain-offset', kernel_shape=(5, 5), model_config=dict(r2_inpaint_thresh=None),
              block_config=dict(threads=2, max_block_mem=0.25))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        full, _ = RasterFuse(src, ref).process(**kw)
        parts = [RasterFuse(src, ref).process(device_config=dict(rank=r, world_size=2), **kw)[0] for r in range(2)]
    filled = [~np.isnan(p) for p in parts]
    assert not (filled[0] & filled[1]).any()
    merged = np.where(filled[0], parts[0], parts[1])
    assert_same_f32(merged, full, 'merged shards')


def test_raster_fuse_uint8_output(ctx):
    from homonim_amd.fuse import RasterFuse
    rng = np.random.default_rng(5)
    src = np.round(rng.uniform(1, 255, (2, 300, 400))).astype(np.float32)
    ref = np.round(0.8 * src + 20 + rng.normal(0, 3, src.shape)).astype(np.float32)
    src[:, :4], src[:, :, -4:] = 0, 0
    corr, _ = RasterFuse(src, ref, src_nodata=0, ref_nodata=None).process(
        model='gain-blk-offset', kernel_shape=(5, 5), out_profile=dict(dtype='uint8', nodata=0))
    assert corr.dtype == np.uint8 and (corr[:, :4] == 0).all() and (corr[:, :, -4:] == 0).all()
    valid = src != 0
    assert abs(float(corr[valid].astype(np.float64).mean()) - float(ref[valid].mean())) < 3


# -- specialisations: dense (nodata None) kernels and the division-free certified r2-mask test --------------------------
@pytest.mark.parametrize('model, kernel_shape, find_r2, thresh', [
    ('gain', (5, 5), True, None), ('gain-offset', (5, 5), True, 0.25), ('gain-offset', (3, 7), False, None),
    ('gain-offset', (15, 15), True, 0.25), ('gain-offset', (9, 9), True, 0.25),
])
@pytest.mark.oracle
@pytest.mark.parametrize('shape', [(260, 1003), (131, 250), (64, 1024)])
def test_dense_path_equals_general_path(ctx, model, kernel_shape, find_r2, thresh, shape, monkeypatch):
    """ nodata None on both rasters selects the DENSE kernels (geometric window count, no mask ring); results must be
    bit-identical to the general kernels and to the oracle -- ragged widths included. """
    import warnings
    src, ref = onp.synth_pair(*shape, seed=shape[1], nodata_variant='none')
    cfg = dict(model=model, kernel_shape=kernel_shape, find_r2=find_r2, r2_inpaint_thresh=thresh, src_nodata=None,
               ref_nodata=None)
    monkeypatch.setenv('HK_FORCE_GENERAL', '1')
    p_gen, c_gen, _, f_gen = _fit_via_abi(ctx, cfg, src, ref)
    monkeypatch.setenv('HK_FORCE_GENERAL', '0')
    p_den, c_den, _, f_den = _fit_via_abi(ctx, cfg, src, ref)
    assert_same_f32(p_den, p_gen, 'params dense vs general')
    assert_same_f32(c_den, c_gen, 'corrected dense vs general')
    assert f_den == f_gen
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        exp, _ = onp.fit(model, src, None, ref, None, kernel_shape, find_r2, thresh)
    assert_close_ulp(p_den, exp, 'params')


def _fused_no_params(ctx, src, ref, nodata, k, thresh):
    desc = _hk.make_desc('gain-offset', k, False, thresh, nodata, nodata)
    _, corr, _, n_fail = ctx.fit_apply(desc, src, ref, 3, want_params=False, want_corr=True)
    return corr, n_fail


@pytest.mark.oracle
@pytest.mark.parametrize('thresh', [0.25, 0.5, 0.0, -1.0, float('-inf'), 0.9, 0.999, 0.9999999, 1.0, 2.0])
@pytest.mark.parametrize('nodata, variant', [(np.nan, 'frame+holes'), (None, 'none')])
def test_certified_r2_test_counts_exactly(ctx, oc, thresh, nodata, variant):
    """ Fused mode without parameter output: the r2-mask decision goes through the division-free certified test with
    an exact fallback; the failure count must equal the oracle's for every threshold, incl. ones in the thick of the
    R2 distribution and degenerate ones (>= 1: everything fails; -inf: everything passes). """
    src, ref = onp.synth_pair(300, 760, seed=21, nodata_variant=variant)
    rng = np.random.default_rng(3)
    ref = (ref + rng.normal(0, 0.08, ref.shape).astype(np.float32)).astype(np.float32)  # R2 spread ~0.85..0.99
    ref[100:104, 200:204] = -5.0                                                         # a low-R2 / odd-gain patch
    # expected: the whole reference branch incl. in-painting of the failing pixels (C oracle == numpy oracle, bit for bit)
    exp_params, exp_corr, exp_fail = oc.fit_apply('gain-offset', src, nodata, ref, nodata, (5, 5), False, thresh)
    corr, n_fail = _fused_no_params(ctx, src, ref, nodata, (5, 5), thresh)
    assert n_fail == exp_fail
    if thresh in (1.0, 2.0):
        valid = ~np.isnan(src) & ~np.isnan(ref)
        assert n_fail == int(valid.sum())
    assert_close_ulp(corr, exp_corr, 'corrected', max_frac=1e-4)


@pytest.mark.oracle
def test_certified_r2_test_degenerate_windows(ctx, oc):
    """ flat reference (sstot == 0), single-valid-pixel windows, negative gains: never certified, always exact. """
    rng = np.random.default_rng(8)
    src = rng.uniform(0.1, 1, (96, 300)).astype(np.float32)
    ref = np.full_like(src, 0.5)                       # sstot = 0 -> r2 = nan / -inf: all fail
    ref[:, 150:] = (-0.7 * src[:, 150:] + 1).astype(np.float32)   # perfect fit, negative gain: fail on gain > 0
    src[40:60, 20:40] = np.nan
    src[50, 30] = 0.3                                  # an isolated valid pixel: N = 1 windows
    for thresh in (0.25, -10.0):
        _, exp_corr, exp_fail = oc.fit_apply('gain-offset', src, np.nan, ref, np.nan, (5, 5), False, thresh)
        corr, n_fail = _fused_no_params(ctx, src, ref, np.nan, (5, 5), thresh)
        assert n_fail == exp_fail and n_fail > 0
        assert_close_ulp(corr, exp_corr, 'corrected', max_frac=1e-3)


def _adversarial_pair(kind, shape, seed):
    """ Rasters whose windows sit where the r2-mask certificate is tight: tiny variance on a large mean (flat DN
    imagery), R2 spread around the threshold, integer data, and magnitudes outside the certificate's windows. """
    rng = np.random.default_rng(seed)
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 'flat-dn':
        sd = 10 ** (-2 + 3.5 * xx / w)                               # std 0.01 .. 30 on a mean of 5000
        src = 5000 + sd * rng.normal(size=shape)
        ref = 0.8 * src + 300 + sd * 10 ** (-1.5 + 2 * yy / h) * rng.normal(size=shape)
    elif kind == 'marginal':
        src = rng.normal(100, 10, shape)
        ref = src + (3 + 40 * xx / w) * rng.normal(size=shape)       # R2 from ~0.9 down to ~0.05 across the columns
    elif kind == 'integer':
        src = rng.integers(0, 255, shape).astype(float)
        ref = np.round(src * (0.5 + yy / h)) + rng.integers(0, 6, shape)
    elif kind == 'tiny':
        src = 1e-17 * rng.uniform(0.05, 1, shape)
        ref = 1.2 * src + 1e-18 + 1e-19 * rng.normal(size=shape)
    elif kind == 'huge':
        src = 1e14 * rng.uniform(0.05, 1, shape)
        ref = 1.2 * src + 1e13 + 1e12 * rng.normal(size=shape)
    elif kind == 'small-gain':
        src = 1e4 * rng.uniform(0.05, 1, shape)
        ref = 10 ** (-8 + 7 * xx / w) * src + 1e-3 * rng.normal(size=shape)   # gains 1e-8 .. 0.1 (window edge 2^-20)
    else:
        raise ValueError(kind)
    return src.astype(np.float32), ref.astype(np.float32)


@pytest.mark.oracle
@pytest.mark.parametrize('kind', ['flat-dn', 'marginal', 'integer', 'tiny', 'huge', 'small-gain'])
@pytest.mark.parametrize('kernel_shape, nodata', [((5, 5), None), ((5, 5), np.nan), ((3, 7), None), ((15, 15), np.nan)])
def test_r2_certificate_on_adversarial_rasters(ctx, oc, kind, kernel_shape, nodata):
    """ The float32 certificate of the r2 mask may never disagree with the reference's own
    arithmetic: failure counts and corrected values equal the oracle's on data built to sit on its error bound. """
    src, ref = _adversarial_pair(kind, (150, 700), seed=len(kind) + kernel_shape[1])
    if nodata is not None:
        src[60:64, 100:130] = np.nan
        ref[10, ::37] = np.nan
    n_valid = int((~np.isnan(src) & ~np.isnan(ref)).sum())
    for thresh in (0.25, 0.0, 0.9):
        # (1) the certificate against the kernel's own exact evaluation (parameter output switches the certificate
        #     off): identical counts and corrected values, bit for bit
        corr, n_fail = _fused_no_params(ctx, src, ref, nodata, kernel_shape, thresh)
        desc = _hk.make_desc('gain-offset', kernel_shape, False, thresh, nodata, nodata)
        _, corr_exact, _, n_fail_exact = ctx.fit_apply(desc, src, ref, 3, want_params=True, want_corr=True)
        assert n_fail == n_fail_exact, (kind, thresh)
        assert_same_f32(corr, corr_exact, 'corrected, certificate vs exact evaluation')
        # (2) against the oracle.  On flat DN data the float64 sums of squares of a window are no longer exact and
        #     sstot = N*sum(r^2) - sum(r)^2 cancels ~8 digits, so a few decisions near the threshold depend on the
        #     summation order (oracle: OpenCV's order; kernel: column sums first) -- bounded, not bit-exact.
        _, exp_corr, exp_fail = oc.fit_apply('gain-offset', src, nodata, ref, nodata, kernel_shape, False, thresh)
        if kind in ('flat-dn', 'huge', 'tiny'):
            assert abs(n_fail - exp_fail) <= 1e-3 * n_valid, (kind, thresh, n_fail, exp_fail)
        else:
            assert n_fail == exp_fail, (kind, thresh)
            assert_close_ulp(corr, exp_corr, 'corrected', max_frac=1e-3)


@pytest.mark.oracle
def test_block_norm_heavy_ties_take_the_fallback_select(ctx):
    """ Few distinct values: the [lo, hi] pivot window holds a third of the block, overflows the compaction buffer and
    routes the band through the full-raster radix select; the order statistics must still be exact. """
    rng = np.random.default_rng(12)
    src = rng.integers(1, 4, (400, 600)).astype(np.float32)          # values 1, 2, 3
    ref = (2 * src + rng.integers(0, 2, src.shape)).astype(np.float32)
    src[:5] = 0
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, 0., None)
    norm = ctx.block_norm(desc, src, ref)
    exp = onp.fit_block_norm(src, 0., ref, None)
    assert norm[0] == pytest.approx(exp[0], rel=2e-6)
    assert norm[1] == pytest.approx(exp[1], rel=2e-6, abs=1e-6)
    # large block, clean data: the one-pass path; both must agree with numpy
    src2, ref2 = onp.synth_pair(1500, 2000, 5, 'frame+holes')
    desc2 = _hk.make_desc('gain-blk-offset', (5, 5), False, None, np.nan, np.nan)
    n2 = ctx.block_norm(desc2, src2, ref2)
    e2 = onp.fit_block_norm(src2, np.nan, ref2, np.nan)
    assert n2[0] == pytest.approx(e2[0], rel=2e-6) and n2[1] == pytest.approx(e2[1], rel=2e-6, abs=1e-6)


@pytest.mark.oracle
@pytest.mark.parametrize('seed', range(40))
def test_block_norm_randomized_vs_numpy(ctx, seed):
    """ Block statistics on random shapes (one pixel to a few hundred thousand, widths that are no multiple of 4), the three
    nodata kinds, continuous / integer-valued (ties: both select paths) / nearly constant data: std ratio and interpolated
    1st percentile against the numpy restatement of _fit_block_norm (kernel_model.py:216-229). """
    rng = np.random.default_rng(7000 + seed)
    h, w = int(rng.integers(1, 500)), int(rng.integers(1, 900))
    kind = rng.integers(4)
    if kind == 0:
        src = rng.uniform(0.05, 1, (h, w))
    elif kind == 1:
        src = rng.integers(0, int(rng.choice([4, 40, 255])), (h, w)).astype(float)     # ties
    elif kind == 2:
        src = 100 + 1e-3 * rng.normal(size=(h, w))                                      # large mean, tiny spread
    else:
        src = np.exp(rng.normal(0, 3, (h, w)))                                          # heavy tail
    ref = (0.5 + rng.random()) * src + rng.random() + 0.05 * src.std() * rng.normal(size=(h, w))
    src, ref = src.astype(np.float32), ref.astype(np.float32)
    nodata = {}
    for name, arr in (('src', src), ('ref', ref)):
        k = rng.integers(3)
        holes = rng.random((h, w)) < [0.0, 0.01, 0.3][rng.integers(3)]
        nodata[name] = [None, np.nan, -7.0][k]
        if k:
            arr[holes] = nodata[name]
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, nodata['src'], nodata['ref'])
    norm = ctx.block_norm(desc, src, ref)
    exp = onp.fit_block_norm(src, nodata['src'], ref, nodata['ref'])
    what = f'{h}x{w} kind={kind} nodata={nodata}'
    if not np.isfinite(exp).all() or exp[0] == 0:
        # degenerate statistics (no valid pixel -> zeros; zero variance -> inf / nan): same class of result
        assert (np.isfinite(norm) == np.isfinite(exp)).all(), what
        if np.isfinite(exp).all():
            assert norm == pytest.approx(exp, rel=2e-6, abs=1e-12), what
        return
    tol = 2e-6 if kind != 2 else 2e-3   # float32 pairwise std of "100 + 1e-3 noise" in numpy itself carries ~1e-4
    assert norm[0] == pytest.approx(exp[0], rel=tol), what
    # norm[1] = pct(ref) - pct(src) * norm[0] cancels: the tolerance is relative to its two terms
    valid = np.ones((h, w), bool)
    for arr, nd in ((src, nodata['src']), (ref, nodata['ref'])):
        valid &= ~np.isnan(arr) if (nd is not None and np.isnan(nd)) else (np.ones((h, w), bool) if nd is None else arr != nd)
    terms = abs(float(np.percentile(ref[valid], 1))) + abs(float(np.percentile(src[valid], 1)) * exp[0])
    assert norm[1] == pytest.approx(exp[1], rel=tol, abs=tol * terms + 1e-12), what


def test_pinned_arrays_and_caller_outputs(ctx):
    """ Pinned (page-locked) inputs/outputs through the host-pointer path give the same bytes as pageable ones. """
    src, ref = onp.synth_pair(333, 517, 9, 'frame+holes')
    ps, pr, po = ctx.pinned_empty(src.shape), ctx.pinned_empty(ref.shape), ctx.pinned_empty(src.shape)
    ps[:], pr[:] = src, ref
    desc = _hk.make_desc('gain-offset', (5, 5), False, 0.25, np.nan, np.nan)
    _, c_page, _, _ = ctx.fit_apply(desc, src, ref, 3, want_params=False, want_corr=True)
    _, c_pin, _, _ = ctx.fit_apply(desc, ps, pr, 3, want_params=False, want_corr=True, out_corr=po)
    assert c_pin is po
    assert_same_f32(np.array(c_pin), c_page, 'pinned vs pageable')
    arr = np.ascontiguousarray(src)
    ctx.pin(arr)
    _, c_reg, _, _ = ctx.fit_apply(desc, arr, ref, 3, want_params=False, want_corr=True)
    ctx.unpin(arr)
    assert_same_f32(c_reg, c_page, 'registered vs pageable')


@pytest.mark.parametrize('model, kernel_shape, find_r2, thresh, nodata', [
    ('gain-offset', (5, 5), True, 0.25, np.nan), ('gain-offset', (15, 15), True, 0.25, np.nan),
    ('gain-offset', (9, 3), False, None, None), ('gain-blk-offset', (15, 15), True, None, np.nan),
    ('gain', (1, 5), True, None, np.nan), ('gain-offset', (31, 31), False, None, np.nan),
])
@pytest.mark.oracle
def test_lds_ring_and_reload_modes_agree(ctx, model, kernel_shape, find_r2, thresh, nodata, monkeypatch):
    """ The leaving / centre rows come from a full LDS ring (mode 1, short kernels), from a centre-only LDS ring plus a
    re-loaded leaving row (mode 2, tall kernels) or are both re-loaded (mode 0); every mode must give the same bytes,
    whichever the default for the shape is. """
    import warnings
    src, ref = onp.synth_pair(300, 520, seed=kernel_shape[0], nodata_variant='frame+holes' if nodata is not None else 'none')
    cfg = dict(model=model, kernel_shape=kernel_shape, find_r2=find_r2, r2_inpaint_thresh=thresh, src_nodata=nodata,
               ref_nodata=nodata)
    norm_in = onp.fit_block_norm(src, nodata, ref, nodata) if model == 'gain-blk-offset' else None
    out = {}
    for mode in ('1', '2', '0'):
        monkeypatch.setenv('HK_USE_RING', mode)
        out[mode] = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
    monkeypatch.delenv('HK_USE_RING')
    for mode in ('2', '0'):
        assert_same_f32(out['1'][0], out[mode][0], f'params ring mode 1 vs {mode}')
        assert_same_f32(out['1'][1], out[mode][1], f'corrected ring mode 1 vs {mode}')
        assert out['1'][3] == out[mode][3]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        exp, _ = onp.fit(model, src, nodata, ref, nodata, kernel_shape, find_r2, thresh, norm_model=norm_in)
    assert_close_ulp(out['0'][0], exp, 'params')


@pytest.mark.parametrize('model, kernel_shape, thresh, nodata', [
    ('gain-offset', (5, 5), 0.25, None), ('gain-offset', (5, 5), 0.25, np.nan), ('gain-offset', (9, 9), 0.25, None),
    ('gain', (3, 3), None, np.nan), ('gain-blk-offset', (5, 7), None, np.nan),
])
@pytest.mark.oracle
def test_two_segment_sizes_equal_one_size(ctx, model, kernel_shape, thresh, nodata, monkeypatch):
    """ Large rasters are cut into long row segments followed by short ones (hk_api.hip fill_grid).  Forcing that
    policy onto a small raster (HK_WAVE_SLOTS: pretend the device holds few waves) must give the bytes of the one-size
    partition wherever the float64 window sums are exact, and the oracle's values. """
    import warnings
    src, ref = onp.synth_pair(700, 530, seed=5, nodata_variant='frame+holes' if nodata is not None else 'none')
    src, ref = np.round(src * 4096) / 4096, np.round(ref * 4096) / 4096   # 12-bit values: every window sum is exact
    src, ref = src.astype(np.float32), ref.astype(np.float32)
    cfg = dict(model=model, kernel_shape=kernel_shape, find_r2=False, r2_inpaint_thresh=thresh, src_nodata=nodata,
               ref_nodata=nodata)
    norm_in = onp.fit_block_norm(src, nodata, ref, nodata) if model == 'gain-blk-offset' else None
    one = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
    for big, tail in (('96', '16'), ('200', '8')):
        monkeypatch.setenv('HK_WAVE_SLOTS', '4')
        monkeypatch.setenv('HK_SEG_BIG', big)
        monkeypatch.setenv('HK_SEG_TAIL', tail)
        two = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
        for k in ('HK_WAVE_SLOTS', 'HK_SEG_BIG', 'HK_SEG_TAIL'):
            monkeypatch.delenv(k)
        assert_same_f32(one[0], two[0], f'params, segments {big}/{tail}')
        assert_same_f32(one[1], two[1], f'corrected, segments {big}/{tail}')
        assert one[3] == two[3]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        exp, _ = onp.fit(model, src, nodata, ref, nodata, kernel_shape, False, thresh, norm_model=norm_in)
    assert_close_ulp(one[0], exp, 'params')


# -- typed rasters either side of the path (raster_array.py:178-188 read, :353-387 write) -------------------------------
@pytest.mark.oracle
@pytest.mark.parametrize('dtype', ['uint8', 'uint16', 'int16', 'int32', 'uint32', 'float64'])
def test_integer_inputs_equal_float32_inputs(ctx, dtype):
    """ Integer / float64 rasters are converted to float32 on the device exactly as rasterio does on read. """
    rng = np.random.default_rng(31)
    hi = {'uint8': 255, 'uint16': 4000, 'int16': 3000, 'int32': 100000, 'uint32': 100000, 'float64': 1000}[dtype]
    src = rng.integers(1, hi, (300, 517)).astype(dtype)
    ref = (0.8 * src.astype(np.float64) + 20 + rng.normal(0, 3, src.shape))
    ref = np.round(ref).clip(1, None).astype(dtype) if dtype != 'float64' else ref
    if dtype != 'float64':
        src[:3], src[:, -2:] = 0, 0   # nodata 0 frame
    src_nd = 0 if dtype != 'float64' else None
    for model, k in (('gain-blk-offset', (5, 5)), ('gain-offset', (5, 5))):
        desc = _hk.make_desc(model, k, True, None, src_nd, None)
        p_t, c_t, n_t, _ = ctx.fit_apply(desc, src, ref, 3, True, True)
        p_f, c_f, n_f, _ = ctx.fit_apply(desc, src.astype(np.float32), ref.astype(np.float32), 3, True, True)
        assert_same_f32(p_t, p_f, f'{model} params typed vs float32')
        assert_same_f32(c_t, c_f, f'{model} corrected typed vs float32')
        np.testing.assert_array_equal(n_t, n_f)
        # ... and the oracle on the values rasterio would have handed the reference (raster_array.py:178-188: astype(float32))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            exp, _ = onp.fit(model, src.astype(np.float32), src_nd, ref.astype(np.float32), None, k, True, None,
                             norm_model=n_t if model == 'gain-blk-offset' else None)
        assert_close_ulp(p_t, exp, f'{model} params of {dtype} rasters vs oracle')
    # strided (windowed) integer views are taken as they are
    big = np.zeros((400, 700), dtype)
    big[50:350, 100:617] = src
    desc = _hk.make_desc('gain-offset', (5, 5), False, None, src_nd, None)
    _, c_v, _, _ = ctx.fit_apply(desc, big[50:350, 100:617], ref, 2, False, True)
    _, c_c, _, _ = ctx.fit_apply(desc, src, ref, 2, False, True)
    assert_same_f32(c_v, c_c, 'strided view vs contiguous')


@pytest.mark.oracle
def test_output_dtype_conversion_matches_reference(ctx):
    """ The corrected block converted on the device == the reference's own RasterArray._convert_array_dtype
    (tests/golden/convert_dtype.npz).  gain 1x1 with ref == src reproduces the input exactly (gain 1, offset 0). """
    import os
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, 'convert_dtype.npz'))
    a = g['input']
    desc = _hk.make_desc('gain', (1, 1), False, None, np.nan, np.nan)
    _, ident, _, _ = ctx.fit_apply(desc, a, a.copy(), 2, False, True)
    assert_same_f32(ident, a, 'identity correction')
    for key in g.files:
        if key == 'input':
            continue
        dtype, nd = key.rsplit('_', 1)
        nodata = float('nan') if nd == 'nan' else float(nd)
        _, out, _, _ = ctx.fit_apply(desc, a, a.copy(), 2, False, True, out_dtype=dtype, out_nodata=nodata)
        exp = g[key]
        assert out.dtype == exp.dtype, key
        np.testing.assert_array_equal(out, exp, err_msg=key)


def test_raster_fuse_byte_in_byte_out(ctx):
    """ uint8 rasters in, uint8 corrected raster out -- all conversions on the device; equals the float32 route. """
    from homonim_amd.fuse import RasterFuse, convert_dtype
    rng = np.random.default_rng(6)
    src = rng.integers(1, 255, (2, 500, 640)).astype(np.uint8)
    ref = np.clip(np.round(0.8 * src + 20 + rng.normal(0, 3, src.shape)), 0, 255).astype(np.uint8)
    src[:, :4], src[:, :, -4:] = 0, 0
    kw = dict(model='gain-blk-offset', kernel_shape=(5, 5), block_config=dict(threads=2, max_block_mem=0.5))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        c8, _ = RasterFuse(src, ref, src_nodata=0, ref_nodata=None).process(out_profile=dict(dtype='uint8', nodata=0), **kw)
        cf, _ = RasterFuse(src.astype(np.float32), ref.astype(np.float32), src_nodata=0, ref_nodata=None).process(**kw)
    assert c8.dtype == np.uint8
    np.testing.assert_array_equal(c8, convert_dtype(cf, 'uint8', 0))


# -- mask_partial on a shared grid (kernel_model.py:375-409) ------------------------------------------------------------
def _mask_partial_cases():
    import json, os
    from conftest import GOLDEN_DIR
    with open(os.path.join(GOLDEN_DIR, 'mask_partial.json')) as f:
        return json.load(f)


@pytest.mark.oracle
@pytest.mark.parametrize('case', _mask_partial_cases(), ids=lambda c: c['name'])
def test_mask_partial_matches_reference(ctx, case):
    """ RefSpaceModel.apply / SrcSpaceModel.fit with mask_partial=True vs outputs of the reference's own classes. """
    import os, warnings
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, 'mask_partial.npz'))
    src, ref = g['src'], g['ref']
    cls = RefSpaceModel if case['space'] == 'ref' else SrcSpaceModel
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        km = cls(case['model'], tuple(case['kernel_shape']), find_r2=True, mask_partial=True, r2_inpaint_thresh=None)
    param_ra = km.fit(_ra(src.copy(), np.nan), _ra(ref.copy(), np.nan))
    corr_ra = km.apply(_ra(src.copy(), np.nan), param_ra)
    if case['model'] == 'gain-blk-offset':  # block statistics: float64 on the GPU vs numpy float32 pairwise
        exp_p, exp_c = g[case['name'] + '_params'], g[case['name'] + '_corr']
        assert (np.isnan(param_ra.array) == np.isnan(exp_p)).all() and (np.isnan(corr_ra.array) == np.isnan(exp_c)).all()
        ok = ~np.isnan(exp_c)
        assert np.max(np.abs(corr_ra.array[ok] - exp_c[ok]) / np.maximum(np.abs(exp_c[ok]), 1e-6)) < 1e-5
    else:
        assert_close_ulp(param_ra.array, g[case['name'] + '_params'], 'params', max_frac=1.0)
        assert_close_ulp(corr_ra.array, g[case['name'] + '_corr'], 'corrected', max_frac=1.0)


@pytest.mark.oracle
@pytest.mark.parametrize('kernel_shape', [(1, 1), (3, 3), (3, 5), (5, 5)])
def test_mask_partial_erosion_properties(ctx, kernel_shape):
    """ reference tests/test_kernel_model.py:206-273: the output mask is the source mask eroded by (k + 2). """
    a = np.array(range(1, 201), dtype='float32').reshape(20, 10)
    a[:, [0, -1]] = np.nan
    a[[0, -1], :] = np.nan
    src = np.kron(a, np.ones((2, 2), np.float32)).astype(np.float32)
    km = RefSpaceModel(Model.gain_blk_offset, kernel_shape, mask_partial=True)
    ones = _ra(np.ones((2, *src.shape), np.float32), np.nan)
    ones.mask = ~np.isnan(src)
    out_ra = km.apply(_ra(src, np.nan), ones)
    src_mask = ~np.isnan(src)
    assert src_mask.sum() > out_ra.mask.sum() and src_mask[out_ra.mask].all()
    exp = onp.full_coverage_mask(src_mask, ones.array, kernel_shape)
    assert (exp == out_ra.mask).all()
    assert out_ra.array[out_ra.mask] == pytest.approx(src[out_ra.mask] + 1, abs=1e-2)
    _, _, m = ctx.partial_mask(src, np.nan, ones.array, kernel_shape, want_mask=True)
    assert (m.astype(bool) == exp).all()


# -- R2 in-painting (kernel_model.py:361-371) ----------------------------------------------------------------------------
@pytest.mark.parametrize('kernel_shape', [(5, 5), (5, 7), (9, 9)])
def test_r2_inpainting_reference_test(ctx, kernel_shape):
    """ reference tests/test_kernel_model.py:166-203: one -100 outlier; in-painting brings the offsets back to ~0 and
    lowers the gain variance; R2 is ~1 outside the outlier's k x k neighbourhood and < 0.5 inside. """
    import warnings
    a = np.array(range(1, 201), dtype='float32').reshape(20, 10)
    a[:, [0, -1]] = np.nan
    a[[0, -1], :] = np.nan
    src = np.kron(a, np.ones((2, 2), np.float32)).astype(np.float32)
    src[:, [0, 1, -2, -1]] = np.nan
    src[[0, 1, -2, -1], :] = np.nan
    ref = src.copy()
    loc = (src.shape[0] // 2, src.shape[1] // 2)
    ul = (loc[0] - kernel_shape[0] // 2, loc[1] - kernel_shape[1] // 2)
    low = np.zeros(src.shape, bool)
    low[ul[0]:ul[0] + kernel_shape[0], ul[1]:ul[1] + kernel_shape[1]] = True
    ref[loc] = -100
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        no_inpaint = RefSpaceModel(Model.gain_offset, kernel_shape, r2_inpaint_thresh=-np.inf, mask_partial=False)
        inpaint = RefSpaceModel(Model.gain_offset, kernel_shape, r2_inpaint_thresh=0.5, mask_partial=False)
    p0 = no_inpaint.fit(_ra(src.copy(), np.nan), _ra(ref.copy(), np.nan))
    p1 = inpaint.fit(_ra(src.copy(), np.nan), _ra(ref.copy(), np.nan))
    mask = ~np.isnan(src)
    for p in (p0, p1):
        assert p.array[2, ~low & mask] == pytest.approx(1, abs=1e-3)
        assert (p.array[2, low] < .5).all()
        assert (p.mask == mask).all()
    assert p0.array[1, p0.mask] != pytest.approx(0, abs=1e-1)
    assert p1.array[1, p1.mask] == pytest.approx(0, abs=1e-1)
    assert p1.array[0, p1.mask].var() < p0.array[0, p0.mask].var()


@pytest.mark.oracle
@pytest.mark.parametrize('want_params', [True, False])
def test_r2_inpainting_vs_oracle(ctx, want_params):
    """ In-painted parameters and corrected block against the oracle's restatement of the same GDAL algorithm, with
    failing regions of several shapes (isolated pixels, a patch, a stripe that crosses the nodata frame). """
    src, ref = onp.synth_pair(90, 140, 17, 'frame+holes')
    ref = ref.copy()
    ref[40:46, 60:70] = -3.0
    ref[20, 30] = 9.0
    ref[70:72, :] = np.float32(0.5)          # flat stripe: sstot ~ 0 -> fails
    exp_params, exp_fail = onp.fit_gain_offset(src, np.nan, ref, np.nan, (5, 5), False, 0.25)
    assert exp_fail > 100
    desc = _hk.make_desc('gain-offset', (5, 5), False, 0.25, np.nan, np.nan)
    for _ in range(2):  # the second call expects failures: its first pass leaves offsets + source flags for the in-painting
        params, corr, _, n_fail = ctx.fit_apply(desc, src, ref, 3, want_params=want_params, want_corr=True)
        assert n_fail == exp_fail
        if want_params:
            assert_close_ulp(params, exp_params, 'in-painted params', max_frac=1e-3)
        assert_close_ulp(corr, onp.apply(src, exp_params), 'corrected after in-painting', max_frac=1e-3)


@pytest.mark.oracle
@pytest.mark.parametrize('h', [1, 2, 3, 5, 63, 64, 65])
def test_r2_inpainting_of_blocks_of_a_few_rows(ctx, oc, h):
    """ Blocks of one to a few rows (and around the 64-row word height of the column table): the in-painting's bit words,
    distance table and search on degenerate heights; kernel 1 x 5 so that a single row still has full windows. """
    w = 300
    rng = np.random.default_rng(50 + h)
    src = rng.uniform(0.1, 1, (h, w)).astype(np.float32)
    ref = (1.3 * src + 0.05 + rng.normal(0, 0.004, (h, w))).astype(np.float32)
    ref[:, 40:52] = rng.uniform(0, 1, (h, 12)).astype(np.float32)     # uncorrelated stretch: fails the r2 mask
    ref[h // 2, 200:203] = 5.0
    src[:, :2] = np.nan
    exp_params, exp_corr, exp_fail = oc.fit_apply('gain-offset', src, np.nan, ref, np.nan, (1, 5), False, 0.25)
    assert exp_fail > 0
    desc = _hk.make_desc('gain-offset', (1, 5), False, 0.25, np.nan, np.nan)
    for _ in range(2):
        params, corr, _, n_fail = ctx.fit_apply(desc, src, ref, 3, want_params=True, want_corr=True)
        assert n_fail == exp_fail
        assert_close_ulp(params, exp_params, f'in-painted params, {h} rows', max_frac=2e-3)
        assert_close_ulp(corr, exp_corr, f'corrected, {h} rows', max_frac=2e-3)
    # only the corrected block asked for (the RasterFuse path): offsets and flags live in the stream's scratch, the closing pass drops the parameters
    for _ in range(2):
        _, corr, _, n_fail = ctx.fit_apply(desc, src, ref, 3, want_params=False, want_corr=True)
        assert n_fail == exp_fail
        assert_close_ulp(corr, exp_corr, f'corrected only, {h} rows', max_frac=2e-3)


@pytest.mark.parametrize('sd, frame, lo, hi', [(0.45, False, 0.15, 0.6), (0.45, True, 0.15, 0.6), (0.9, False, 0.6, 0.97),
                                                 (2.5, True, 0.9, 0.999), (0.3, False, 0.0005, 0.1)])
@pytest.mark.oracle
def test_r2_inpainting_of_noisy_pairs(ctx, oc, sd, frame, lo, hi):
    """ Failing pixels scattered all over a raster wide enough for the PACKED search (hk_inpaint.hip fill_fast: 16-bit keys, two
    quadrants per instruction) in the middle columns and the general one at the edges: from a few failing pixels among many sources
    to a few sources among failing pixels (searches that run past the packed range and are handed on).  Bit for bit against the
    restatement, offsets and gains of every pixel. """
    h, w = 150, 710
    rng = np.random.default_rng(int(sd * 100) + frame)
    src = rng.uniform(0.05, 1, (h, w)).astype(np.float32)
    ref = (1.3 * src + 0.05 + rng.normal(0, sd, (h, w))).astype(np.float32)
    if frame:
        src[:3], src[-3:], src[:, :3], src[:, -3:] = np.nan, np.nan, np.nan, np.nan
        holes = rng.uniform(size=(h, w)) < 0.002
        src[holes] = np.nan
        ref[60:75, 300:330] = np.nan   # a hole larger than the window: pixels that are neither sources nor worth filling
    exp_params, exp_corr, exp_fail = oc.fit_apply('gain-offset', src, np.nan, ref, np.nan, (5, 5), False, 0.25)
    valid = int(np.count_nonzero(~np.isnan(exp_params[0])))
    assert lo < exp_fail / valid < hi, exp_fail / valid
    desc = _hk.make_desc('gain-offset', (5, 5), False, 0.25, np.nan, np.nan)
    for _ in range(2):
        params, corr, _, n_fail = ctx.fit_apply(desc, src, ref, 3, want_params=True, want_corr=True)
        assert n_fail == exp_fail
        for got, exp, what in ((params[1], exp_params[1], 'in-painted offsets'), (params[0], exp_params[0], 'gains'), (corr, exp_corr, 'corrected')):
            bad = np.argwhere(~((got == exp) | (np.isnan(got) & np.isnan(exp))))
            assert len(bad) == 0, f'{what}, noise {sd}: {len(bad)} pixels differ, first at {bad[:5].tolist()}'
    # Only the corrected block asked for (the RasterFuse path): the first call of the pair runs the fit again for the in-painting's
    # inputs, the second expects failures and leaves them behind its first pass; the same bytes either way.
    for _ in range(2):
        _, corr, _, n_fail = ctx.fit_apply(desc, src, ref, 3, want_params=False, want_corr=True)
        assert n_fail == exp_fail
        bad = np.argwhere(~((corr == exp_corr) | (np.isnan(corr) & np.isnan(exp_corr))))
        assert len(bad) == 0, f'corrected only, noise {sd}: {len(bad)} pixels differ, first at {bad[:5].tolist()}'


@pytest.mark.oracle
@pytest.mark.parametrize('h, w', [(7, 63), (8, 64), (9, 65), (70, 255), (64, 256), (65, 257), (23, 511), (130, 301), (16, 1030)])
def test_r2_inpainting_at_tile_and_word_boundaries(ctx, oc, h, w):
    """ Heights around the 8-row tiles of the packed search and the 64-row words of its bit planes, widths around the 64-column
    waves, the 256-column workgroups and the 4-pixel quads of the staged table (a quad that straddles the last column), with
    failing pixels everywhere -- the last column and the last row included, whose targets the packed search hands on. """
    rng = np.random.default_rng(1000 * h + w)
    src = rng.uniform(0.05, 1, (h, w)).astype(np.float32)
    ref = (1.3 * src + 0.05 + rng.normal(0, 0.6, (h, w))).astype(np.float32)
    exp_params, exp_corr, exp_fail = oc.fit_apply('gain-offset', src, None, ref, None, (3, 5), False, 0.25)
    assert 0.1 * h * w < exp_fail < 0.98 * h * w
    desc = _hk.make_desc('gain-offset', (3, 5), False, 0.25, None, None)
    for _ in range(2):
        params, corr, _, n_fail = ctx.fit_apply(desc, src, ref, 3, want_params=True, want_corr=True)
        assert n_fail == exp_fail
        for got, exp, what in ((params[1], exp_params[1], 'in-painted offsets'), (params[0], exp_params[0], 'gains'), (corr, exp_corr, 'corrected')):
            bad = np.argwhere(~((got == exp) | (np.isnan(got) & np.isnan(exp))))
            assert len(bad) == 0, f'{what}, {h} x {w}: {len(bad)} pixels differ, first at {bad[:5].tolist()}'
    for _ in range(2):   # ... and with only the corrected block asked for
        _, corr, _, n_fail = ctx.fit_apply(desc, src, ref, 3, want_params=False, want_corr=True)
        assert n_fail == exp_fail
        bad = np.argwhere(~((corr == exp_corr) | (np.isnan(corr) & np.isnan(exp_corr))))
        assert len(bad) == 0, f'corrected only, {h} x {w}: {len(bad)} pixels differ, first at {bad[:5].tolist()}'


@pytest.mark.oracle
def test_r2_inpainting_of_a_block_taller_than_a_grid_dimension(ctx, oc):
    """ 66 000 rows: the in-painting kernels stride over the rows (a launch has at most 65 535 workgroups along y), and the
    column bit words / distance table cover the whole height; failing patches near the top, the middle and the last rows. """
    h, w = 66000, 40
    src, ref = onp.synth_pair(h, w, 23, 'frame+holes')
    ref = ref.copy()
    for y0 in (10, 32990, 65530, 65990):
        ref[y0:y0 + 6, 8:20] = -3.0
    exp_params, exp_corr, exp_fail = oc.fit_apply('gain-offset', src, np.nan, ref, np.nan, (5, 5), False, 0.25)
    assert exp_fail > 200
    desc = _hk.make_desc('gain-offset', (5, 5), False, 0.25, np.nan, np.nan)
    params, corr, _, n_fail = ctx.fit_apply(desc, src, ref, 3, want_params=True, want_corr=True)
    assert n_fail == exp_fail
    assert_close_ulp(params, exp_params, 'in-painted params', max_frac=1e-3)
    assert_close_ulp(corr, exp_corr, 'corrected after in-painting', max_frac=1e-3)


@pytest.mark.parametrize('with_params, split_api, scratch', [(False, False, False), (True, False, False), (False, True, False),
                                                             (False, False, True), (True, False, True), (False, True, True)])
@pytest.mark.oracle
def test_device_resident_job_with_inpainting(ctx, oc, with_params, split_api, scratch):
    """ hk_fit_apply_dev + hk_inpaint_dev on a 3-band job resident in HBM: only the bands whose r2 mask has failures
    are in-painted; every band equals the oracle's whole reference branch.  Run twice.  `scratch`: the job carries
    hk_dev_job.scratch, so the pass that counts the failures (then the complete build) leaves offsets + source flags for the in-painting;
    without, the certificate build + list launch count and the in-painting runs the fit once more for its inputs. """
    h, w, nb = 96, 300, 3
    stride = (w + 63) // 64 * 64
    band_stride = stride * h
    srcs, refs = [], []
    for b in range(nb):
        s_, r_ = onp.synth_pair(h, w, 40 + b, 'frame+holes')
        r_ = r_.copy()
        if b == 1:
            r_[30:36, 100:112] = -3.0
            r_[60, 200] = 9.0
        srcs.append(s_), refs.append(r_)
    pad = lambda planes: np.stack([np.pad(p, ((0, 0), (0, stride - w))) for p in planes]).astype(np.float32)  # noqa: E731
    nbytes = 4 * band_stride * nb
    names = ('src', 'ref', 'corr') + (('gain', 'offset', 'r2') if with_params else ())
    d = {k: ctx.dev_alloc(nbytes) for k in names}
    d['fail'] = ctx.dev_alloc(8 * nb)
    try:
        ctx.h2d(d['src'], pad(srcs)), ctx.h2d(d['ref'], pad(refs))
        ctx.memset(d['fail'], 0, 8 * nb)
        desc = _hk.make_desc('gain-offset', (5, 5), False, 0.25, np.nan, np.nan)
        job = _hk.DevJob()
        job.src, job.ref, job.corr, job.fail_count = d['src'], d['ref'], d['corr'], d['fail']
        job.gain, job.offset, job.r2 = (d['gain'], d['offset'], d['r2']) if with_params else (None, None, None)
        job.norm = None
        job.n_bands, job.height, job.width, job.stride, job.band_stride = nb, h, w, stride, band_stride
        job.seg_rows, job.stream = 0, 0
        if scratch:
            job.scratch_bytes = ctx.job_scratch_bytes(job)
            assert job.scratch_bytes == 5 * ((nb - 1) * band_stride + stride * h)
            d['scratch'] = ctx.dev_alloc(job.scratch_bytes)
            job.scratch = d['scratch']
        for round_i in range(2):
            for k in names:
                if k not in ('src', 'ref'):
                    ctx.memset(d[k], 0, nbytes)
            ctx.fit_apply_dev(desc, job)
            ctx.stream_sync(0)
            counts = np.zeros(nb, np.uint64)
            ctx.d2h(counts, d['fail'])           # per-band counts of the first pass ...
            if split_api:                        # ... consumed and cleared by hk_fail_counts_async + hk_inpaint_dev_counts
                host_counts = ctx.pinned_empty((nb,), np.uint64)
                ready = ctx.event()
                ctx.fail_counts_async(job, host_counts, ready)
                ctx.event_sync(ready)
                assert (host_counts == counts).all()
                n_fail = ctx.inpaint_dev_counts(desc, job, host_counts.copy())
                ctx.event_destroy(ready)
            else:                                # ... or by hk_inpaint_dev in one blocking call
                n_fail = ctx.inpaint_dev(desc, job)
            ctx.stream_sync(0)
            cleared = np.ones(nb, np.uint64)
            ctx.d2h(cleared, d['fail'])
            assert not cleared.any()
            out = {k: np.empty((nb, h, stride), np.float32) for k in names if k not in ('src', 'ref')}
            for k, arr in out.items():
                ctx.d2h(arr, d[k])
            exp_total = 0
            for b in range(nb):
                exp_params, exp_corr, exp_fail = oc.fit_apply('gain-offset', srcs[b], np.nan, refs[b], np.nan, (5, 5), False, 0.25)
                exp_total += exp_fail
                assert (exp_fail > 0) == (b == 1)
                assert int(counts[b]) == exp_fail   # (a count, whichever builds ran: ABI 6)
                assert_close_ulp(out['corr'][b, :, :w], exp_corr, f'band {b} corrected', max_frac=1e-3)
                if with_params:
                    got = np.stack([out['gain'][b, :, :w], out['offset'][b, :, :w], out['r2'][b, :, :w]])
                    assert_close_ulp(got, exp_params, f'band {b} params', max_frac=1e-3)
            assert n_fail == exp_total
    finally:
        for v in d.values():
            ctx.dev_free(v)


def _dev_job_corr_only(c, src, ref, thresh, kernel_shape=(5, 5)):
    """ one single-band device-resident gain-offset job keeping only the corrected plane; returns (job, buffers) """
    h, w = src.shape
    stride = (w + 63) // 64 * 64
    pad = lambda a: np.pad(a, ((0, 0), (0, stride - w))).astype(np.float32)  # noqa: E731
    d = {k: c.dev_alloc(4 * stride * h) for k in ('src', 'ref', 'corr')}
    d['fail'] = c.dev_alloc(8)
    c.h2d(d['src'], pad(src)), c.h2d(d['ref'], pad(ref))
    c.memset(d['fail'], 0, 8)
    job = _hk.DevJob()
    job.src, job.ref, job.corr, job.fail_count = d['src'], d['ref'], d['corr'], d['fail']
    job.gain = job.offset = job.r2 = job.norm = None
    job.n_bands, job.height, job.width, job.stride, job.band_stride = 1, h, w, stride, stride * h
    job.seg_rows, job.stream = 0, 0
    return job, d


@pytest.mark.oracle
@pytest.mark.parametrize('kernel_shape, nodata', [((5, 5), None), ((5, 5), np.nan), ((3, 7), None), ((15, 15), np.nan), ((9, 5), None), ((31, 31), np.nan)])
def test_certificate_build_and_its_list_launch(oc, kernel_shape, nodata):
    """ Gain-offset jobs that keep nothing but the corrected block start with the CERTIFICATE build (hk_fit_kernel.h launch_one:
    a wave-row whose every valid pixel certainly passes the r2 mask is settled without the reference's R2 expression); the wave-rows
    it cannot settle are marked in a bit plane and done by the LIST launch that follows -- the complete build over the runs of
    marked rows.  Whatever the share of open rows -- none (clean), a patch, scattered failures, nearly all (noise) -- the two
    launches together give the bytes of the complete build over the whole grid (which a job that also keeps the parameter planes
    runs) and of the oracle, and the raw counter is the count (up to ABI 5 it could come back as HK_COUNT_RETRY).  With job scratch
    the complete build runs and leaves the in-painting's inputs there: same results again. """
    h, w = 200, 900
    clean_s, clean_r = onp.synth_pair(h, w, 5, 'none' if nodata is None else 'frame+holes')
    patch_r = clean_r.copy()
    patch_r[90:96, 300:330] = -2.0                                    # a patch the fit cannot explain
    rng = np.random.default_rng(4)
    sparse_r = clean_r.copy()
    sparse_r[rng.random((h, w)) < 0.002] = 7.0                         # failing windows all over: runs of a few rows everywhere
    noisy_r = (clean_r + rng.normal(0, 0.5, (h, w))).astype(np.float32)  # most pixels fail: nearly every wave-row is open
    thresh = 0.25
    desc = _hk.make_desc('gain-offset', kernel_shape, False, thresh, nodata, nodata)
    c = _hk.Context(0, n_streams=1)
    try:
        def run(src, ref, keep_params=False, scratch=False):
            job, d = _dev_job_corr_only(c, src, ref, thresh)
            try:
                if keep_params:   # parameter planes (R2 included): the complete build over the whole grid
                    for k in ('gain', 'offset', 'r2'):
                        d[k] = c.dev_alloc(4 * job.stride * job.height)
                    job.gain, job.offset, job.r2 = d['gain'], d['offset'], d['r2']
                if scratch:
                    job.scratch_bytes = c.job_scratch_bytes(job)
                    d['scratch'] = c.dev_alloc(job.scratch_bytes)
                    job.scratch = d['scratch']
                c.fit_apply_dev(desc, job)
                c.stream_sync(0)
                raw = np.zeros(1, np.uint64)
                c.d2h(raw, d['fail'])
                first = np.empty((job.height, job.stride), np.float32)
                c.d2h(first, d['corr'])
                n_fail = c.inpaint_dev(desc, job)
                c.stream_sync(0)
                corr = np.empty((job.height, job.stride), np.float32)
                c.d2h(corr, d['corr'])
                return int(raw[0]), n_fail, first[:, :job.width].copy(), corr[:, :job.width].copy()
            finally:
                for v in d.values():
                    c.dev_free(v)

        for name, ref in (('clean', clean_r), ('patch', patch_r), ('sparse', sparse_r), ('noisy', noisy_r)):
            _, exp_corr, exp_fail = oc.fit_apply('gain-offset', clean_s, nodata, ref, nodata, kernel_shape, False, thresh)
            raw_l, n_l, first_l, corr_l = run(clean_s, ref)                       # certificate build + list launch
            raw_c, n_c, first_c, corr_c = run(clean_s, ref, keep_params=True)     # complete build over the whole grid
            raw_s, n_s, first_s, corr_s = run(clean_s, ref, scratch=True)         # complete build, in-painting inputs in job scratch
            assert raw_l == raw_c == raw_s == exp_fail == n_l == n_c == n_s, (name, raw_l, raw_c, raw_s, exp_fail)
            assert (exp_fail == 0) == (name == 'clean')
            assert_same_f32(first_l, first_c, f'{name}: first pass, certificate + list vs complete build')
            assert_same_f32(first_s, first_c, f'{name}: first pass with job scratch vs complete build')
            assert_same_f32(corr_l, corr_c, f'{name}: after in-painting, certificate + list vs complete build')
            assert_same_f32(corr_s, corr_c, f'{name}: after in-painting from job scratch')
            assert_close_ulp(corr_l, exp_corr, f'{name}: vs oracle', max_frac=1e-3)
            # the host-pointer path (hk_fit_apply) runs the same pair of launches and the in-painting internally; the second call
            # expects failures where the first found some (its first pass leaves the in-painting's inputs)
            for _ in range(2):
                _, corr_h, _, n_h = c.fit_apply(desc, clean_s, ref, 3, want_params=False, want_corr=True)
                assert n_h == exp_fail
                assert_same_f32(corr_h, corr_c, f'{name}: hk_fit_apply')
    finally:
        c.close()


def _exact_compare_sums(src, src_nodata, ref, ref_nodata):
    """ float64 sums of the reference's float32 per-pixel terms (compare.py:243-255) """
    s, r = np.array(src, np.float32), np.array(ref, np.float32)
    m = onp.mask_of(s, src_nodata) & onp.mask_of(r, ref_nodata)
    s[~m], r[~m] = 0, 0
    f = lambda a: float(a.astype(np.float64).sum())  # noqa: E731
    return np.array([f(s), f(r), f(s * s), f(r * r), f(s * r), f((r - s) ** 2), float(m.sum())])


@pytest.mark.parametrize('shape, variant, src_nodata, ref_nodata', [
    ((300, 761), 'frame+holes', np.nan, np.nan), ((64, 64), 'none', None, None), ((1, 5), 'none', None, np.nan),
    ((257, 3), 'frame+holes', np.nan, None), ((2100, 1030), 'frame+holes', np.nan, np.nan),
])
@pytest.mark.oracle
def test_compare_sums_equal_exact_float64_sums(ctx, shape, variant, src_nodata, ref_nodata):
    """ hk_compare_sums against numpy: same float32 per-pixel terms, float64 accumulation (order-independent to 1e-13) """
    src, ref = onp.synth_pair(*shape, seed=3, nodata_variant=variant)
    got = ctx.compare_sums(src, src_nodata, ref, ref_nodata)
    exp = _exact_compare_sums(src, src_nodata, ref, ref_nodata)
    assert got[6] == exp[6]
    assert np.allclose(got, exp, rtol=1e-12, atol=0)
    # against the reference's float32 pairwise sums (oracle): equal to numpy's own summation error
    ora = onp.compare_sums(src, src_nodata, ref, ref_nodata)
    assert np.allclose(got, [float(ora[k]) for k in onp.COMPARE_KEYS], rtol=3e-6, atol=0)


@pytest.mark.oracle
def test_compare_sums_numeric_nodata_strided_rows_and_empty(ctx):
    rng = np.random.default_rng(8)
    big = np.round(rng.uniform(0, 255, (120, 400))).astype(np.float32)
    src = big[:, 3:330]                       # rows are strided, not 16-byte aligned
    ref = (0.7 * src + 12).astype(np.float32)
    got = ctx.compare_sums(src, 0., ref, None)
    assert np.allclose(got, _exact_compare_sums(src, 0., ref, None), rtol=1e-12, atol=0)
    assert got[6] == np.count_nonzero(src)
    none = ctx.compare_sums(np.full((40, 50), np.nan, np.float32), np.nan, ref[:40, :50], None)
    assert not none.any()                     # nothing valid: all seven sums are zero
    with pytest.raises(ValueError):
        ctx.compare_sums(src, None, ref[:, :-1], None)


def test_compare_sums_on_device_resident_bands(ctx):
    h, w, nb = 200, 333, 3
    stride = (w + 63) // 64 * 64
    planes = [onp.synth_pair(h, w, 60 + b, 'frame+holes') for b in range(nb)]
    pad = lambda k: np.stack([np.pad(p[k], ((0, 0), (0, stride - w)), constant_values=np.nan) for p in planes])  # noqa: E731
    d = {k: ctx.dev_alloc(4 * stride * h * nb) for k in ('src', 'ref')}
    d['sums'] = ctx.dev_alloc(8 * 7 * nb)
    try:
        ctx.h2d(d['src'], pad(0).astype(np.float32)), ctx.h2d(d['ref'], pad(1).astype(np.float32))
        job = _hk.DevJob()
        job.src, job.ref = d['src'], d['ref']
        job.corr = job.gain = job.offset = job.r2 = job.norm = job.fail_count = None
        job.n_bands, job.height, job.width, job.stride, job.band_stride = nb, h, w, stride, stride * h
        job.seg_rows, job.stream = 0, 0
        ctx.compare_sums_dev(job, np.nan, np.nan, d['sums'])
        ctx.stream_sync(0)
        got = np.zeros((nb, 7), np.float64)
        ctx.d2h(got, d['sums'])
        for b in range(nb):
            assert np.array_equal(got[b], ctx.compare_sums(planes[b][0], np.nan, planes[b][1], np.nan)), b
    finally:
        for v in d.values():
            ctx.dev_free(v)


@pytest.mark.oracle
def test_raster_compare_matches_reference_statistics():
    """ RasterCompare.process against the statistics of the reference's own RasterCompare.process (compare.npz),
    block partition and thread pool included. """
    from homonim_amd.compare import RasterCompare
    from test_compare_cpu import assert_stats_close, compare_cases, stats_rows
    for case in compare_cases():
        for threads in (1, 3):
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                with RasterCompare(case['src'], case['ref'], src_nodata=case['src_nodata'], ref_nodata=case['ref_nodata'],
                                   proc_crs=case['proc_crs']) as cmp:
                    stats = cmp.process(threads=threads, max_block_mem=case['max_block_mem'])
            assert list(stats.keys()) == case['bands']
            assert_stats_close(stats_rows(stats), case['stats'])


def test_c_program_drives_the_library_on_the_gpu(tmp_path):
    """ tests/c/abi_consumer.c (plain C99, dlopen) with a GPU present: context, one compute call, known answer. """
    import os
    import subprocess
    from conftest import REPO
    exe = tmp_path / 'abi_consumer'
    subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(REPO, 'include'),
                    os.path.join(REPO, 'tests', 'c', 'abi_consumer.c'), '-o', str(exe), '-ldl'], check=True)
    run = subprocess.run([str(exe), _hk.lib_path()], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert 'ok (GPU present)' in run.stdout


def test_raster_compare_known_answers_across_grids():
    """ The reference's RasterCompare API tests (tests/test_compare.py:30-148) on in-memory rasters: a source compared
    with its own 2:1 block average is a perfect match (r2 = 1, RMSE = 0) on the reference grid; results do not depend on
    the thread count, barely on the block size (1e-5), and the two processing grids agree to 1e-3. """
    from homonim_amd.compare import RasterCompare
    rng = np.random.default_rng(11)
    h, w, nb = 192, 256, 2
    yy, xx = np.mgrid[0:h, 0:w]
    src = np.stack([(np.sin(xx / (9.0 + b)) * np.cos(yy / (7.0 + b)) + 2 + 0.05 * rng.standard_normal((h, w))) for b in range(nb)])
    src = src.astype(np.float32)
    src[:, :4], src[:, -4:], src[:, :, :4], src[:, :, -4:] = np.nan, np.nan, np.nan, np.nan   # even-aligned NaN frame
    ref = src.reshape(nb, h // 2, 2, w // 2, 2).mean(axis=(2, 4), dtype=np.float64).astype(np.float32)
    src_tf, ref_tf = Affine(0.5, 0., 10., 0., -0.5, 200.), Affine(1., 0., 10., 0., -1., 200.)
    kw = dict(src_nodata=np.nan, ref_nodata=np.nan, transform=src_tf, ref_transform=ref_tf, band_names=['b1', 'b2'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with RasterCompare(src, ref, proc_crs='ref', **kw) as cmp:
            assert cmp.proc_crs.name == 'ref'
            one = cmp.process(threads=1)
            many = cmp.process(threads=4, max_block_mem=0.02)
        with RasterCompare(src, ref, proc_crs='src', **kw) as cmp:
            on_src = cmp.process(threads=2)
    assert list(one.keys()) == ['b1', 'b2', 'Mean']
    n_valid = int((~np.isnan(ref[0])).sum())
    for band in ('b1', 'b2', 'Mean'):
        assert one[band]['n'] == n_valid
        assert one[band]['r2'] == pytest.approx(1, abs=1e-6) and one[band]['rmse'] == pytest.approx(0, abs=1e-6)
        assert one[band]['rrmse'] == pytest.approx(0, abs=1e-6)
        for k in ('r2', 'rmse', 'rrmse'):
            assert many[band][k] == pytest.approx(one[band][k], rel=1e-5, abs=1e-6)
        assert many[band]['n'] == one[band]['n']
        assert on_src[band]['r2'] == pytest.approx(one[band]['r2'], rel=2e-2)    # up-sampled reference vs 0.5 m detail
        assert on_src[band]['n'] > 3.5 * one[band]['n']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with RasterCompare(src, ref, proc_crs='ref', **kw) as cmp:
            assert cmp.process(threads=3) == one                                 # block order, not completion order


def test_fill_nodata_known_answers():
    """ the restated GDAL fill on hand-checkable cases (oracle only; the GPU version is compared with it above). """
    img = np.zeros((5, 7), np.float32)
    msk = np.zeros((5, 7), bool)
    img[2, 1], img[2, 5] = 10, 20
    msk[2, 1] = msk[2, 5] = True
    out = onp.fill_nodata(img, msk)
    assert out[2, 3] == pytest.approx(15)                         # equidistant
    assert out[2, 2] == pytest.approx((10 / 1 + 20 / 3) / (1 / 1 + 1 / 3))
    assert out[2, 1] == 10 and out[2, 5] == 20                    # sources untouched
    far = onp.fill_nodata(np.pad(img, ((0, 0), (0, 300))), np.pad(msk, ((0, 0), (0, 300))))
    assert far[2, 5 + 150] == 0                                   # nothing within 100 px: left as it was


# -- re-sampling around the path (raster_array.py:526-578; kernel_model.py:466-535) --------------------------------------
def _ref_arrays():
    """ reference tests/conftest.py:74-140: 100 cm gradient with a 1-px NaN frame, its 2x up-sampled 50 cm version with a
    2-px frame, and their north-up transforms (origin (5, -5)). """
    a100 = np.array(range(1, 201), dtype='float32').reshape(20, 10)
    a100[:, [0, -1]] = np.nan
    a100[[0, -1], :] = np.nan
    a50 = np.kron(a100, np.ones((2, 2))).astype(np.float32)
    a50[:, [0, 1, -2, -1]] = np.nan
    a50[[0, 1, -2, -1], :] = np.nan
    t100 = Affine(1, 0, 0, 0, -1, 0) * Affine.translation(5, 5)
    t50 = t100 * Affine.scale(0.5)
    return RasterArray(a100, CRS(), t100), RasterArray(a50, CRS(), t50)


@pytest.mark.parametrize('resampling, mapping, dst_shape', [
    ('average', (2., 0., 2., 0.), (75, 120)), ('average', (2.2, 0.3, 1.9, -0.4), (70, 100)),
    ('average', (10., 3., 10., 1.), (15, 24)), ('average', (.5, 0., .5, 0.), (300, 480)),
    ('cubic_spline', (.5, 0., .5, 0.), (300, 480)), ('cubic_spline', (.45, -.5, .45, -.5), (340, 540)),
    ('bilinear', (.45, -.5, .45, -.5), (340, 540)), ('nearest', (.5, 0., .5, 0.), (300, 480)),
    ('nearest', (2.2, 0.3, 1.9, -0.4), (70, 100)), ('average', (1., 0., 1., 0.), (150, 240)),
])
@pytest.mark.oracle
@pytest.mark.parametrize('nodata', [np.nan, None, 0.])
def test_device_resamplers_equal_oracle(ctx, resampling, mapping, dst_shape, nodata):
    """ hk_reproject vs the oracle's restatement of the same GDAL kernels: bit-exact float32, NaN pattern included. """
    src, _ = onp.synth_pair(150, 240, 13, 'frame+holes' if nodata is not None else 'none')
    if nodata == 0.:
        src = np.nan_to_num(src, nan=0.)
    code = onp.RESAMPLING_CODES[resampling]
    got = ctx.reproject(src, nodata, mapping, dst_shape, code, np.nan)
    exp = onp.reproject(src, nodata, mapping, dst_shape, dst_nodata=np.nan, resampling=resampling)
    assert_same_f32(got, exp, f'{resampling} {mapping}')


@pytest.mark.parametrize('model, kernel_shape', [
    (Model.gain, (1, 1)), (Model.gain, (3, 3)), (Model.gain_blk_offset, (1, 1)), (Model.gain_blk_offset, (5, 5)),
    (Model.gain_offset, (5, 5)),
])
def test_ref_and_src_space_fit_different_grids(ctx, model, kernel_shape):
    """ reference tests/test_kernel_model.py:32-81: src = 2x up-sampled ref  =>  gain ~ 1, offset ~ 0 in both spaces. """
    ra100, ra50 = _ref_arrays()
    km = RefSpaceModel(model, kernel_shape, mask_partial=False, r2_inpaint_thresh=0.25)
    param_ra = km.fit(ra50, ra100.copy())
    assert param_ra.shape == ra100.shape and param_ra.transform == ra100.transform
    assert (ra100.mask == param_ra.mask).all()
    assert param_ra.array[0, param_ra.mask] == pytest.approx(1, abs=1e-2)
    assert param_ra.array[1, param_ra.mask] == pytest.approx(0, abs=1e-2)
    km = SrcSpaceModel(model, kernel_shape, mask_partial=False, r2_inpaint_thresh=0.25)
    param_ra = km.fit(ra100, ra50)
    assert param_ra.shape == ra100.shape and param_ra.transform == ra100.transform
    assert (ra100.mask == param_ra.mask).all()
    assert param_ra.array[0, param_ra.mask] == pytest.approx(1, abs=1e-2)
    assert param_ra.array[1, param_ra.mask] == pytest.approx(0, abs=1e-2)


def test_ref_space_apply_different_grids(ctx):
    """ reference tests/test_kernel_model.py:84-117: parameters == 1 on the 100 cm grid applied to the 50 cm source. """
    ra100, ra50 = _ref_arrays()
    km = RefSpaceModel(Model.gain_blk_offset, (5, 5), mask_partial=False)
    param_ra = ra100.copy()
    pmask = param_ra.mask
    param_ra.array = np.ones((2, *param_ra.shape), dtype='float32')
    param_ra.mask = pmask
    out_ra = km.apply(ra50, param_ra)
    assert out_ra.transform == ra50.transform and out_ra.shape == ra50.shape
    assert (ra50.mask == out_ra.mask).all()
    assert out_ra.array[out_ra.mask] == pytest.approx(ra50.array[out_ra.mask] + 1, abs=1e-2)


@pytest.mark.parametrize('kernel_shape, mask_partial', [((1, 1), False), ((1, 1), True), ((3, 3), True), ((3, 5), True),
                                                        ((5, 5), True)])
@pytest.mark.oracle
def test_ref_and_src_masking_different_grids(ctx, kernel_shape, mask_partial):
    """ reference tests/test_kernel_model.py:206-273: mask_partial across grids -- the output mask equals the source mask
    averaged to the parameter grid (>= 1), eroded by (k + 2), brought back with nearest. """
    ra100, ra50 = _ref_arrays()
    km = RefSpaceModel(Model.gain_blk_offset, kernel_shape, mask_partial=mask_partial)
    param_ra = ra100.copy()
    pmask = param_ra.mask
    param_ra.array = np.ones((2, *param_ra.shape), dtype='float32')
    param_ra.mask = pmask
    out_ra = km.apply(ra50.copy(), param_ra)
    if not mask_partial:
        assert (ra50.mask == out_ra.mask).all()
    else:
        assert ra50.mask.sum() > out_ra.mask.sum() and ra50.mask[out_ra.mask].all()
        cover = onp.reproject(ra50.mask.astype(np.float32), None, (2., 0., 2., 0.), ra100.shape, dst_nodata=None,
                              resampling='average')
        exp = onp.full_coverage_mask(cover >= 1, param_ra.array, kernel_shape)
        exp_us = onp.reproject(exp.astype(np.float32), None, (.5, 0., .5, 0.), ra50.shape, dst_nodata=0,
                               resampling='nearest').astype(bool)
        assert (exp_us == out_ra.mask).all()
    # src space: parameters on the 100 cm source grid, reference at 50 cm
    km = SrcSpaceModel(Model.gain_blk_offset, kernel_shape, mask_partial=mask_partial)
    p = km.fit(ra100, ra50)
    if not mask_partial:
        assert (ra100.mask == p.mask).all()
    else:
        assert ra100.mask.sum() > p.mask.sum() and ra100.mask[p.mask].all()
        exp = onp.full_coverage_mask(ra100.mask, np.where(ra100.mask, 1, np.nan)[None].repeat(2, 0), kernel_shape)
        assert (exp == p.mask).all()


def test_force_proc_crs(ctx):
    """ reference tests/test_kernel_model.py:276-293: the 'wrong' space still gives corrected ~ source. """
    ra100, ra50 = _ref_arrays()
    km = RefSpaceModel(Model.gain_blk_offset, (5, 5), mask_partial=False)
    p = km.fit(ra100, ra50.copy())       # low-res source, high-res reference, fitted on the reference grid
    out = km.apply(ra100, p)
    assert ra100.array[ra100.mask] == pytest.approx(out.array[out.mask], abs=2)
    km = SrcSpaceModel(Model.gain_blk_offset, (5, 5), mask_partial=False)
    p = km.fit(ra50, ra100)
    out = km.apply(ra50, p)
    assert ra50.array[ra50.mask] == pytest.approx(out.array[out.mask], abs=2)


# -- round 2: out-block windows (host-pointer and device-resident), SrcSpaceModel.fit_apply with mask_partial ------------
@pytest.mark.parametrize('model, kernel_shape, thresh, out_dtype', [
    ('gain-offset', (5, 5), 0.25, 'float32'), ('gain-blk-offset', (5, 5), None, 'float32'), ('gain', (3, 3), None, 'uint8'),
    ('gain-offset', (5, 5), 0.9999, 'float32'),     # most pixels fail the r2 mask: in-painting passes + second copy-out
])
def test_fit_apply_block_writes_the_out_block_in_place(ctx, model, kernel_shape, thresh, out_dtype):
    """ hk_fit_apply_block = hk_fit_apply_io + crop, written straight into a window of the caller's larger rasters
    (strided rows, strided parameter planes) -- pageable and page-locked -- must equal the packed result, cropped. """
    src, ref = onp.synth_pair(301, 515, 21, 'frame+holes')
    if out_dtype == 'uint8':
        src, ref = (np.where(np.isnan(a), np.nan, np.round(a * 200)).astype(np.float32) for a in (src, ref))
    desc = _hk.make_desc(model, kernel_shape, True, thresh, np.nan, np.nan)
    nodata = 0 if out_dtype == 'uint8' else None
    exp_p, exp_c, _, exp_fail = ctx.fit_apply(desc, src, ref, 3, want_params=True, want_corr=True, out_dtype=out_dtype,
                                              out_nodata=nodata)
    r0, c0, rows, cols = 3, 8, 290, 500
    for pinned in (False, True):
        alloc = (lambda shape, dt: ctx.pinned_empty(shape, dt)) if pinned else (lambda shape, dt: np.empty(shape, dt))
        big_c = alloc((2, 400, 700), out_dtype)     # "corrected raster" of 2 bands: the block lands in band 1 at (50, 100)
        big_p = alloc((6, 400, 700), np.float32)    # 3 parameters x 2 bands, band-interleaved like the reference's file
        big_c[:] = 7
        big_p[:] = 7
        corr_dst = big_c[1, 50:50 + rows, 100:100 + cols]
        params_dst = big_p[1::2][:, 50:50 + rows, 100:100 + cols]
        _, n_fail = ctx.fit_apply_block(desc, src, ref, (r0, c0, rows, cols), corr_dst, params_dst, out_nodata=nodata)
        assert n_fail == exp_fail
        got_c, exp_win = np.array(corr_dst), exp_c[r0:r0 + rows, c0:c0 + cols]
        assert (got_c == exp_win).all() if out_dtype == 'uint8' else bool(((got_c == exp_win) | (np.isnan(got_c) & np.isnan(exp_win))).all())
        assert_same_f32(np.array(params_dst), exp_p[:, r0:r0 + rows, c0:c0 + cols], 'parameter window')
        big_c[1, 50:50 + rows, 100:100 + cols] = 7      # nothing outside the window was touched
        big_p[1::2][:, 50:50 + rows, 100:100 + cols] = 7
        assert (np.array(big_c) == 7).all() and (np.array(big_p) == 7).all()
    if thresh == 0.9999:
        assert exp_fail > 1000


@pytest.mark.oracle
@pytest.mark.parametrize('model, kernel_shape', [('gain-blk-offset', (15, 15)), ('gain', (5, 5)), ('gain-offset', (7, 7))])
def test_device_job_store_window(ctx, oc, model, kernel_shape):
    """ A block of a larger device-resident raster processed in place (BASELINE.json configs[3]): job = the in-block
    including its halo, store window = its out-block; pixels outside the window keep their previous contents and the
    window equals the whole-job result there. """
    H, W = 520, 1100
    stride = 1152
    src, ref = onp.synth_pair(H, W, 33, 'frame+holes')
    pad = lambda a: np.pad(a, ((0, 0), (0, stride - W)))
    d = {k: ctx.dev_alloc(4 * stride * H) for k in ('src', 'ref', 'corr', 'full')}
    ctx.h2d(d['src'], pad(src)), ctx.h2d(d['ref'], pad(ref))
    marker = np.full((H, stride), -5.0, np.float32)
    ctx.h2d(d['corr'], marker), ctx.h2d(d['full'], marker)
    desc = _hk.make_desc(model, kernel_shape, False, None, np.nan, np.nan)
    norm = ctx.dev_alloc(16)
    # the in-block: rows 40..440, columns 96..996 of the raster (16-byte aligned origin), out-block inset by 8
    y0, x0, h, w = 40, 96, 400, 900
    job = _hk.DevJob()
    off = 4 * (y0 * stride + x0)
    job.src, job.ref = d['src'] + off, d['ref'] + off
    job.gain = job.offset = job.r2 = job.fail_count = None
    job.norm = norm if model == 'gain-blk-offset' else None
    job.n_bands, job.height, job.width, job.stride, job.band_stride, job.seg_rows, job.stream = 1, h, w, stride, 0, 0, 0
    if model == 'gain-blk-offset':
        ctx.block_norm_dev(desc, job, norm)
    job.corr = d['full'] + off
    ctx.fit_apply_dev(desc, job)                               # whole in-block
    job.corr = d['corr'] + off
    job.out_row0, job.out_col0, job.out_rows, job.out_cols = 8, 8, h - 16, w - 16
    ctx.fit_apply_dev(desc, job)                               # out-block only
    ctx.stream_sync(0)
    full, win = np.empty((H, stride), np.float32), np.empty((H, stride), np.float32)
    ctx.d2h(full, d['full']), ctx.d2h(win, d['corr'])
    inner = (slice(y0 + 8, y0 + h - 8), slice(x0 + 8, x0 + w - 8))
    assert_same_f32(win[inner], full[inner], 'store window vs whole job')
    win[inner] = -5.0
    assert (win == -5.0).all()                                 # the halo (and everything else) was not written
    # and the block equals the oracle on the in-block
    nm = None
    if model == 'gain-blk-offset':
        nm = np.zeros(2)
        ctx.d2h(nm, norm)
    params, _ = onp.fit(model, src[y0:y0 + h, x0:x0 + w], np.nan, ref[y0:y0 + h, x0:x0 + w], np.nan, kernel_shape, False, None,
                        norm_model=nm)
    assert_close_ulp(full[y0:y0 + h, x0:x0 + w], onp.apply(src[y0:y0 + h, x0:x0 + w], params), 'in-block vs oracle')
    for k in d.values():
        ctx.dev_free(k)
    ctx.dev_free(norm)
    # a window that does not start on a quad, or together with the r2 mask, is refused
    job.out_col0 = 6
    with pytest.raises(Exception):
        ctx.fit_apply_dev(desc, job)


@pytest.mark.oracle
@pytest.mark.parametrize('case', [c for c in _mask_partial_cases() if c['space'] == 'src'], ids=lambda c: c['name'])
def test_src_space_fit_apply_honours_mask_partial(ctx, case):
    """ SrcSpaceModel.fit_apply on a shared grid with mask_partial=True (what RasterFuse(proc_crs='src') calls per block)
    must apply the eroded full-coverage mask of SrcSpaceModel.fit (kernel_model.py:526-531): same outputs as the
    reference's fit -> apply goldens. """
    import os, warnings
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, 'mask_partial.npz'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        km = SrcSpaceModel(case['model'], tuple(case['kernel_shape']), find_r2=True, mask_partial=True, r2_inpaint_thresh=None)
    corr_ra, param_ra = km.fit_apply(_ra(g['src'].copy(), np.nan), _ra(g['ref'].copy(), np.nan), want_params=True)
    exp_p, exp_c = g[case['name'] + '_params'], g[case['name'] + '_corr']
    assert (np.isnan(param_ra.array) == np.isnan(exp_p)).all() and (np.isnan(corr_ra.array) == np.isnan(exp_c)).all()
    assert np.isnan(exp_c).sum() > np.isnan(g['src']).sum()          # the mask did shrink
    ok = ~np.isnan(exp_c)
    assert np.max(np.abs(corr_ra.array[ok] - exp_c[ok]) / np.maximum(np.abs(exp_c[ok]), 1e-6)) < 1e-5
    # and through the block loop
    from homonim_amd.fuse import RasterFuse
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        corr, _ = RasterFuse(g['src'], g['ref'], proc_crs='src').process(
            None, case['model'], tuple(case['kernel_shape']), model_config=dict(mask_partial=True, r2_inpaint_thresh=None))
    assert (np.isnan(corr[0]) == np.isnan(exp_c)).all()


# -- round 2: the other GDAL re-sampling methods, stretched kernels, grids of opposite orientation ---------------------
@pytest.mark.parametrize('resampling, mapping, dst_shape', [
    ('bilinear', (2.2, 0.3, 1.9, -0.4), (40, 55)), ('cubic_spline', (3.0, 0., 3.0, 0.), (27, 40)),
    ('cubic', (.45, -.5, .45, -.5), (180, 270)), ('cubic', (2.5, 0.2, 2.5, 0.1), (30, 45)), ('cubic', (1., 0., 1., 0.), (80, 120)),
    ('lanczos', (.5, 0., .5, 0.), (160, 240)), ('lanczos', (2.0, 0., 2.0, 0.), (40, 60)),
    ('max', (2.2, 0.3, 1.9, -0.4), (40, 55)), ('min', (4., 0., 4., 0.), (20, 30)), ('sum', (2., 0., 2., 0.), (40, 60)),
    ('rms', (2.2, 0.3, 1.9, -0.4), (40, 55)), ('bilinear', (1.5, 0., .5, 0.), (160, 80)),
])
@pytest.mark.oracle
@pytest.mark.parametrize('nodata', [np.nan, None])
def test_more_device_resamplers_equal_oracle(ctx, resampling, mapping, dst_shape, nodata):
    """ cubic / lanczos / max / min / sum / rms and the stretched (down-sampling) bilinear / cubic_spline kernels of
    GDAL's GWKResample vs the oracle's restatement: bit-exact float32, except lanczos (its weights go through sin(),
    whose last bit differs between the device's and the host's libm: 1e-6 relative). """
    src, _ = onp.synth_pair(80, 120, 14, 'frame+holes' if nodata is not None else 'none')
    got = ctx.reproject(src, nodata, mapping, dst_shape, onp.RESAMPLING_CODES[resampling], np.nan)
    exp = onp.reproject(src, nodata, mapping, dst_shape, dst_nodata=np.nan, resampling=resampling)
    if resampling == 'lanczos':
        assert (np.isnan(got) == np.isnan(exp)).all()
        ok = ~np.isnan(exp)
        assert np.max(np.abs(got[ok] - exp[ok]) / np.abs(exp[ok])) < 1e-6
    else:
        assert_same_f32(got, exp, f'{resampling} {mapping}')


@pytest.mark.parametrize('resampling', ['mode', 'med', 'q1', 'q3'])
@pytest.mark.parametrize('mapping, dst_shape, nodata', [((2., 0., 2., 0.), (40, 60), None), ((3., .5, 2.5, .25), (31, 39), np.nan),
                                                        ((6., 0., 6., 0.), (13, 20), np.nan), ((.5, 0., .5, 0.), (160, 240), None)])
@pytest.mark.oracle
def test_rank_order_resamplers_equal_oracle(ctx, resampling, mapping, dst_shape, nodata):
    """ GWKAverageOrMode's rank-order branches (mode, med, q1, q3) over the `average` footprint vs the oracle's restatement;
    integer-valued data so that `mode` has real ties. """
    rng = np.random.default_rng(31)
    src = rng.integers(0, 6, (80, 120)).astype(np.float32)
    if nodata is not None:
        src[rng.random(src.shape) < 0.1] = np.nan
        src[:3], src[:, :2] = np.nan, np.nan
    got = ctx.reproject(src, nodata, mapping, dst_shape, onp.RESAMPLING_CODES[resampling], np.nan)
    exp = onp.reproject(src, nodata, mapping, dst_shape, dst_nodata=np.nan, resampling=resampling)
    assert_same_f32(got, exp, f'{resampling} {mapping}')
    cont, _ = onp.synth_pair(80, 120, 15, 'frame+holes' if nodata is not None else 'none')
    got = ctx.reproject(cont, nodata, mapping, dst_shape, onp.RESAMPLING_CODES[resampling], np.nan)
    exp = onp.reproject(cont, nodata, mapping, dst_shape, dst_nodata=np.nan, resampling=resampling)
    assert_same_f32(got, exp, f'{resampling} {mapping} continuous')


def test_gauss_is_not_a_warp_method(ctx):
    src, _ = onp.synth_pair(20, 30, 1, 'none')
    with pytest.raises(Exception, match='(?i)resampling'):
        ctx.reproject(src, None, (2., 0., 2., 0.), (10, 15), 7, np.nan)


@pytest.mark.parametrize('resampling', ['bilinear', 'cubic', 'cubic_spline', 'lanczos', 'average', 'max', 'min', 'rms'])
def test_resamplers_known_answers(ctx, resampling):
    """ Properties that hold for GDAL whatever its rounding: a constant raster stays constant (up and down), `sum` of whole
    2 x 2 cells is their sum, max >= average >= min. """
    const = np.full((40, 60), 3.25, np.float32)
    code = onp.RESAMPLING_CODES[resampling]
    for mapping, shape in (((.5, 0., .5, 0.), (80, 120)), ((2., 0., 2., 0.), (20, 30)), ((2.5, .25, 3., .5), (13, 23))):
        out = ctx.reproject(const, None, mapping, shape, code, np.nan)
        assert np.allclose(out, 3.25, rtol=1e-6), (resampling, mapping)
    src, _ = onp.synth_pair(40, 60, 3, 'none')
    m = (2., 0., 2., 0.)
    s = ctx.reproject(src, None, m, (20, 30), onp.RESAMPLING_CODES['sum'], np.nan)
    np.testing.assert_allclose(s, src.reshape(20, 2, 30, 2).astype(np.float64).sum(axis=(1, 3)), rtol=1e-6)
    mx, mn, av = (ctx.reproject(src, None, m, (20, 30), onp.RESAMPLING_CODES[k], np.nan) for k in ('max', 'min', 'average'))
    assert (mx >= av).all() and (av >= mn).all()
    np.testing.assert_array_equal(mx, src.reshape(20, 2, 30, 2).max(axis=(1, 3)))


def test_south_up_and_mirrored_grids(ctx):
    """ reference fixtures tests/conftest.py:423-436,499-517 (south-up rasters): re-sampling between grids of opposite
    orientation = flipping the array and re-sampling between north-up grids; RasterFuse brings such rasters to north-up
    first (as utils.same_orientation_crs does through a WarpedVRT) and returns north-up outputs. """
    import warnings
    from homonim_amd.fuse import RasterFuse
    src, ref = onp.synth_pair(120, 90, 5, 'frame+holes')
    crs = CRS('EPSG:3857')
    north = Affine(2., 0., 100., 0., -2., 900.)                        # 2 m pixels, origin top-left
    south = Affine(2., 0., 100., 0., 2., 900. - 2. * 120)              # the same extent, rows bottom-up
    ra_n = RasterArray(src, crs, north, nodata=np.nan)
    ra_s = RasterArray(np.ascontiguousarray(src[::-1]), crs, south, nodata=np.nan)
    coarse = Affine(4., 0., 100., 0., -4., 900.)
    for rs in (Resampling.average, Resampling.cubic_spline, Resampling.nearest, Resampling.bilinear):
        a = ra_n.reproject(transform=coarse, shape=(60, 45), resampling=rs) if rs == Resampling.average else ra_n.reproject(
            transform=Affine(1., 0., 100., 0., -1., 900.), shape=(240, 180), resampling=rs)
        b = ra_s.reproject(transform=a.transform, shape=a.shape, resampling=rs)
        assert_same_f32(b.array, a.array, f'south-up source, {rs.name}')
        # and onto a south-up destination: the flipped result
        dst_s = Affine(a.transform.a, 0., a.transform.c, 0., -a.transform.e, a.transform.f + a.transform.e * a.shape[0])
        c = ra_n.reproject(transform=dst_s, shape=a.shape, resampling=rs)
        assert_same_f32(c.array, a.array[::-1], f'south-up destination, {rs.name}')
    # mirrored columns
    west = Affine(-2., 0., 100. + 2. * 90, 0., -2., 900.)
    ra_w = RasterArray(np.ascontiguousarray(src[:, ::-1]), crs, west, nodata=np.nan)
    a = ra_n.reproject(transform=coarse, shape=(60, 45), resampling=Resampling.average)
    assert_same_f32(ra_w.reproject(transform=coarse, shape=(60, 45), resampling=Resampling.average).array, a.array, 'mirrored')
    # RasterFuse: a south-up source against a north-up reference on a coarser grid
    ref_c = RasterArray(ref, crs, north, nodata=np.nan).reproject(transform=coarse, shape=(60, 45), resampling=Resampling.average)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        exp, _ = RasterFuse(ra_n, ref_c).process(None, 'gain-blk-offset', (3, 3))
        got, _ = RasterFuse(ra_s, ref_c).process(None, 'gain-blk-offset', (3, 3))
        ref_s = RasterArray(np.ascontiguousarray(ref_c.array[::-1]), crs, Affine(4., 0., 100., 0., 4., 900. - 4. * 60), nodata=np.nan)
        got2, _ = RasterFuse(ra_n, ref_s).process(None, 'gain-blk-offset', (3, 3))
    assert_same_f32(got, exp, 'south-up source through RasterFuse')
    assert_same_f32(got2, exp, 'south-up reference through RasterFuse')


@pytest.mark.parametrize('model, kernel_shape, nodata', [
    ('gain', (15, 15), np.nan), ('gain', (11, 11), None), ('gain', (13, 7), np.nan), ('gain', (7, 5), np.nan),
    ('gain-blk-offset', (9, 9), np.nan), ('gain-blk-offset', (11, 5), np.nan), ('gain-blk-offset', (15, 15), 0.0),
])
@pytest.mark.oracle
def test_split_ring_agrees_with_the_other_ring_modes(ctx, model, kernel_shape, nodata, monkeypatch):
    """ Ring mode 3 (round 3): the rh newest rows of a tall kernel's window stay in registers, the rh + 1 older ones in an LDS
    ring -- no re-load of the leaving row.  Same bytes as the centre-ring / re-load modes, on rasters with a NaN frame and
    holes, numeric nodata and none, taller than one wave segment; and equal to the oracle. """
    import warnings
    h, w = 700, 530
    src, ref = onp.synth_pair(h, w, seed=kernel_shape[0] + 100, nodata_variant='none' if nodata is None else 'frame+holes')
    if nodata == 0.0:
        src, ref = np.where(np.isnan(src), np.float32(0), src), np.where(np.isnan(ref), np.float32(0), ref)
    cfg = dict(model=model, kernel_shape=kernel_shape, find_r2=False, r2_inpaint_thresh=None, src_nodata=nodata, ref_nodata=nodata)
    norm_in = onp.fit_block_norm(src, nodata, ref, nodata) if model == 'gain-blk-offset' else None
    out = {}
    for mode in ('3', '2', '0'):
        monkeypatch.setenv('HK_USE_RING', mode)
        out[mode] = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
    monkeypatch.delenv('HK_USE_RING')
    out['default'] = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
    for mode in ('2', '0', 'default'):
        assert_same_f32(out['3'][0], out[mode][0], f'params ring mode 3 vs {mode}')
        assert_same_f32(out['3'][1], out[mode][1], f'corrected ring mode 3 vs {mode}')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        exp, _ = onp.fit(model, src, nodata, ref, nodata, kernel_shape, False, None, norm_model=norm_in)
    assert_close_ulp(out['3'][0], exp, 'params')


# ----------------------------------------------------------------------------------------------------------------------
# round 3: hygiene of the host layer
@pytest.mark.oracle
def test_a_tolerated_hip_failure_does_not_poison_the_next_launch(ctx):
    """ HIP keeps a per-thread last error until it is read and the launch wrappers report through hipGetLastError():
    registering memory that is page-locked already (the advertised `corr_out = ctx.pinned_empty(...)` use) must not make
    the NEXT kernel launch of the calling thread fail.  Also RasterFuse.process with such a corr_out, one thread, pin=True. """
    from homonim_amd.fuse import RasterFuse
    src, ref = onp.synth_pair(200, 300, 4, 'frame+holes')
    arr = np.ascontiguousarray(src)
    vp = arr.ctypes.data_as(_hk.C.c_void_p)
    assert ctx._lib.hk_host_unregister(ctx._h, vp) != _hk.HK_OK                          # a failing HIP call, tolerated
    desc = _hk.make_desc('gain', (3, 3), False, None, np.nan, np.nan)
    _, c1, _, _ = ctx.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True)     # the next launch on this thread
    big = np.zeros((64, 1024), np.float32)                                               # (whole pages: see the pin test)
    bp = big.ctypes.data_as(_hk.C.c_void_p)
    assert ctx._lib.hk_host_register(ctx._h, bp, big.nbytes) == _hk.HK_OK
    rc = ctx._lib.hk_host_register(ctx._h, bp, big.nbytes)                               # refused (or counted) by the runtime
    assert rc in (_hk.HK_ERR_ALREADY, _hk.HK_OK)
    _, c2, _, _ = ctx.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True)
    assert ctx._lib.hk_host_unregister(ctx._h, bp) == _hk.HK_OK
    if rc == _hk.HK_OK:
        ctx._lib.hk_host_unregister(ctx._h, bp)
    exp, _ = onp.fit('gain', src, np.nan, ref, np.nan, (3, 3), False, None)
    assert_same_f32(c1, onp.apply(src, exp), 'launch after a tolerated failure')
    assert_same_f32(c2, c1, 'launch after a tolerated failure')
    corr_out = ctx.pinned_empty((1,) + src.shape)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        rf = RasterFuse(src, ref, src_nodata=np.nan, ref_nodata=np.nan)
        corr, _ = rf.process(model='gain', kernel_shape=(3, 3), corr_out=corr_out, block_config=dict(threads=1),
                             device_config=dict(devices=[ctx.device], pin=True))
    assert corr is corr_out
    assert_same_f32(np.array(corr[0]), onp.apply(src, exp), 'process() into a page-locked corr_out')


def test_pin_registrations_are_counted(ctx):
    """ Two users of one array share a registration; the last unpin removes it; page-locked memory of somebody else is left
    alone (concurrent RasterFuse.process calls on one source raster). """
    arr = np.ascontiguousarray(np.random.default_rng(1).random((64, 1024), np.float32))
    key = (arr.ctypes.data, arr.nbytes)
    assert ctx.pin(arr) and ctx.pin(arr)
    assert _hk._pins[key] == [2, True]
    ctx.unpin(arr)
    assert _hk._pins[key] == [1, True]
    # (whether the runtime refuses a second hipHostRegister of a registered range or counts it differs from call to call on
    # ROCm 7.2 -- the registry above never issues one)
    ctx.unpin(arr)
    assert key not in _hk._pins
    assert ctx._lib.hk_host_register(ctx._h, arr.ctypes.data_as(_hk.C.c_void_p), arr.nbytes) == _hk.HK_OK   # it was released
    assert ctx._lib.hk_host_unregister(ctx._h, arr.ctypes.data_as(_hk.C.c_void_p)) == _hk.HK_OK
    foreign = ctx.pinned_empty((16, 16))
    fkey = (foreign.ctypes.data, foreign.nbytes)
    assert ctx.pin(foreign) and _hk._pins[fkey][0] == 1
    ctx.unpin(foreign)                      # only undoes a registration it made itself
    assert fkey not in _hk._pins
    foreign[:] = 1.0                        # still alive and page-locked
    ctx.unpin(arr)                          # unknown range: no-op


def test_fit_apply_block_with_different_row_strides_and_bad_views(ctx):
    """ The corrected and the parameter rasters need not share a row stride (hk_out_window.param_stride); views the copies
    cannot address raise ValueError (not an assert that `python -O` strips). """
    src, ref = onp.synth_pair(120, 260, 3, 'frame+holes')
    desc = _hk.make_desc('gain-offset', (3, 3), False, None, np.nan, np.nan)
    exp_p, exp_c, _, _ = ctx.fit_apply(desc, src, ref, 2, want_params=True, want_corr=True)
    r0, c0, rows, cols = 2, 4, 100, 240
    big_c = np.full((150, 300), 7, np.float32)
    big_p = np.full((2, 130, 512), 7, np.float32)           # another row stride than the corrected raster's
    ctx.fit_apply_block(desc, src, ref, (r0, c0, rows, cols), big_c[10:10 + rows, 20:20 + cols], big_p[:, 5:5 + rows, 8:8 + cols])
    assert_same_f32(big_c[10:10 + rows, 20:20 + cols], exp_c[r0:r0 + rows, c0:c0 + cols], 'corrected window')
    assert_same_f32(big_p[:, 5:5 + rows, 8:8 + cols], exp_p[:, r0:r0 + rows, c0:c0 + cols], 'parameter window')
    big_c[10:10 + rows, 20:20 + cols] = 7
    big_p[:, 5:5 + rows, 8:8 + cols] = 7
    assert (big_c == 7).all() and (big_p == 7).all()
    with pytest.raises(ValueError):
        ctx.fit_apply_block(desc, src, ref, (r0, c0, rows, cols), big_c[10:10 + rows, 20:20 + 2 * cols:2], None)   # column stride 2
    with pytest.raises(ValueError):
        ctx.fit_apply_block(desc, src, ref, (r0, c0, rows, cols), big_c[10:10 + rows + 1, 20:20 + cols], None)    # wrong shape
    with pytest.raises(ValueError):
        ctx.fit_apply_block(desc, src, ref, (r0, c0, rows, cols), None, big_p.astype(np.float64)[:, 5:5 + rows, 8:8 + cols])


def test_stream_probe_adds_its_two_streams(ctx):
    """ hk_stream_probe_dev (bench.py `roofline.copy_gbps_measured`): out = a + b over device buffers, incl. a ragged tail. """
    n = 4 * (256 * 4 * 1024 * 3 + 777)     # floats, not a multiple of the kernel's chunk
    rng = np.random.default_rng(3)
    a, b = rng.random(n, np.float32), rng.random(n, np.float32)
    d = [ctx.dev_alloc(4 * n) for _ in range(3)]
    ctx.h2d(d[0], a), ctx.h2d(d[1], b)
    ctx.memset(d[2], 0, 4 * n)
    ctx.stream_probe_dev(d[0], d[1], d[2], 4 * n, stream=0)
    ctx.stream_sync(0)
    out = np.empty(n, np.float32)
    ctx.d2h(out, d[2])
    assert (out == a + b).all()
    with pytest.raises(ValueError):
        ctx.stream_probe_dev(d[0], d[1], d[2], 4 * n + 4)
    for p in d:
        ctx.dev_free(p)


def test_host_calls_and_device_jobs_on_one_stream_are_kept_apart():
    """ include/homonim_hk.h: host-pointer calls lease the pooled streams device-resident jobs name by index.  The library keeps
    the two apart at run time (a lease drains a stream that carries device jobs, a device-job call waits while its stream is
    leased): on a ONE-stream context, a thread of host-pointer calls beside a thread of device jobs (block statistics + fit,
    i.e. users of the stream's shared scratch) gives the same bytes as each alone. """
    import threading
    c1 = _hk.Context(0, n_streams=1)
    try:
        src, ref = onp.synth_pair(260, 520, 5, 'frame+holes')
        desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, np.nan, np.nan)
        _, exp_c, exp_norm, _ = c1.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True)
        stride = 576
        pad = lambda a: np.ascontiguousarray(np.pad(a, ((0, 0), (0, stride - a.shape[1]))))   # noqa: E731
        d = {k: c1.dev_alloc(4 * stride * 260) for k in ('src', 'ref', 'corr')}
        d_norm = c1.dev_alloc(16)
        c1.h2d(d['src'], pad(src)), c1.h2d(d['ref'], pad(ref))
        job = _hk.DevJob()
        job.src, job.ref, job.corr = d['src'], d['ref'], d['corr']
        job.gain = job.offset = job.r2 = job.fail_count = None
        job.norm = d_norm
        job.n_bands, job.height, job.width, job.stride, job.band_stride, job.seg_rows, job.stream = 1, 260, 520, stride, stride * 260, 0, 0
        errors = []

        def host_loop():
            try:
                for _ in range(25):
                    _, c, n, _ = c1.fit_apply(desc, src, ref, 2, want_params=False, want_corr=True)
                    assert_same_f32(c, exp_c, 'host-pointer call beside device jobs')
                    assert (n == exp_norm).all()
            except Exception as ex:   # noqa: BLE001
                errors.append(ex)

        def dev_loop():
            try:
                out = np.empty((260, stride), np.float32)
                for _ in range(25):
                    c1.memset(d['corr'], 0, 4 * stride * 260)
                    c1.block_norm_dev(desc, job, d_norm)
                    c1.fit_apply_dev(desc, job)
                    c1.stream_sync(0)
                    c1.d2h(out, d['corr'])
                    assert_same_f32(out[:, :520], exp_c, 'device job beside host-pointer calls')
            except Exception as ex:   # noqa: BLE001
                errors.append(ex)

        ts = [threading.Thread(target=host_loop), threading.Thread(target=dev_loop)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors[0]
        for p in list(d.values()) + [d_norm]:
            c1.dev_free(p)
    finally:
        c1.close()


@pytest.mark.oracle
@pytest.mark.parametrize('find_r2', [False, True])
def test_gain_blk_offset_with_degenerate_block_statistics(ctx, find_r2):
    """ the fused gain-blk-offset build without R2 normalises the window SUM (n0 * sum(s) + n1 * N) instead of
    every pixel.  With degenerate block statistics -- a constant source block: std(src) = 0, so norm = (inf | nan, nan) -- the
    reference's normalised source is NaN everywhere, every pixel is masked and every output is NaN; the window-sum form must
    give the same (and the per-pixel form used with R2 as well). """
    import warnings
    rng = np.random.default_rng(3)
    src = np.full((96, 300), 0.5, np.float32)   # exactly representable: numpy's float32 mean and std are exact (0.5, 0)
    ref = (0.5 + rng.random((96, 300))).astype(np.float32)
    cfg = dict(model='gain-blk-offset', kernel_shape=(5, 5), find_r2=find_r2, r2_inpaint_thresh=None, src_nodata=np.nan, ref_nodata=np.nan)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        norm = onp.fit_block_norm(src, np.nan, ref, np.nan)
        assert not np.isfinite(norm).all()
        exp, _ = onp.fit('gain-blk-offset', src, np.nan, ref, np.nan, (5, 5), find_r2, None, norm_model=norm)
        exp_c = onp.apply(src, exp)
    got_p, got_c, _, _ = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm)
    assert np.isnan(exp).all() and np.isnan(exp_c).all()
    assert np.isnan(got_p).all() and np.isnan(got_c).all()
    # ... and the statistics the GPU computes itself for that block are the same non-finite pair
    desc = _hk.make_desc('gain-blk-offset', (5, 5), find_r2, None, np.nan, np.nan)
    gn = ctx.block_norm(desc, src, ref)
    assert (np.isnan(gn) == np.isnan(norm)).all() and (gn[~np.isnan(norm)] == norm[~np.isnan(norm)]).all()


@pytest.mark.oracle
@pytest.mark.parametrize('shape', [(1, 1), (3, 7), (5, 300), (16, 40), (11, 11), (40, 1), (130, 259), (257, 65)])
def test_split_ring_on_rasters_smaller_than_the_kernel_or_the_strip(ctx, shape):
    """ ring mode 3 on degenerate shapes: rasters smaller than the kernel, narrower than a lane quad, one wave segment + 1 row,
    one strip + 1 column -- gain 11x11 and gain-blk-offset 9x9 (the kernels that use it by default) against the oracle. """
    import warnings
    src, ref = onp.synth_pair(shape[0], shape[1], seed=shape[0] * 1000 + shape[1], nodata_variant='frame+holes' if min(shape) > 8 else 'none')
    nodata = np.nan
    for model, k in (('gain', (11, 11)), ('gain-blk-offset', (9, 9)), ('gain', (15, 5))):
        cfg = dict(model=model, kernel_shape=k, find_r2=False, r2_inpaint_thresh=None, src_nodata=nodata, ref_nodata=nodata)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            norm_in = onp.fit_block_norm(src, nodata, ref, nodata) if model == 'gain-blk-offset' else None
            exp, _ = onp.fit(model, src, nodata, ref, nodata, k, False, None, norm_model=norm_in)
        got = _fit_via_abi(ctx, cfg, src, ref, norm_in=norm_in)
        assert_close_ulp(got[0], exp, f'{model} {k} on {shape}')
        assert_close_ulp(got[1], onp.apply(src, exp), f'{model} {k} on {shape}: corrected')


@pytest.mark.oracle
@pytest.mark.parametrize('model, kernel_shape, thresh', [('gain-offset', (5, 5), 0.25), ('gain-offset', (15, 15), 0.25), ('gain', (5, 5), None),
                                                         ('gain-blk-offset', (15, 15), None), ('gain-offset', (31, 31), 0.25)])
def test_clustered_holes_vs_oracle(ctx, oc, model, kernel_shape, thresh):
    """ The synthetic workload's cloud / shadow-mask-like nodata (hk_synth_fill_dev variant 6: NaN frame + ~1 % of the area in round
    holes 32 - 128 pixels across, source and reference independently; bench.py --nodata 6): wave-rows that leave a hole, run along
    its edge or lie wholly inside one, against the C oracle (raster_array.py:298-308, utils.py:54-56). """
    h, w = 1100, 1600
    d_src, d_ref = ctx.dev_alloc(4 * h * w), ctx.dev_alloc(4 * h * w)
    try:
        ctx.synth_fill_dev(d_src, d_ref, 1, h, w, w, h * w, seed=4321, nodata_variant=6, stream=0)
        ctx.stream_sync(0)
        src, ref = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
        ctx.d2h(src, d_src), ctx.d2h(ref, d_ref)
    finally:
        ctx.dev_free(d_src), ctx.dev_free(d_ref)
    holes = np.isnan(src[3:-3, 3:-3]).mean(), np.isnan(ref[3:-3, 3:-3]).mean()
    assert all(0.002 < f < 0.05 for f in holes), holes
    assert not np.array_equal(np.isnan(src), np.isnan(ref))           # independent masks
    norm_in = oc.fit_block_norm(src, np.nan, ref, np.nan) if model == 'gain-blk-offset' else None
    exp_p, exp_c, exp_fail = oc.fit_apply(model, src, np.nan, ref, np.nan, kernel_shape, False, thresh, norm_model=norm_in)
    desc = _hk.make_desc(model, kernel_shape, False, thresh, np.nan, np.nan)
    for want_params in (True, False):
        params, corr, _, n_fail = ctx.fit_apply(desc, src, ref, exp_p.shape[0], want_params=want_params, want_corr=True, norm_in=norm_in)
        assert_close_ulp(corr, exp_c, 'corrected')
        if want_params:
            assert_close_ulp(params, exp_p, 'params')
        assert n_fail == (exp_fail if thresh is not None else 0)


def test_checksum_of_a_device_window_is_the_sum_of_its_bit_patterns(ctx):
    """ hk_debug_checksum_dev (bench.py `shard_checksum`: the union of N ranks' shards against the single-rank result): the sum of
    the 32-bit patterns of a strided window, modulo 2^64 -- NaN payloads and negative zeros included. """
    rng = np.random.default_rng(9)
    h, w, stride = 301, 1003, 1024
    a = rng.normal(0, 1e3, (h, stride)).astype(np.float32)
    a[5, 7], a[6, 8], a[200, 1000] = np.nan, -0.0, np.float32('inf')
    d = ctx.dev_alloc(4 * h * stride)
    try:
        ctx.h2d(d, a)
        for y0, x0, hh, ww in ((0, 0, h, w), (17, 12, 100, 333), (300, 1002, 1, 1)):
            exp = int(a[y0:y0 + hh, x0:x0 + ww].view(np.uint32).astype(np.uint64).sum(dtype=np.uint64))
            assert ctx.checksum_dev(d + 4 * (y0 * stride + x0), stride, hh, ww) == exp
    finally:
        ctx.dev_free(d)
