"""
CPU tests: the numpy oracle (oracle/oracle_np.py) against golden vectors produced by executing the reference's own
kernel_model.py (oracle/gen_golden.py).  Bar: bit-exact float32 (NaN == NaN).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_CASES, GOLDEN_DIR, assert_same_f32, case_id
from oracle import oracle_np as onp


@pytest.mark.parametrize('case', GOLDEN_CASES, ids=case_id)
def test_oracle_np_matches_reference_goldens(case, goldens):
    src = goldens[f"in_{case['variant']}_src"]
    ref = goldens[f"in_{case['variant']}_ref"]
    exp_params = goldens[f"{case['name']}_params"]
    exp_corr = goldens[f"{case['name']}_corr"]
    assert not case['inpaint_had_holes']  # goldens never depend on the (un-restated) GDAL in-painting

    params, aux = onp.fit(
        case['model'], src, case['src_nodata'], ref, case['ref_nodata'], case['kernel_shape'], case['find_r2'],
        case['r2_inpaint_thresh']
    )
    assert_same_f32(params, exp_params, 'params')
    corr = onp.apply(src, params)
    assert_same_f32(corr, exp_corr, 'corrected')
    if case['model'] == 'gain-blk-offset':
        np.testing.assert_array_equal(aux, goldens[f"{case['name']}_norm"])
    if case['model'] == 'gain-offset' and case['r2_inpaint_thresh'] is not None:
        assert aux == 0


def test_oracle_np_known_answer_conftest_arrays():
    """ reference tests/conftest.py:74-80 array vs itself, gain-offset 5x5: gain 1, offset 0, R2 1 (the known answer
    behind tests/data/parameter/*PARAM*.tif and tests/test_kernel_model.py:41-81). """
    g = np.load(os.path.join(GOLDEN_DIR, 'conftest_100cm_gain_offset_k5.npz'))
    params, n_fail = onp.fit_gain_offset(g['src'], np.nan, g['src'].copy(), np.nan, (5, 5), True, 0.25)
    assert_same_f32(params, g['params'], 'params')
    mask = ~np.isnan(g['src'])
    assert params[0][mask] == pytest.approx(1, abs=1e-2)
    assert params[1][mask] == pytest.approx(0, abs=1e-2)
    assert params[2][mask] == pytest.approx(1, abs=1e-3)
    assert np.isnan(params[:, ~mask]).all()


def test_box_sum_definition():
    """ zero-border, centre-anchored, un-normalised; float64 accumulate -> input depth. """
    x = np.arange(1, 13, dtype=np.float32).reshape(3, 4)
    s = onp.box_sum(x, (3, 3))
    assert s.dtype == np.float32
    assert s[0, 0] == 1 + 2 + 5 + 6
    assert s[1, 1] == x[0:3, 0:3].sum()
    assert s[2, 3] == 7 + 8 + 11 + 12
    s2 = onp.box_sum(x, (1, 3), square=True)
    assert s2[0, 0] == 1 + 4 and s2[0, 1] == 1 + 4 + 9
    s64 = onp.box_sum(x.astype(np.float64), (3, 1))
    assert s64.dtype == np.float64 and s64[1, 0] == 1 + 5 + 9
    # float64 accumulation: 2^24 + 1 + 1 is exact in f64, rounds once to f32
    y = np.array([[2**24, 1, 1]], dtype=np.float32)
    assert onp.box_sum(y, (1, 3))[0, 1] == np.float32(2**24 + 2)


def test_reference_param_tif():
    """
    The reference's own golden data file tests/data/parameter/float_100cm_rgb_FUSE_cREF_mGAIN-OFFSET_k5_5_PARAM.tif
    (decoded by oracle/decode_ref_param_tif.py), produced by the REAL homonim + OpenCV + GDAL stack: gain-offset 5x5,
    r2_inpaint_thresh 0.25, source == reference == reference tests/conftest.py:74-80 array.  The oracle must
    reproduce it bit-for-bit -- including the two pixels (16,4), (16,6) where gain = 1.0000079, which only happens
    when sqrBoxFilter returns float64 (this is what pins the OpenCV boundary of the restatement).
    """
    g = np.load(os.path.join(GOLDEN_DIR, 'ref_param_tif.npz'))['params']
    assert g.shape == (9, 20, 10)
    src = np.array(range(1, 201), dtype='float32').reshape(20, 10)
    src[:, [0, -1]] = np.nan
    src[[0, -1], :] = np.nan
    params, n_fail = onp.fit_gain_offset(src, np.nan, src.copy(), np.nan, (5, 5), True, 0.25)
    assert n_fail == 0
    for band_i in range(3):  # the file holds 3 identical source bands
        for pi, pname in enumerate(('gain', 'offset', 'r2')):
            assert_same_f32(params[pi], g[pi * 3 + band_i], f'{pname} (file band {pi * 3 + band_i + 1})')
    assert params[0, 16, 4] == np.float32(1.0000079) and params[0, 16, 4] != 1  # the deviation is real


def _mask_partial_cases():
    import json
    with open(os.path.join(GOLDEN_DIR, 'mask_partial.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('case', _mask_partial_cases(), ids=lambda c: c['name'])
def test_oracle_full_coverage_mask_matches_reference(case):
    """ mask_partial on a shared grid: oracle restatement of _full_coverage_mask vs the reference's own
    RefSpaceModel.apply / SrcSpaceModel.fit outputs (tests/golden/mask_partial.npz). """
    g = np.load(os.path.join(GOLDEN_DIR, 'mask_partial.npz'))
    src, ref = g['src'], g['ref']
    k = tuple(case['kernel_shape'])
    norm = onp.fit_block_norm(src, np.nan, ref, np.nan) if case['model'] == 'gain-blk-offset' else None
    params, _ = onp.fit(case['model'], src, np.nan, ref, np.nan, k, True, None, norm_model=norm)
    if case['space'] == 'ref':
        # RefSpaceModel: fit unmasked, apply with parameters masked by the eroded source-mask & param-mask
        assert_same_f32(params, g[case['name'] + '_params'], 'params')
        cover = onp.full_coverage_mask(~np.isnan(src), params, k)
        masked = np.where(cover, params[:2], np.float32(np.nan))
        assert_same_f32(onp.apply(src, masked), g[case['name'] + '_corr'], 'corrected')
    else:
        # SrcSpaceModel: all parameter bands masked by the eroded reference-mask & param-mask at fit time
        cover = onp.full_coverage_mask(~np.isnan(ref), params, k)
        masked = np.where(cover, params, np.float32(np.nan))
        assert_same_f32(masked, g[case['name'] + '_params'], 'params')
        assert_same_f32(onp.apply(src, masked), g[case['name'] + '_corr'], 'corrected')


# -- restated GDAL warp kernels (parity with GDAL unpinned): invariants every implementation of them must satisfy ---------
def test_reproject_oracle_invariants():
    a = np.array(range(1, 201), dtype='float32').reshape(20, 10)
    a[:, [0, -1]] = np.nan
    a[[0, -1], :] = np.nan
    k = np.kron(a, np.ones((2, 2))).astype(np.float32)
    a50 = k.copy()
    a50[:, [0, 1, -2, -1]] = np.nan
    a50[[0, 1, -2, -1], :] = np.nan
    # reference conftest arrays (tests/conftest.py:74-89): average 50 cm -> 100 cm gives back the 100 cm array exactly
    down = onp.reproject(a50, np.nan, (2., 0., 2., 0.), (20, 10), resampling='average')
    assert_same_f32(down, a, 'average 2:1')
    # nearest 1:2 replicates pixels; the identity mapping is the identity for every kernel
    assert_same_f32(onp.reproject(a, np.nan, (.5, 0., .5, 0.), (40, 20), resampling='nearest'), k, 'nearest 1:2')
    for rs in ('nearest', 'average', 'bilinear', 'cubic_spline'):
        ident = onp.reproject(a, np.nan, (1., 0., 1., 0.), (20, 10), resampling=rs)
        if rs == 'cubic_spline':   # the B-spline smooths: identity only for the mask and for linear ramps
            assert (np.isnan(ident) == np.isnan(a)).all()
            assert ident[5:15, 3:7] == pytest.approx(a[5:15, 3:7], abs=1e-4)
        else:
            assert_same_f32(ident, a, f'{rs} identity')
    # partition of unity: constants are preserved wherever anything valid contributes, for odd scales and offsets
    c = np.full((9, 11), 3.25, np.float32)
    c[4, 5] = np.nan
    for rs, m in (('cubic_spline', (.37, .2, .41, .1)), ('bilinear', (.5, .25, .5, .25)), ('average', (2.3, .2, 1.7, .1)),
                  ('average', (.6, 0., .6, 0.))):
        shape = (int(9 / m[2]), int(11 / m[0]))
        out = onp.reproject(c, np.nan, m, shape, resampling=rs)
        # (GDAL skips the renormalisation while the weight sum is within 1 +- 1e-5, hence 1e-4 and not 1e-7)
        assert np.nanmax(np.abs(out - 3.25)) < 1e-4 and (~np.isnan(out)).sum() > 0.5 * out.size, rs
    # average is mean preserving on aligned integer ratios; nodata=None destinations are 0 where empty
    rng = np.random.default_rng(0)
    r = rng.uniform(0, 1, (12, 18)).astype(np.float32)
    avg = onp.reproject(r, None, (3., 0., 3., 0.), (4, 6), resampling='average')
    assert avg == pytest.approx(r.reshape(4, 3, 6, 3).mean(axis=(1, 3)), rel=1e-6)
    off = onp.reproject(r, None, (1., 30., 1., 0.), (12, 18), dst_nodata=None, resampling='nearest')
    assert (off == 0).all()
    # stretched kernels (round 2): down-sampling with a convolution kernel scales its support; constants survive, and a
    # 2:1 bilinear down-sampling of whole cells weights the 4 x 4 neighbourhood (1 3 3 1) x (1 3 3 1) / 64
    for rs in ('bilinear', 'cubic', 'cubic_spline', 'lanczos'):
        out = onp.reproject(np.full((12, 18), 2.5, np.float32), None, (2., 0., 2., 0.), (6, 9), resampling=rs)
        assert np.allclose(out, 2.5, rtol=1e-6), rs
    bl = onp.reproject(r, None, (2., 0., 2., 0.), (6, 9), resampling='bilinear')
    w = np.array([1., 3., 3., 1.])
    assert bl[2, 3] == pytest.approx(float((r[3:7, 5:9].astype(np.float64) * np.outer(w, w)).sum() / 64), rel=1e-6)
    # rank-order methods over whole 2 x 2 cells: med / q1 / q3 = element ceil(q * n - 1) of the sorted cell (n = 4: the 2nd,
    # 1st and 3rd smallest), mode = the value whose count first reaches the maximum in row-major order
    cells = np.sort(r.reshape(6, 2, 9, 2).transpose(0, 2, 1, 3).reshape(6, 9, 4), axis=2)
    for rs, k in (('med', 1), ('q1', 0), ('q3', 2)):
        np.testing.assert_array_equal(onp.reproject(r, None, (2., 0., 2., 0.), (6, 9), resampling=rs), cells[..., k])
    q = np.array([[5, 7, 7, 1], [7, 5, 5, 1], [2, 2, 3, 3], [9, 2, 3, 3]], np.float32)
    np.testing.assert_array_equal(onp.reproject(q, None, (2., 0., 2., 0.), (2, 2), resampling='mode'), [[7, 1], [2, 3]])  # 5 7 / 7 5: 7 is second first
    np.testing.assert_array_equal(onp.reproject(q, None, (4., 0., 4., 0.), (1, 1), resampling='mode'), [[3]])   # 3 occurs four times
    with pytest.raises((NotImplementedError, KeyError, ValueError)):
        onp.reproject(r, None, (2., 0., 2., 0.), (6, 9), resampling='gauss')   # not a warp method (rasterio.enums)
