"""
CPU tests: the plain-C oracle (oracle/hk_oracle.c) against the reference goldens and against the numpy oracle.
Bar: bit-exact float32 (it keeps oracle_np's summation order), for any thread count.
"""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, assert_same_f32, case_id
from oracle import oracle_np as onp


@pytest.fixture(scope='module')
def oc():
    from homonim_amd import build
    build.build_oracle(verbose=False)
    from oracle import oracle_c
    assert oracle_c.available()
    return oracle_c


@pytest.mark.parametrize('case', GOLDEN_CASES, ids=case_id)
def test_oracle_c_matches_reference_goldens(oc, goldens, case):
    src = goldens[f"in_{case['variant']}_src"]
    ref = goldens[f"in_{case['variant']}_ref"]
    norm = goldens[f"{case['name']}_norm"] if case['model'] == 'gain-blk-offset' else None
    params, corr, n_fail = oc.fit_apply(case['model'], src, case['src_nodata'], ref, case['ref_nodata'],
                                        case['kernel_shape'], case['find_r2'], case['r2_inpaint_thresh'], norm)
    assert_same_f32(params, goldens[f"{case['name']}_params"], 'params')
    assert_same_f32(corr, goldens[f"{case['name']}_corr"], 'corrected')
    assert n_fail == 0


@pytest.mark.parametrize('model, k, find_r2, thresh', [
    ('gain', (3, 3), True, None), ('gain-offset', (5, 5), True, 0.25), ('gain-offset', (15, 15), False, None),
    ('gain-blk-offset', (5, 7), True, None),
])
@pytest.mark.parametrize('n_threads', [1, 3, 8])
def test_oracle_c_equals_oracle_np_any_thread_count(oc, model, k, find_r2, thresh, n_threads):
    src, ref = onp.synth_pair(211, 333, 5, 'frame+holes')
    norm = onp.fit_block_norm(src, np.nan, ref, np.nan) if model == 'gain-blk-offset' else None
    exp, aux = onp.fit(model, src, np.nan, ref, np.nan, k, find_r2, thresh, norm_model=norm)
    params, corr, n_fail = oc.fit_apply(model, src, np.nan, ref, np.nan, k, find_r2, thresh, norm, n_threads=n_threads)
    assert_same_f32(params, exp, 'params')
    assert_same_f32(corr, onp.apply(src, exp), 'corrected')
    if model == 'gain-offset' and thresh is not None:
        assert n_fail == aux


def test_oracle_c_fail_count(oc):
    src, ref = onp.synth_pair(64, 96, 9)
    ref = ref.copy()
    ref[30, 40] = -100
    exp, n_np = onp.fit_gain_offset(src, None, ref, None, (5, 5), True, 0.5)
    params, _, n_c = oc.fit_apply('gain-offset', src, None, ref, None, (5, 5), True, 0.5)
    assert n_c == n_np > 0
    assert_same_f32(params, exp, 'params')


def test_oracle_c_block_norm_close_to_numpy(oc):
    """ float64 statistics vs numpy's float32 pairwise ones: ~5e-7 relative. """
    for variant, nodata in (('frame+holes', np.nan), ('none', None)):
        src, ref = onp.synth_pair(300, 400, 2, variant)
        n_np = onp.fit_block_norm(src, nodata, ref, nodata)
        n_c = oc.fit_block_norm(src, nodata, ref, nodata)
        assert n_c[0] == pytest.approx(n_np[0], rel=2e-6)
        assert n_c[1] == pytest.approx(n_np[1], rel=2e-6, abs=2e-6)
    allnan = np.full((8, 8), np.nan, np.float32)
    assert (oc.fit_block_norm(allnan, np.nan, allnan, np.nan) == 0).all()


def test_oracle_c_inpaint_branch_equals_oracle_np(oc):
    """ gain-offset with failing pixels: the in-paint branch (restated GDAL fill + gain recomputation,
    kernel_model.py:361-371) of the C oracle against the numpy one, bit for bit. """
    src, ref = onp.synth_pair(60, 90, 17, 'frame+holes')
    ref = ref.copy()
    ref[20:26, 30:40] = -3
    ref[10, 12] = 9
    ref[45:47, :] = 0.5
    exp, n_np = onp.fit_gain_offset(src, np.nan, ref, np.nan, (5, 5), False, 0.25)
    params, corr, n_c = oc.fit_apply('gain-offset', src, np.nan, ref, np.nan, (5, 5), False, 0.25)
    assert n_c == n_np > 100
    assert_same_f32(params, exp, 'in-painted params')
    assert_same_f32(corr, onp.apply(src, exp), 'corrected')
    _, corr_only, _ = oc.fit_apply('gain-offset', src, np.nan, ref, np.nan, (5, 5), False, 0.25, want_params=False)
    assert_same_f32(corr_only, corr, 'corrected without parameter output')
    rng = np.random.default_rng(0)
    img = rng.normal(size=(50, 70)).astype(np.float32)
    msk = rng.random((50, 70)) < 0.05
    assert_same_f32(oc.fill_nodata(img, msk), onp.fill_nodata(img, msk), 'fill_nodata')
