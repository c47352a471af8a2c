"""
CPU tests of the comparison statistics (homonim/compare.py): the numpy restatement against statistics produced by
the reference's own ``RasterCompare.process`` (tests/golden/compare.npz, oracle/gen_golden.py), and the host-side
``RasterCompare`` surface that needs no GPU.
"""
import json
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from homonim_amd import fuse
from homonim_amd.compare import RasterCompare
from homonim_amd.enums import Resampling
from oracle import oracle_np as onp

# The reference accumulates float32 block sums in completion order (concurrent.futures.as_completed), so its own
# statistics move by ~1e-6 relative from run to run; N is exact.
REL_TOL = 5e-6


def compare_cases():
    g = np.load(os.path.join(GOLDEN_DIR, 'compare.npz'))
    nd = lambda v: None if v is None else (np.nan if v == 'nan' else v)  # noqa: E731
    out = []
    for c in json.loads(bytes(g['cases_json']).decode()):
        out.append(dict(c, src=g[c['name'] + '_src'], ref=g[c['name'] + '_ref'], stats=g[c['name'] + '_stats'],
                        src_nodata=nd(c['src_nodata']), ref_nodata=nd(c['ref_nodata'])))
    return out


def stats_rows(stats):
    return np.array([[v['r2'], v['rmse'], v['rrmse'], v['n']] for v in stats.values()], np.float64)


def assert_stats_close(got, exp, rel=REL_TOL):
    assert got.shape == exp.shape
    assert np.array_equal(got[:, 3], exp[:, 3]), 'pixel counts differ'
    assert np.allclose(got[:, :3], exp[:, :3], rtol=rel, atol=0), np.abs(got[:, :3] / exp[:, :3] - 1).max()


@pytest.mark.parametrize('case', compare_cases(), ids=lambda c: c['name'])
def test_oracle_reproduces_reference_statistics(case):
    nb, h, w = case['src'].shape
    sums = [{} for _ in range(nb)]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        blocks = list(fuse.block_pairs((h, w), nb, (0, 0), case['max_block_mem']))
    assert len(blocks) > nb or case['max_block_mem'] >= 512
    for bp in blocks:
        rs, cs = bp.src_in_block.toslices()
        d = onp.compare_sums(case['src'][bp.band_i][rs, cs], case['src_nodata'], case['ref'][bp.band_i][rs, cs],
                             case['ref_nodata'])
        sums[bp.band_i] = {k: sums[bp.band_i].get(k, 0) + v for k, v in d.items()}
    stats = onp.compare_stats(sums)
    assert list(stats.keys()) == case['bands']
    assert_stats_close(stats_rows(stats), case['stats'])


def test_statistics_from_sums_follow_the_reference_formulas():
    rng = np.random.default_rng(0)
    s = rng.uniform(0, 1, (50, 70)).astype(np.float32)
    r = (1.5 * s + 0.1 + rng.normal(0, 0.05, s.shape)).astype(np.float32)
    s[:3] = np.nan
    sums = [onp.compare_sums(s, np.nan, r, None), onp.compare_sums(r, None, s, np.nan)]
    exp = onp.compare_stats(sums, ['a', 'b'])
    rc = object.__new__(RasterCompare)
    rc._band_names = ['a', 'b']
    got = rc._get_image_stats([{k: float(v) for k, v in d.items()} for d in sums])
    assert list(got.keys()) == ['a', 'b', 'Mean']
    assert_stats_close(stats_rows(got), stats_rows(exp), rel=1e-12)
    assert isinstance(got['Mean']['n'], int) and got['a']['n'] == 47 * 70
    m = ~np.isnan(s)
    assert got['a']['r2'] == pytest.approx(np.corrcoef(s[m].astype('f8'), r[m].astype('f8'))[0, 1] ** 2, rel=1e-5)
    assert got['a']['rmse'] == pytest.approx(np.sqrt(np.mean((r[m].astype('f8') - s[m]) ** 2)), rel=1e-5)
    # no valid pixel: NaN statistics, N = 0, no exception
    empty = RasterCompare._band_stats()
    assert empty['n'] == 0 and np.isnan(empty['r2']) and np.isnan(empty['rmse'])


def test_config_schema_and_tables():
    cfg = RasterCompare.create_config()
    assert cfg['max_block_mem'] == 512 and cfg['downsampling'] == Resampling.average
    assert cfg['upsampling'] == Resampling.cubic_spline and cfg['threads'] >= 1
    with pytest.raises(TypeError):
        RasterCompare.create_config(unknown=1)
    rc = object.__new__(RasterCompare)
    assert rc._get_resampling((1, 1), (2, 2)) == Resampling.average          # to the coarser grid
    assert rc._get_resampling((2, 2), (1, 1)) == Resampling.cubic_spline     # to the finer grid
    assert rc._get_resampling((2, 2), (1, 1), upsampling=Resampling.bilinear) == Resampling.bilinear
    assert list(RasterCompare.schema) == ['r2', 'rmse', 'rrmse', 'n']
    table = RasterCompare.stats_table({'B1': dict(r2=0.5, rmse=1.25, rrmse=0.1, n=10), 'Mean': dict(r2=0.5, rmse=1.25, rrmse=0.1, n=10)})
    assert 'RMSE' in table and 'B1' in table and '1.250' in table
    assert 'ABBREV' in RasterCompare.schema_table()
