""" Worker of tests/test_gpu_staging.py: the host-pointer entry points on fixed inputs, results into an .npz.  Run in a process
of its own so that HK_STAGE_CHUNK_KB (read once by the library) can shrink the pinned staging chunks to a few device rows. """
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from homonim_amd import _hk  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402  (inputs only: the seeded generator shared with the goldens)


def run(out_path: str):
    ctx = _hk.Context(0, 2)
    res = {}
    src, ref = onp.synth_pair(333, 517, 9, 'frame+holes')
    desc = _hk.make_desc('gain-offset', (5, 5), True, 0.25, np.nan, np.nan)
    params, corr, _, nf = ctx.fit_apply(desc, src, ref, 3, want_params=True, want_corr=True)
    res['go_params'], res['go_corr'], res['go_nf'] = params, corr, np.array([nf])
    # strided (non-contiguous rows) inputs: a window of a larger raster
    big_s, big_r = onp.synth_pair(400, 700, 3, 'none')
    desc2 = _hk.make_desc('gain-blk-offset', (3, 3), False, None, None, None)
    p2, c2, n2, _ = ctx.fit_apply(desc2, big_s[7:390, 11:650], big_r[7:390, 11:650], 2, want_params=True, want_corr=True)
    res['blk_params'], res['blk_corr'], res['blk_norm'] = p2, c2, n2
    # typed rasters: uint8 in, uint8 out with nodata
    s8 = np.clip(src * 200, 0, 255).astype(np.uint8)
    r8 = np.clip(ref * 200, 0, 255).astype(np.uint8)
    desc3 = _hk.make_desc('gain', (5, 5), False, None, 0, None)
    _, c8, _, _ = ctx.fit_apply(desc3, s8, r8, 2, want_params=False, want_corr=True, out_dtype='uint8', out_nodata=0)
    res['u8_corr'] = c8
    # mask_partial with the uint8 mask (rows of 10 / 20 bytes: the round-3 abort's call) and a large one
    for name, (h, w) in (('mask_small', (20, 10)), ('mask_mid', (40, 20)), ('mask_big', (301, 999))):
        rng = np.random.default_rng(h)
        a = rng.random((h, w), dtype=np.float32)
        a[rng.random((h, w)) < 0.02] = np.nan
        par = rng.random((2, h, w), dtype=np.float32)
        par[:, rng.random((h, w)) < 0.01] = np.nan
        pm, cm, mm = ctx.partial_mask(a, np.nan, par, (5, 5), src=a, want_params=True, want_corr=True, want_mask=True)
        res[name + '_p'], res[name + '_c'], res[name + '_m'] = pm, cm, mm
    # the block loop of RasterFuse.process: out-windows of blocks with halos, written straight into strided caller rasters
    # (hk_fit_apply_block; pageable arrays, no registration)
    import warnings
    from homonim_amd.fuse import RasterFuse
    cube_s = np.stack([onp.synth_pair(300, 410, 40 + b, 'frame+holes')[0] for b in range(3)])
    cube_r = np.stack([onp.synth_pair(300, 410, 40 + b, 'frame+holes')[1] for b in range(3)])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        corr_f, par_f = RasterFuse(cube_s, cube_r, src_nodata=np.nan, ref_nodata=np.nan).process(
            None, 'gain-offset', (5, 5), param_filename=True, block_config=dict(threads=2, max_block_mem=0.05),
            device_config=dict(devices=[0], streams=2, pin=False))
    res['fuse_corr'], res['fuse_params'] = np.asarray(corr_f), np.asarray(par_f)
    res['apply'] = ctx.apply(src, params[:2])
    res['reproject'] = ctx.reproject(np.stack([src, ref]), np.nan, (2.0, 0.0, 2.0, 0.0), (166, 258), 5, np.nan)
    res['sums'] = ctx.compare_sums(src, np.nan, ref, np.nan)
    res['norm'] = ctx.block_norm(desc2, big_s, big_r)
    # raw transfers of an odd number of bytes, larger than any chunk
    blob = np.random.default_rng(1).integers(0, 256, 9_000_011, dtype=np.uint8)
    d = ctx.dev_alloc(blob.nbytes)
    ctx.h2d(d, blob)
    back = np.empty_like(blob)
    ctx.d2h(back, d)
    ctx.dev_free(d)
    res['blob_equal'] = np.array([int(np.array_equal(back, blob))])
    np.savez(out_path, **res)
    ctx.close()


if __name__ == '__main__':
    run(sys.argv[1])
