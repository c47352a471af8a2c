"""
End-to-end (PCIe-inclusive) throughput of the host-pointer path: BASELINE.json configs[4]-style mosaic of independent
4096 x 4096 band-tiles living in (pinned) host memory, one tile per call, T host threads sharing one context whose
pooled streams overlap H2D / kernel / D2H of different tiles.  NOT the headline number (bench.py, HBM-resident).

    python tools/bench_streamed.py [--tiles 64] [--bands 4] [--threads 4] [--streams 4] [--pageable]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homonim_amd import _hk  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--tiles', type=int, default=64)
    p.add_argument('--bands', type=int, default=4)
    p.add_argument('--size', type=int, default=4096)
    p.add_argument('--threads', type=int, default=4)
    p.add_argument('--streams', type=int, default=4)
    p.add_argument('--distinct', type=int, default=8, help='distinct host tile buffers (re-used round-robin)')
    p.add_argument('--model', default='gain-offset')
    p.add_argument('--kernel', type=int, default=5)
    p.add_argument('--pageable', action='store_true', help='plain numpy arrays instead of pinned ones')
    p.add_argument('--refspace', type=int, default=0, help='N > 0: reference tiles N x coarser than the source tiles; fused RefSpaceModel pipeline')
    p.add_argument('--dtype', default='float32', help='host raster dtype in and out (float32 | uint8 | uint16 | int16)')
    args = p.parse_args()

    ctx = _hk.Context(int(os.environ.get('LOCAL_RANK', '0')), n_streams=args.streams)
    n = args.size
    dt = np.dtype(args.dtype)
    alloc = (lambda shape: np.empty(shape, dt)) if args.pageable else (lambda shape: ctx.pinned_empty(shape, dt))
    rng = np.random.default_rng(0)
    srcs, refs = [], []
    for i in range(args.distinct):
        s, r = alloc((n, n)), alloc((n, n))
        if dt.kind == 'f':
            s[:] = rng.uniform(0.05, 1.0, (n, n)).astype(dt)
            r[:] = (1.2 * s + 0.05 + rng.normal(0, 0.01, (n, n))).astype(dt)
        else:
            s[:] = rng.integers(10, 200, (n, n)).astype(dt)
            r[:] = np.clip(np.round(1.2 * s + 5 + rng.normal(0, 2, (n, n))), 0, 255).astype(dt)
        srcs.append(s)
        refs.append(r)
    outs = [alloc((n, n)) for _ in range(args.threads)]
    if args.refspace:
        f = args.refspace
        m = n // f
        coarse = []
        for r in refs:   # block-average the full-resolution reference to the coarse grid
            c = alloc((m, m))
            c[:] = r[:m * f, :m * f].reshape(m, f, m, f).astype(np.float64).mean(axis=(1, 3)).round().astype(dt) \
                if dt.kind != 'f' else r[:m * f, :m * f].reshape(m, f, m, f).mean(axis=(1, 3)).astype(dt)
            coarse.append(c)
        refs = coarse
    thresh = 0.25 if args.model == 'gain-offset' else None
    desc = _hk.make_desc(args.model, (args.kernel, args.kernel), False, thresh, None, None)
    n_param = 3 if thresh is not None else 2
    work = list(range(args.tiles * args.bands))

    def worker(tid):
        fails = 0
        for w in work[tid::args.threads]:
            i = w % args.distinct
            if args.refspace:
                k = float(args.refspace)
                _, _, f = ctx.refspace_fit_apply(desc, srcs[i], refs[i], (k, 0., k, 0.), (1 / k, 0., 1 / k, 0.), 5, 3,
                                                 False, n_param, False, out_dtype=dt.name,
                                                 out_nodata=None if dt.kind == 'f' else 0, out_corr=outs[tid])
            else:
                _, _, _, f = ctx.fit_apply(desc, srcs[i], refs[i], n_param, want_params=False, want_corr=True,
                                           out_corr=outs[tid], out_dtype=dt.name, out_nodata=None if dt.kind == 'f' else 0)
            fails += f
        return fails

    with ThreadPoolExecutor(args.threads) as ex:
        list(ex.map(worker, range(args.threads)))  # warm-up: grows the per-stream device slabs
        t0 = time.perf_counter()
        fails = sum(ex.map(worker, range(args.threads)))
        dt_s = time.perf_counter() - t0
    px = len(work) * n * n
    print(json.dumps(dict(
        metric='Mpixels*bands/s fit+apply end-to-end incl. PCIe (host-resident tiles)', value=round(px / dt_s / 1e6, 1),
        seconds=round(dt_s, 3), tiles=args.tiles, bands=args.bands, tile=n, threads=args.threads, streams=args.streams,
        pinned=not args.pageable, model=args.model, kernel=args.kernel, r2_mask_failures=int(fails),
        refspace=args.refspace, dtype=dt.name, pcie_gbps_in=round(px * 2 * dt.itemsize / dt_s / 1e9, 1),
        pcie_gbps_out=round(px * dt.itemsize / dt_s / 1e9, 1))))
    ctx.close()


if __name__ == '__main__':
    main()
