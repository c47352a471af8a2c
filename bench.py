"""
bench.py -- headline benchmark of the homonim kernel-model hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): synthetic float32 4-band
16384 x 16384 source / reference pair resident in HBM, Model.gain_offset, 5x5 kernel, R2 + r2-mask test
(r2_inpaint_thresh 0.25), fused fit+apply.  One "step" = one pass of the hot path over the whole 4-band raster = ONE
kernel launch.  Each rank owns one GPU and its own raster (tiles/bands are independent: no data-path collective),
so scaling is weak and value = units processed by all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
PyTorch is used only for the multi-process rendezvous/barrier when N > 1 (torch.distributed, backend nccl = RCCL).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ALGO_BYTES_PER_PX = 12   # read src 4 + read ref 4 + write corrected 4


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--size', type=int, default=16384, help='raster height = width')
    p.add_argument('--bands', type=int, default=4)
    p.add_argument('--model', default='gain-offset', choices=['gain', 'gain-blk-offset', 'gain-offset'])
    p.add_argument('--kernel', type=int, default=5)
    p.add_argument('--seg-rows', type=int, default=0)
    p.add_argument('--nodata', type=int, default=0, help='0: no nodata, 1: NaN frame + 0.1%% holes, 2: NaN frame only, 3 / 4: no nodata, noisy reference (35 %% / 85 %% of the pixels fail the r2 mask)')
    p.add_argument('--no-thresh', action='store_true', help='gain-offset without r2_inpaint_thresh (no R2 work)')
    p.add_argument('--params', action='store_true', help='also materialise the gain / offset / R2 planes in the fused launch (find_r2=True; 24 B per pixel*band of HBM traffic, reported against the same 12 algorithmic bytes)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-parity', action='store_true')
    p.add_argument('--cpu-sample', type=int, default=0, help='CPU baseline sample size (square); 0 = auto')
    return p.parse_args()


def cpu_baseline(model, k, sample):
    """ Times the oracle (the CPU restatement of the reference path; kind = "port") on a bounded sample of the same
    workload.  Prefers the compiled C oracle (OpenMP, all host cores), falls back to the numpy one (1 core). """
    from oracle import oracle_np as onp
    try:
        from oracle import oracle_c
        have_c = oracle_c.available()
    except Exception:
        have_c = False
    if sample <= 0:
        sample = (8192 if (os.cpu_count() or 1) >= 32 else 4096) if have_c else 1536
    src, ref = onp.synth_pair(sample, sample, seed=0)
    thresh = 0.25 if model == 'gain-offset' else None
    if have_c:
        cores = os.cpu_count() or 1
        oracle_c.fit_apply(model, src, np.nan, ref, np.nan, (k, k), False, thresh, n_threads=cores)  # warm-up
        t0 = time.perf_counter()
        reps = 0
        while True:
            oracle_c.fit_apply(model, src, np.nan, ref, np.nan, (k, k), False, thresh, n_threads=cores)
            reps += 1
            if time.perf_counter() - t0 > 10.0 or reps >= 40:
                break
        dt = (time.perf_counter() - t0) / reps
        impl = f'C oracle (oracle/hk_oracle.c, OpenMP {cores} threads)'
    else:
        cores = 1
        t0 = time.perf_counter()
        params, _ = onp.fit(model, src, np.nan, ref, np.nan, (k, k), False, thresh)
        onp.apply(src, params)
        dt = time.perf_counter() - t0
        impl = 'numpy oracle (oracle/oracle_np.py)'
    return dict(value=round(sample * sample / dt / 1e6, 3), unit='Mpixels*bands/s', cores=cores, kind='port',
                sample=f'{sample}x{sample} float32 1-band block of the same synthetic workload, {model} {k}x{k} '
                       f'fit+apply, {impl}, {dt:.3f} s per pass')


def parity_spot_check(ctx, args, bufs, stride, band_stride, thresh, n_fail=0):
    """ Not timed: download a window of band 0 and compare the GPU output with the numpy oracle. """
    from oracle import oracle_np as onp
    H = W = args.size
    k = args.kernel
    r = k // 2
    wh, ww = min(H, 384), min(W, 1200)
    y0 = max(0, min(H - wh, H // 3))
    x0 = max(0, min(W - ww, (W // 2) // 4 * 4))
    rows = np.empty((wh, stride), np.float32)
    win = {}
    for name in ('src', 'ref', 'corr'):
        ctx.d2h(rows, bufs[name] + 4 * (y0 * stride))
        win[name] = rows[:, x0:x0 + ww].copy()
    nodata = np.nan if args.nodata in (1, 2) else None
    norm = None
    if args.model == 'gain-blk-offset':
        norm = bufs['norm_host'][0]
    params, _ = onp.fit(args.model, win['src'], nodata, win['ref'], nodata, (k, k), False, thresh, norm_model=norm)
    exp = onp.apply(win['src'], params)
    # windows of interior pixels see the same data as on the GPU; drop the r-px rim of the downloaded window (plus the
    # 100-px search radius of the in-painting when pixels failed the r2 mask: it looks that far for passing neighbours)
    if n_fail:
        r += 101
    sl = (slice(r if y0 > 0 else 0, wh - r if y0 + wh < H else wh), slice(r if x0 > 0 else 0, ww - r if x0 + ww < W else ww))
    got, exp = win['corr'][sl], exp[sl]
    nan_ok = bool((np.isnan(got) == np.isnan(exp)).all())
    ok = ~np.isnan(exp)
    rel = float(np.max(np.abs(got[ok] - exp[ok]) / np.maximum(np.abs(exp[ok]), 1e-30))) if ok.any() else 0.0
    n_diff = int((got[ok] != exp[ok]).sum())
    return dict(window=[int(got.shape[0]), int(got.shape[1])], bitwise_mismatches=n_diff, max_rel_diff=rel,
                nan_pattern_equal=nan_ok, passed=bool(nan_ok and rel <= 1e-5))


def measured_traffic(args):
    """ HBM bytes per launch from the committed rocprofv3 PMC summary (collected in separate --pmc passes, FETCH_SIZE
    corrected x2 for gfx950) when it was taken at exactly this configuration; None otherwise. """
    try:
        with open(os.path.join(REPO, 'profiles', 'pmc_summary.json')) as f:
            pmc = json.load(f)
        c = pmc['config']
        if (c['model'], c['kernel'], c['size'], c['bands'], c['nodata'], bool(c.get('no_thresh', False))) == (
                args.model, args.kernel, args.size, args.bands, args.nodata, bool(args.no_thresh)) and not args.params:
            return float(pmc['hbm_traffic_bytes'])
    except Exception:
        pass
    return None


def main():
    args = parse_args()
    from homonim_amd import _hk, dist
    rank, world, local_rank = dist.init()  # torch.distributed (nccl = RCCL) only when WORLD_SIZE > 1
    n_gpus = args.gpus
    ctx = _hk.Context(local_rank % max(1, _hk.device_count()), n_streams=2)  # one GPU per rank on a full node
    ctx.selftest()

    H = W = args.size
    B = args.bands
    k = args.kernel
    stride = (W + 63) // 64 * 64
    band_stride = stride * H
    plane_bytes = 4 * band_stride * B
    thresh = 0.25 if (args.model == 'gain-offset' and not args.no_thresh) else None
    nd = np.nan if args.nodata in (1, 2) else None
    desc = _hk.make_desc(args.model, (k, k), bool(args.params), thresh, nd, nd)

    bufs = {name: ctx.dev_alloc(plane_bytes) for name in ('src', 'ref', 'corr') + (('gain', 'offset', 'r2') if args.params else ())}
    bufs['fail'] = ctx.dev_alloc(8 * B)
    bufs['norm'] = ctx.dev_alloc(16 * B)
    ctx.memset(bufs['fail'], 0, 8 * B)
    ctx.synth_fill_dev(bufs['src'], bufs['ref'], B, H, W, stride, band_stride, seed=1234 + rank,
                       nodata_variant=args.nodata, stream=0)
    ctx.stream_sync(0)

    job = _hk.DevJob()
    job.src, job.ref, job.corr = bufs['src'], bufs['ref'], bufs['corr']
    job.gain = job.offset = job.r2 = None
    if args.params:
        job.gain, job.offset, job.r2 = bufs['gain'], bufs['offset'], bufs['r2']
    job.fail_count = bufs['fail']
    job.norm = bufs['norm'] if args.model == 'gain-blk-offset' else None
    job.n_bands, job.height, job.width, job.stride, job.band_stride = B, H, W, stride, band_stride
    job.seg_rows, job.stream = args.seg_rows, 0

    # Two sets of failure counters + pinned host copies: the r2-mask check of step i (the reference's `if not all(mask)`,
    # kernel_model.py:363-371) reads counters that were copied to the host behind launch i, AFTER launch i + 1 has been
    # queued -- the stream never drains between steps, and the in-painting passes (if a band has failures; none on the
    # clean synthetic workload) are queued behind launch i + 1 and recompute their band from src / ref.
    fail_dev = [bufs['fail'], ctx.dev_alloc(8 * B)]
    bufs['fail2'] = fail_dev[1]
    ctx.memset(fail_dev[1], 0, 8 * B)
    fail_host = [ctx.pinned_empty((B,), np.uint64) for _ in range(2)]
    fail_ready = [ctx.event(), ctx.event()]

    def launch(i):
        job.fail_count = fail_dev[i % 2]
        if args.model == 'gain-blk-offset':
            ctx.block_norm_dev(desc, job, bufs['norm'])  # the block statistics are part of the fit
        ctx.fit_apply_dev(desc, job)

    def queue_check(i):
        if thresh is not None:
            job.fail_count = fail_dev[i % 2]
            ctx.fail_counts_async(job, fail_host[i % 2], fail_ready[i % 2])

    def finish(i):
        """ the host side of step i's r2-mask check; returns the number of failing pixels """
        if thresh is None:
            return 0
        ctx.event_sync(fail_ready[i % 2])
        counts = fail_host[i % 2].copy()
        job.fail_count = fail_dev[i % 2]  # a band the certificate-only build gave up on is re-counted here
        return ctx.inpaint_dev_counts(desc, job, counts) if counts.any() else 0

    def run(n_steps, events=None):
        n_fail = 0
        for i in range(n_steps):
            if events:
                ctx.event_record(events[i][0], 0)
            launch(i)
            if events:
                ctx.event_record(events[i][1], 0)
            queue_check(i)
            if i > 0:
                n_fail += finish(i - 1)
        if n_steps:
            n_fail += finish(n_steps - 1)
        return n_fail

    barrier = dist.barrier

    run(args.warmup)
    ctx.stream_sync(0)

    events = [(ctx.event(), ctx.event()) for _ in range(args.steps)]
    barrier()
    ctx.sync()
    t0 = time.perf_counter()
    n_fail = run(args.steps, events)
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0

    launch_ms = [ctx.event_elapsed_ms(e0, e1) for e0, e1 in events]
    elapsed = dist.max_over_ranks(elapsed)
    n_fail //= max(1, args.steps)

    px_bands = H * W * B
    value = px_bands * args.steps * world / elapsed / 1e6
    avg_ms = float(np.mean(launch_ms))
    achieved = ALGO_BYTES_PER_PX * px_bands / (avg_ms * 1e-3) / 1e9

    out = None
    if rank == 0:
        parity = None
        if not args.no_parity:
            if args.model == 'gain-blk-offset':
                nh = np.zeros((B, 2), np.float64)
                ctx.d2h(nh, bufs['norm'])
                bufs['norm_host'] = nh
            parity = parity_spot_check(ctx, args, bufs, stride, band_stride, thresh, n_fail)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.model, k, args.cpu_sample)
        out = {
            'metric': 'Mpixels*bands/s fit+apply (5x5 gain-offset, float32)' if (args.model == 'gain-offset' and k == 5)
                      else f'Mpixels*bands/s fit+apply ({k}x{k} {args.model}, float32)',
            'value': round(value, 1), 'unit': 'Mpixels*bands/s', 'n_gpus': n_gpus, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 in/out, f64 window sums', 'data': 'synthetic',
            'config': {
                'workload': f'synthetic float32 {B}-band {H}x{W} src/ref resident in HBM per GPU, Model.{args.model}, '
                            f'kernel {k}x{k}, r2_inpaint_thresh {thresh}, fused fit+apply'
                            + (' + gain / offset / R2 planes written' if args.params else '')
                            + (' (BASELINE.json configs[2])' if (args.model, k, H, B, args.nodata, args.no_thresh, args.params)
                               == ('gain-offset', 5, 16384, 4, 0, False, False) else ''),
                'bands': B, 'height': H, 'width': W, 'nodata_variant': args.nodata,
                'parallelism': f'{world} rank(s) x 1 GPU, independent rasters, no collective',
                'r2_mask_failures_per_step': n_fail,
            },
            'roofline': {
                'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBPS, 4), 'traffic': measured_traffic(args),
                'kernel': 'hk::fit_apply_kernel', 'avg_launch_ms': round(avg_ms, 4),
                'algorithmic_bytes_per_launch': ALGO_BYTES_PER_PX * px_bands,
            },
            'cpu_baseline': cpu,
            'parity_spot_check': parity,
        }
        print(json.dumps(out), flush=True)

    for pair in events + [tuple(fail_ready)]:
        for e in pair:
            ctx.event_destroy(e)
    for name in ('src', 'ref', 'corr', 'fail', 'fail2', 'norm'):
        ctx.dev_free(bufs[name])
    ctx.close()
    dist.finalize()


if __name__ == '__main__':
    main()
