"""
bench.py -- benchmarks of the homonim kernel-model hot path on MI355X.

    python bench.py [--config N] [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config N]

--config (index into BASELINE.json `configs`; default 2, the configuration the metric is quoted on):
  1  synthetic float32 4-band 8192 x 8192, Model.gain 5x5, resident in HBM; one step = one fused launch.
  2  synthetic float32 4-band 16384 x 16384, Model.gain_offset 5x5 (R2 + r2-mask test, r2_inpaint_thresh 0.25),
     resident in HBM; one step = one fused launch + the host's look at the failure counters.          [HEADLINE]
  3  synthetic float32 8-band 16384 x 16384, Model.gain_blk_offset 15x15, cut into the reference's own blocks
     (4096 x 4096 + 8-pixel halo, homonim/raster_pair.py:342-428), every block normalised by its own statistics and
     processed in place in the resident raster; blocks are dealt to the ranks in band-major runs.  One step = all
     128 blocks (STRONG scaling: the work is fixed, ranks split it).
  4  mosaic of 64 independent 4-band 4096 x 4096 tiles, Model.gain_offset 5x5, one tile per stream, tiles dealt to the
     ranks (strong scaling).  One step = all tiles.
Configs 1 and 2 run one raster per rank (weak scaling: tiles / bands are independent, there is no data-path collective).

`value` is always the device-resident rate (inputs in HBM when the timed region starts).  Configs 3 and 4 add
`end_to_end`: the same blocks / tiles through RasterFuse.process from page-locked HOST rasters (H2D || kernel || D2H on
the context's streams) -- the PCIe-inclusive rate, never the headline.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
No PyTorch anywhere: the ranks of a launch meet over loopback TCP (homonim_amd/dist.py); RCCL is the library's own communicator.
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

CONFIGS = {
    1: dict(model='gain', kernel=5, size=8192, bands=4),
    2: dict(model='gain-offset', kernel=5, size=16384, bands=4),
    3: dict(model='gain-blk-offset', kernel=15, size=16384, bands=8, batches=4),
    4: dict(model='gain-offset', kernel=5, size=4096, bands=4, tiles=64),
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=None)
    p.add_argument('--warmup', type=int, default=None)
    p.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS))
    p.add_argument('--size', type=int, default=None, help='raster (config 4: tile) height = width')
    p.add_argument('--bands', type=int, default=None)
    p.add_argument('--tiles', type=int, default=None, help='config 4: number of tiles of the mosaic')
    p.add_argument('--model', default=None, choices=['gain', 'gain-blk-offset', 'gain-offset'])
    p.add_argument('--kernel', type=int, default=None)
    p.add_argument('--seg-rows', type=int, default=0)
    p.add_argument('--batches', type=int, default=None,
                   help='configs 3 / 4: the jobs of a step in this many batched launches, one stream each (0 = one launch per job; '
                        'default 4 for config 3: -4 %%, 0 for config 4: no difference, profiles/r03_batch.txt)')
    p.add_argument('--nodata', type=int, default=0, help='0: no nodata, 1: NaN frame + 0.1%% holes, 2: NaN frame only, 3 / 4: no nodata, noisy reference (35 %% / 85 %% of the pixels fail the r2 mask), 5: low-entropy data (64 source levels, exactly affine reference; the same instruction stream at a lower energy per launch), 6: NaN frame + ~1 %% of the area in round NaN blobs 32 - 128 px across, src and ref independently (cloud / shadow-mask-like)')
    p.add_argument('--no-thresh', action='store_true', help='gain-offset without r2_inpaint_thresh (no R2 work)')
    p.add_argument('--params', action='store_true', help='also materialise the gain / offset / R2 planes in the fused launch (find_r2=True; 24 B per pixel*band of HBM traffic, reported against the same 12 algorithmic bytes)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-parity', action='store_true')
    p.add_argument('--no-end-to-end', action='store_true', help='configs 3 / 4: skip the PCIe-inclusive RasterFuse pass')
    p.add_argument('--no-nan-variant', action='store_true', help='default run: skip the second, shorter measurement on NaN-nodata rasters')
    p.add_argument('--cpu-sample', type=int, default=0, help='CPU baseline sample size (square); 0 = auto')
    p.add_argument('--no-other-configs', action='store_true', help='default run: skip the compact records of BASELINE configs 1 / 3 / 4')
    p.add_argument('--no-power-probe', action='store_true', help='default run: skip the package power / clock samples (rocm-smi, ~3 s outside the timed region)')
    p.add_argument('--as-rank', default=None, metavar='R/N',
                   help='configs 3 / 4 on ONE GPU: run exactly the shard rank R of an N-rank launch would run, alone (no process group, no '
                        'peers) -- a projection aid for strong scaling, never a scaling measurement')
    p.add_argument('--checksum', action='store_true',
                   help='add `shard_checksum` to the line: the exact checksum (sum of the float32 bit patterns, modulo 2^64) and the pixel '
                        'count of every corrected pixel this launch produced -- per rank and over all ranks; the union of N ranks\' shards '
                        'equals the single-rank result iff the totals do (always on for N > 1)')
    p.add_argument('--no-projection', action='store_true',
                   help='configs 3 / 4 at one rank: skip the single-GPU projection (every rank\'s shard of an N = 2 / 4 / 8 launch timed alone)')
    args = p.parse_args()
    preset = CONFIGS[args.config]
    for k, v in preset.items():
        if getattr(args, k, None) is None:
            setattr(args, k, v)
    if args.batches is None:
        args.batches = 0
    if args.steps is None:
        args.steps = {1: 100, 2: 50, 3: 10, 4: 5}[args.config]
    if args.warmup is None:
        args.warmup = {1: 5, 2: 5, 3: 2, 4: 1}[args.config]
    args.power_probe = not args.no_power_probe and args.config in (1, 2)
    if args.as_rank is not None:
        try:
            r, n = (int(v) for v in args.as_rank.split('/'))
            assert 0 <= r < n
        except Exception:
            p.error('--as-rank wants R/N with 0 <= R < N')
        if args.config not in (3, 4) or args.gpus != 1:
            p.error('--as-rank applies to --config 3 / 4 on one GPU')
        args.as_rank = (r, n)
    return args


class SoloDist:
    """ The process group of a rank that runs alone (`--as-rank R/N`, the single-GPU projection): nobody to wait for. """

    @staticmethod
    def barrier():
        pass

    @staticmethod
    def max_over_ranks(value):
        return float(value)

    @staticmethod
    def backend():
        return None


def timed_steps(ctx, step, steps, warmup):
    """ seconds per step of `step()` on this GPU alone: warm-up, then `steps` steps between two device synchronisations """
    for _ in range(warmup):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    return (time.perf_counter() - t0) / max(1, steps)


def project_scaling(t1_ms, shard_ms_of):
    """ Single-GPU PROJECTION of strong scaling: `shard_ms_of(r, n)` = the time of rank r's shard of an n-rank launch, run alone
    on this GPU.  If the ranks of a real launch do not disturb each other (separate GPUs and HBM stacks; no data-path
    collective), an n-rank step takes the slowest shard's time: speedup = t(1) / max_r t_shard(r, n).  What a projection cannot
    see: host-side contention of n processes (launch threads, PCIe root complexes), clocks of a fully loaded node. """
    out = {}
    for n in (2, 4, 8):
        ms = [shard_ms_of(r, n) for r in range(n)]
        worst = max(ms)
        out[str(n)] = dict(shard_ms=[round(v, 4) for v in ms], max_shard_ms=round(worst, 4),
                           projected_speedup=round(t1_ms / worst, 3) if worst > 0 else None,
                           projected_efficiency=round(t1_ms / (n * worst), 4) if worst > 0 else None)
    return dict(kind='PROJECTION from one GPU -- every shard of an N-rank launch timed ALONE on this GPU; not a scaling measurement',
                t1_ms=round(t1_ms, 4), by_world_size=out)


# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(model, k, sample):
    """ Times the oracle (the CPU restatement of the reference path; kind = "port") on a bounded sample of the same
    workload: the compiled C oracle on all host cores (the headline CPU figure) and on one, and the numpy restatement
    (which mirrors the reference's pass structure) on one thread and under a ThreadPoolExecutor over blocks, the shape
    of homonim/fuse.py:396-408. """
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_np as onp
    try:
        from oracle import oracle_c
        have_c = oracle_c.available()
    except Exception:
        have_c = False
    try:
        cores = len(os.sched_getaffinity(0))   # the CPUs this process may run on (a container's share, not the box's count)
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    if sample <= 0:
        sample = (8192 if cores >= 32 else 4096) if have_c else 1536
    thresh = 0.25 if model == 'gain-offset' else None
    variants = []

    def timed(fn, budget, max_reps):
        fn()
        t0 = time.perf_counter()
        reps = 0
        while True:
            fn()
            reps += 1
            if time.perf_counter() - t0 > budget or reps >= max_reps:
                break
        return (time.perf_counter() - t0) / reps

    def c_pass(s, r, threads):
        # gain-blk-offset: the block statistics are part of the fit (kernel_model.py:289)
        norm = oracle_c.fit_block_norm(s, np.nan, r, np.nan) if model == 'gain-blk-offset' else None
        oracle_c.fit_apply(model, s, np.nan, r, np.nan, (k, k), False, thresh, norm_model=norm, n_threads=threads)

    if have_c:
        # The headline CPU figure has the shape of the reference's own block loop (homonim/fuse.py:396-401: one worker thread
        # per block, os.cpu_count() blocks in flight; raster_pair.py's default 4096 x 4096 blocks): one single-threaded pass
        # of the C oracle per block, `cores` blocks at a time.
        try:
            import psutil
            avail = psutil.virtual_memory().available
        except Exception:
            avail = 32 << 30
        nbk = 4096
        distinct = max(1, min(cores, 8))
        while nbk > 1024 and (cores * 4 + distinct * 2) * 4 * nbk * nbk > 0.5 * avail:
            nbk //= 2      # a small host: smaller blocks rather than fewer of them in flight
        pairs = [onp.synth_pair(nbk, nbk, seed=100 + i) for i in range(distinct)]

        def one_block(i):
            c_pass(pairs[i % distinct][0], pairs[i % distinct][1], 1)

        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one_block, range(cores)))          # warm-up round (page faults of the outputs, thread start)
            t0 = time.perf_counter()
            rounds = 0
            while True:
                list(ex.map(one_block, range(cores)))
                rounds += 1
                if time.perf_counter() - t0 > 6.0 or rounds >= 8:
                    break
            dtb = (time.perf_counter() - t0) / rounds
        del pairs
        main = dict(value=round(cores * nbk * nbk / dtb / 1e6, 3), cores=cores,
                    impl=f'C oracle (oracle/hk_oracle.c), one thread per {nbk}x{nbk} block, {cores} blocks in flight '
                         f'(the shape of homonim/fuse.py:396-401)', sample_px=nbk, seconds=round(dtb, 3))
        src, ref = onp.synth_pair(sample, sample, seed=0)
        dt = timed(lambda: c_pass(src, ref, cores), 5.0, 40)
        variants.append(dict(value=round(sample * sample / dt / 1e6, 3), cores=cores,
                             impl=f'C oracle, ONE {sample}x{sample} block row-sliced over {cores} OpenMP threads (round 2\'s headline CPU figure)',
                             sample=f'{sample}x{sample}', seconds=round(dt, 3)))
        s1 = min(sample, 2048)
        src1, ref1 = np.ascontiguousarray(src[:s1, :s1]), np.ascontiguousarray(ref[:s1, :s1])
        dt1 = timed(lambda: c_pass(src1, ref1, 1), 3.0, 5)
        variants.append(dict(value=round(s1 * s1 / dt1 / 1e6, 3), cores=1, impl='C oracle, 1 thread',
                             sample=f'{s1}x{s1}', seconds=round(dt1, 3)))
    # numpy restatement: one thread on one block, then min(cores, 16) threads with one block each
    nb = 1024
    blocks = [onp.synth_pair(nb, nb, seed=10 + i) for i in range(min(cores, 16))]

    def np_block(pair):
        params, _ = onp.fit(model, pair[0], np.nan, pair[1], np.nan, (k, k), False, thresh)
        return onp.apply(pair[0], params)

    t0 = time.perf_counter()
    np_block(blocks[0])
    dt_np1 = time.perf_counter() - t0
    variants.append(dict(value=round(nb * nb / dt_np1 / 1e6, 3), cores=1, impl='numpy oracle (oracle/oracle_np.py), 1 thread',
                         sample=f'{nb}x{nb}', seconds=round(dt_np1, 3)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(blocks)) as ex:
        list(ex.map(np_block, blocks))
    dt_npt = time.perf_counter() - t0
    variants.append(dict(value=round(len(blocks) * nb * nb / dt_npt / 1e6, 3), cores=len(blocks),
                         impl=f'numpy oracle under ThreadPoolExecutor({len(blocks)}) over {nb}x{nb} blocks (the shape of homonim/fuse.py:396-408)',
                         sample=f'{len(blocks)} blocks of {nb}x{nb}', seconds=round(dt_npt, 3)))
    if not have_c:
        main = dict(value=variants[0]['value'], cores=1, impl=variants[0]['impl'], sample_px=nb, seconds=variants[0]['seconds'])
    cpu_model = None
    try:
        with open('/proc/cpuinfo') as f:
            cpu_model = next((ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')), None)
    except Exception:
        pass
    return dict(value=main['value'], unit='Mpixels*bands/s', cores=main['cores'], kind='port', cpu_model=cpu_model,
                sample=f"{main['sample_px']}x{main['sample_px']} float32 1-band blocks of the same synthetic workload, {model} "
                       f"{k}x{k} fit+apply, {main['impl']}, {main['seconds']:.3f} s per round",
                variants=variants)


def spot_check(ctx, model, k, thresh, nodata_variant, d_src, d_ref, d_corr, stride, H, W, y0, x0, wh, ww, norm=None, n_fail=0,
               band=0, band_stride=0):
    """ Not timed: download a window (whole rows y0 .. y0 + wh of band plane `band`) and compare the GPU output with the numpy
    oracle.  `norm`: the block statistics the GPU used (gain-blk-offset).  Returns the parity record. """
    from oracle import oracle_np as onp
    r = k // 2
    rows = np.empty((wh, stride), np.float32)
    win = {}
    for name, ptr in (('src', d_src), ('ref', d_ref), ('corr', d_corr)):
        ctx.d2h(rows, ptr + 4 * (band * band_stride + y0 * stride))
        win[name] = rows[:, x0:x0 + ww].copy()
    nodata = np.nan if nodata_variant in (1, 2, 6) else None
    params, _ = onp.fit(model, win['src'], nodata, win['ref'], nodata, (k, k), False, thresh, norm_model=norm)
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
    return dict(window=[int(got.shape[0]), int(got.shape[1])], band=int(band), origin=[int(y0), int(x0)], bitwise_mismatches=n_diff,
                max_rel_diff=rel, nan_pattern_equal=nan_ok, passed=bool(nan_ok and rel <= 1e-5))


def windows_checksum(ctx, windows):
    """ [(device pointer of the window's first pixel, row stride, height, width)] -> (sum of the pixels' bit patterns mod 2^64, pixels) """
    total, px = 0, 0
    for ptr, stride, h, w in windows:
        total = (total + ctx.checksum_dev(ptr, stride, h, w)) & 0xFFFFFFFFFFFFFFFF
        px += h * w
    return total, px


def merge_parity(records):
    """ several spot-check windows -> one record (`passed` = all of them) that keeps the individual windows """
    records = [r for r in records if r is not None]
    if not records:
        return None
    if len(records) == 1:
        return records[0]
    return dict(window=records[0]['window'], bands=[r['band'] for r in records],
                bitwise_mismatches=sum(r['bitwise_mismatches'] for r in records),
                max_rel_diff=max(r['max_rel_diff'] for r in records),
                nan_pattern_equal=all(r['nan_pattern_equal'] for r in records),
                passed=all(r['passed'] for r in records), windows=records)


def probe_copy(ctx, a, b, out, nbytes, reps=12):
    """ Achievable bandwidth of THIS box for the fused kernel's byte mix without its stencil: a flat float4 stream, two reads
    + one write over the run's own planes (hk_stream_probe_dev; shape from tools/ubench_copy.hip), timed with HIP events on
    the launch stream before the timed region.  SURVEY.md 8(d): "builder also measures achievable copy bandwidth on the box
    and reports both fractions".  -> GB/s (median over `reps` launches), GB/s (best) """
    for _ in range(3):
        ctx.stream_probe_dev(a, b, out, nbytes, 0)
    evs = [(ctx.event(), ctx.event()) for _ in range(reps)]
    for e0, e1 in evs:
        ctx.event_record(e0, 0)
        ctx.stream_probe_dev(a, b, out, nbytes, 0)
        ctx.event_record(e1, 0)
    ctx.stream_sync(0)
    ms = sorted(ctx.event_elapsed_ms(e0, e1) for e0, e1 in evs)
    for pair in evs:
        for e in pair:
            ctx.event_destroy(e)
    return 3 * nbytes / (ms[len(ms) // 2] * 1e-3) / 1e9, 3 * nbytes / (ms[0] * 1e-3) / 1e9


def power_probe(run_steps, step_ms, seconds=3.0):
    """ Package power and shader clock while the timed launches repeat back to back for ~`seconds` (OUTSIDE the timed region):
    rocm-smi samples from a side thread.  The fused 5x5 gain-offset kernel runs at the package power cap with the shader clock
    throttled, so its time is energy / cap: this record says whether that is the case on this box. """
    import re
    import shutil
    import subprocess
    import threading
    smi = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    if not os.path.exists(smi):
        return None

    def smi_read(*flags):
        try:
            return subprocess.run([smi, *flags], capture_output=True, text=True, timeout=20).stdout
        except Exception:
            return ''

    cap = re.search(r'Max Graphics Package Power \(W\):\s*([0-9.]+)', smi_read('--showmaxpower'))
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            t = time.perf_counter()
            txt = smi_read('--showpower', '--showclocks')
            w = re.search(r'Package Power \(W\):\s*([0-9.]+)', txt)
            c = re.search(r'sclk clock level:\s*\S+\s*\((\d+)Mhz\)', txt)
            if w and c:
                samples.append((t, float(w.group(1)), float(c.group(1))))

    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = max(8, int(seconds * 1e3 / max(step_ms, 1e-3)))
    run_steps(n)
    t1 = time.perf_counter()
    stop.set()
    th.join(timeout=30)
    # samples taken while the queue was busy (the first ~0.5 s ramps up)
    busy = [(w, c) for (t, w, c) in samples if t0 + 0.6 <= t <= t1 - 0.3]
    if not busy:
        return dict(samples=0)
    ws, cs = sorted(x[0] for x in busy), sorted(x[1] for x in busy)
    return dict(package_watts=ws[len(ws) // 2], sclk_mhz=cs[len(cs) // 2], cap_watts=float(cap.group(1)) if cap else None,
                sclk_max_mhz=2400, samples=len(busy), seconds=round(t1 - t0, 2),
                method='rocm-smi --showpower --showclocks sampled while the timed launches repeat back to back (outside the timed region); medians')


def measured_traffic(args):
    """ HBM bytes per launch from the committed rocprofv3 PMC summary (collected in separate --pmc passes, FETCH_SIZE
    corrected x2 for gfx950) when it was taken at exactly this configuration; None otherwise.  Evidence from
    profiles/, not a measurement of this run (`traffic_source` says so). """
    try:
        with open(os.path.join(REPO, 'profiles', 'pmc_summary.json')) as f:
            pmc = json.load(f)
        c = pmc['config']
        if args.config in (1, 2) and (c['model'], c['kernel'], c['size'], c['bands'], c['nodata'], bool(c.get('no_thresh', False))) == (
                args.model, args.kernel, args.size, args.bands, args.nodata, bool(args.no_thresh)) and not args.params \
                and not c.get('params', False):
            return float(pmc['hbm_traffic_bytes']), 'profiles/pmc_summary.json (rocprofv3 --pmc passes of this command, committed)'
    except Exception:
        pass
    return None, None


# ----------------------------------------------------------------------------------------------------------------------
# configs 1 / 2 (and the sweeps): one raster resident in HBM per rank, one fused launch per step
def run_resident(args, ctx, dist, rank, world):
    from homonim_amd import _hk
    H = W = args.size
    B = args.bands
    k = args.kernel
    stride = (W + 63) // 64 * 64
    band_stride = stride * H
    plane_bytes = 4 * band_stride * B
    thresh = 0.25 if (args.model == 'gain-offset' and not args.no_thresh) else None
    nd = np.nan if args.nodata in (1, 2, 6) else None
    desc = _hk.make_desc(args.model, (k, k), bool(args.params), thresh, nd, nd)

    names = ('src', 'ref', 'corr') + (('gain', 'offset', 'r2') if args.params else ())
    allocs = {name: ctx.dev_alloc(plane_bytes) for name in names}
    bufs = dict(allocs)
    bufs['fail'] = ctx.dev_alloc(8 * B)
    bufs['fail2'] = ctx.dev_alloc(8 * B)
    bufs['norm'] = ctx.dev_alloc(16 * B)
    ctx.memset(bufs['fail'], 0, 8 * B)
    ctx.memset(bufs['fail2'], 0, 8 * B)
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
    fail_dev = [bufs['fail'], bufs['fail2']]
    fail_host = [ctx.pinned_empty((B,), np.uint64) for _ in range(2)]
    fail_ready = [ctx.event(), ctx.event()]
    # ... and, where pixels are EXPECTED to fail the r2 mask (the noisy-reference variants --nodata 3 / 4; real imagery), two scratch
    # buffers (hk_dev_job.scratch): a job that carries scratch gets the in-painting's inputs (offsets + source flags, 5 bytes per
    # pixel of stores) left there by its fit, so the in-painting of step i does not run the fit again for them.  The clean workloads
    # carry none: their fit is the certificate build + list launch and moves the 12 algorithmic bytes.
    scratch = [None, None]
    if thresh is not None and args.nodata in (3, 4):
        job.scratch_bytes = ctx.job_scratch_bytes(job)
        scratch = [ctx.dev_alloc(job.scratch_bytes) for _ in range(2)]
        bufs['scratch0'], bufs['scratch1'] = scratch

    def launch(i):
        job.fail_count = fail_dev[i % 2]
        job.scratch = scratch[i % 2]
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
        job.scratch = scratch[i % 2]
        return ctx.inpaint_dev_counts(desc, job, counts) if ctx.counts_pending(counts) else 0

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

    # what this box's HBM gives the same byte mix as a flat stream (before anything is timed; it scribbles on corr)
    copy_med, copy_best = probe_copy(ctx, bufs['src'], bufs['ref'], bufs['corr'], plane_bytes)

    run(args.warmup)
    ctx.stream_sync(0)

    events = [(ctx.event(), ctx.event()) for _ in range(args.steps)]
    dist.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    n_fail = run(args.steps, events)
    ctx.sync()
    dist.barrier()
    elapsed = time.perf_counter() - t0

    launch_ms = [ctx.event_elapsed_ms(e0, e1) for e0, e1 in events]
    elapsed = dist.max_over_ranks(elapsed)
    n_fail //= max(1, args.steps)
    px_bands = H * W * B
    avg_ms = float(np.mean(launch_ms))

    parity = None
    if rank == 0 and not args.no_parity:
        norm = None
        if args.model == 'gain-blk-offset':
            nh = np.zeros((B, 2), np.float64)
            ctx.d2h(nh, bufs['norm'])
            norm = nh[0]
        wh, ww = min(H, 384), min(W, 1200)
        y0 = max(0, min(H - wh, H // 3))
        x0 = max(0, min(W - ww, (W // 2) // 4 * 4))
        checks = [spot_check(ctx, args.model, k, thresh, args.nodata, bufs['src'], bufs['ref'], bufs['corr'], stride, H, W,
                             y0, x0, wh, ww, norm, n_fail)]
        if B > 1:
            # ... and in the LAST band plane of the launch (plane offsets of several GB), at another place of the raster
            if args.model == 'gain-blk-offset':
                norm = nh[B - 1]
            y1 = max(0, min(H - wh, (2 * H) // 3 + 5))
            x1 = max(0, min(W - ww, (W // 5) // 4 * 4))
            checks.append(spot_check(ctx, args.model, k, thresh, args.nodata, bufs['src'], bufs['ref'], bufs['corr'], stride, H, W,
                                     y1, x1, wh, ww, norm, n_fail, band=B - 1, band_stride=band_stride))
        parity = merge_parity(checks)

    power = None
    if rank == 0 and world == 1 and getattr(args, 'power_probe', False):
        power = power_probe(lambda n: (run(n), ctx.sync()), elapsed / args.steps * 1e3)
    checksum = None
    if getattr(args, 'checksum', False) or world > 1:
        checksum = windows_checksum(ctx, [(bufs['corr'] + 4 * b * band_stride, stride, H, W) for b in range(B)])

    for pair in events + [tuple(fail_ready)]:
        for e in pair:
            ctx.event_destroy(e)
    del fail_host  # page-locked arrays go before the context that allocated them
    for name in names:
        bufs.pop(name)
    for ptr in list(bufs.values()) + list(allocs.values()):
        ctx.dev_free(ptr)

    # an in-painting step is more than its first launch: when pixels failed, the roofline is taken over the whole step
    if n_fail:
        avg_ms = elapsed / args.steps * 1e3
    traffic, traffic_source = measured_traffic(args)
    headline = (args.model, k, H, B, args.nodata, args.no_thresh, args.params) == ('gain-offset', 5, 16384, 4, 0, False, False)
    return dict(
        value=px_bands * args.steps * world / elapsed / 1e6, elapsed=elapsed, scaling='weak',
        workload=f'synthetic float32 {B}-band {H}x{W} src/ref resident in HBM per GPU, Model.{args.model}, '
                 f'kernel {k}x{k}, r2_inpaint_thresh {thresh}, fused fit+apply'
                 + (' + gain / offset / R2 planes written' if args.params else '')
                 + (' (BASELINE.json configs[2])' if headline else '')
                 + (' (BASELINE.json configs[1])' if (args.model, k, H, B, args.nodata, args.params) == ('gain', 5, 8192, 4, 0, False) else ''),
        config=dict(bands=B, height=H, width=W, nodata_variant=args.nodata,
                    parallelism=f'{world} rank(s) x 1 GPU, one raster per rank, no collective',
                    r2_mask_failures_per_step=n_fail),
        roofline=dict(achieved_bytes=ALGO_BYTES_PER_PX * px_bands, avg_launch_ms=avg_ms, traffic=traffic,
                      traffic_source=traffic_source, copy_gbps=copy_med, copy_gbps_best=copy_best,
                      kernel='hk::fit_apply_kernel' + (' + block statistics (hk_norm.hip)' if args.model == 'gain-blk-offset' else '')
                             + (' + in-painting passes (whole step)' if n_fail else '')),
        parity=parity, power=power, checksum=checksum)


# ----------------------------------------------------------------------------------------------------------------------
# config 3: the reference's blocks of a resident raster, processed in place; ranks split the block positions
def run_blocks(args, ctx, dist, rank, world):
    from homonim_amd import _hk, utils
    from homonim_amd.fuse import block_pairs, shard
    H = W = args.size
    B, k = args.bands, args.kernel
    thresh = 0.25 if (args.model == 'gain-offset' and not args.no_thresh) else None
    if thresh is not None:
        raise SystemExit('--config 3 processes blocks in place with a store window: not with r2_inpaint_thresh (use --no-thresh)')
    overlap = utils.overlap_for_kernel((k, k))
    all_blocks = list(block_pairs((H, W), B, overlap, 100))  # reference partition at its default max_block_mem (100 MB)
    # The (band, block) work items of one block POSITION differ only by the band plane they sit in, so the bands of a
    # position go out as one launch (n_bands = B, band_stride = plane) -- per-band statistics and fits as ever.  Ranks split
    # the positions; every rank keeps the whole raster resident (25.8 GB of the 288).
    positions = [bp for bp in all_blocks if bp.band_i == 0]
    mine = shard(positions, rank, world)
    for bp in positions:
        win_in, win_out = bp.src_in_block, bp.src_out_block
        if (win_in.col_off % 4) or ((win_out.col_off - win_in.col_off) % 4):
            raise SystemExit(f'block origin {win_in.col_off} is not 16-byte aligned: kernel {k}x{k} needs a halo that is a multiple of 4')
    stride = (W + 63) // 64 * 64
    band_stride = stride * H
    nd = np.nan if args.nodata in (1, 2, 6) else None
    desc = _hk.make_desc(args.model, (k, k), False, None, nd, nd)
    bufs = {name: ctx.dev_alloc(4 * band_stride * B) for name in ('src', 'ref', 'corr')}
    bufs['norm'] = ctx.dev_alloc(16 * B * max(1, len(mine)))
    ctx.synth_fill_dev(bufs['src'], bufs['ref'], B, H, W, stride, band_stride, seed=1234, nodata_variant=args.nodata, stream=0)
    ctx.stream_sync(0)
    copy_med, copy_best = probe_copy(ctx, bufs['src'], bufs['ref'], bufs['corr'], 4 * band_stride * B)
    n_streams = ctx.n_streams

    bpj = B   # the bands of a block position share a launch (fewer bands per job: no gain, also not for a rank's small shard -- profiles/r06_c3_rank.txt)

    def make_step(my_positions, norm_buf):
        """ the jobs of a rank that holds `my_positions` and the function that queues one step of them -> (step, jobs, batches) """
        jobs = []
        for i, bp in enumerate(my_positions):
            win_in, win_out = bp.src_in_block, bp.src_out_block
            for b0 in range(0, B, bpj):
                off = 4 * (b0 * band_stride + win_in.row_off * stride + win_in.col_off)
                job = _hk.DevJob()
                job.src, job.ref, job.corr = bufs['src'] + off, bufs['ref'] + off, bufs['corr'] + off
                job.gain = job.offset = job.r2 = job.fail_count = None
                job.norm = norm_buf + 16 * (B * i + b0)
                job.n_bands, job.height, job.width, job.stride, job.band_stride = min(bpj, B - b0), win_in.height, win_in.width, stride, band_stride
                job.seg_rows, job.stream = args.seg_rows, len(jobs) % n_streams   # the latency-bound statistics of one position overlap another's fit
                job.out_row0, job.out_col0 = win_out.row_off - win_in.row_off, win_out.col_off - win_in.col_off
                job.out_rows, job.out_cols = win_out.height, win_out.width
                jobs.append(job)
        # Batched launches (hk_block_norm_batch_dev / hk_fit_apply_batch_dev): the rank's jobs in `n_batches` groups, each group one
        # launch per kernel stage on its own stream (0 = one launch per job, spread over the streams)
        n_batches = min(int(os.environ.get('HK_BENCH_BATCHES', str(args.batches))), len(jobs))
        if len(jobs) <= 4 and 'HK_BENCH_BATCHES' not in os.environ:
            # a rank of a wide launch holds a few block positions only: one launch per position on a stream each, so that one position's
            # latency-bound statistics chain runs beside the other's fit (the shard of rank 3 of 8: 1.60 ms against 1.70 as two batches)
            n_batches = 0
        batches = []
        if n_batches > 0:
            per = (len(jobs) + n_batches - 1) // n_batches
            for g in range(n_batches):
                group = jobs[g * per:(g + 1) * per]
                for job in group:
                    job.stream = g % n_streams
                if group:
                    batches.append((ctx.job_array(group), group[0].norm))

        def step():
            if batches:
                for arr, norm0 in batches:
                    if args.model == 'gain-blk-offset':
                        ctx.block_norm_batch_dev(desc, arr, norm0)
                    ctx.fit_apply_batch_dev(desc, arr)
                return
            for job in jobs:
                if args.model == 'gain-blk-offset':
                    ctx.block_norm_dev(desc, job, job.norm)
                ctx.fit_apply_dev(desc, job)
        return step, jobs, batches

    step, jobs, batches = make_step(mine, bufs['norm'])

    for _ in range(args.warmup):
        step()
    ctx.sync()
    dist.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    dist.barrier()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0)
    my_px = B * sum(bp.src_out_block.height * bp.src_out_block.width for bp in mine)

    parity = None
    if rank == 0 and not args.no_parity and jobs:
        # the LAST block position of the rank (a batched launch that lost its job table would never get there)
        bp = mine[-1]
        nm = np.zeros((B, 2))
        ctx.d2h(nm, bufs['norm'] + 16 * B * (len(mine) - 1))
        wi = bp.src_in_block
        # a block's statistics are taken over its whole in-block; check a window of band 0's first out-block against the oracle
        y0, x0 = wi.row_off + 1000, (wi.col_off + 1200) // 4 * 4
        parity = spot_check(ctx, args.model, k, None, args.nodata, bufs['src'], bufs['ref'], bufs['corr'], stride, H, W, y0, x0,
                            256, 1000, nm[0], 0)

    checksum = None
    if getattr(args, 'checksum', False) or world > 1:   # the out-blocks this rank wrote, every band
        wins = []
        for bp in mine:
            wo = bp.src_out_block
            wins += [(bufs['corr'] + 4 * (b * band_stride + wo.row_off * stride + wo.col_off), stride, wo.height, wo.width) for b in range(B)]
        checksum = windows_checksum(ctx, wins)

    projection = None
    if world == 1 and not args.no_projection and args.as_rank is None:
        # what rank r of an n-rank launch would run, alone on this GPU (tools: bench.py --config 3 --as-rank R/N)
        pnorm = ctx.dev_alloc(16 * B * len(positions))
        psteps = max(2, min(args.steps, 4))

        def shard_ms(r, n):
            sub = shard(positions, r, n)
            if not sub:
                return 0.0
            st, _, _ = make_step(sub, pnorm)
            return timed_steps(ctx, st, psteps, 1) * 1e3
        projection = project_scaling(elapsed / args.steps * 1e3, shard_ms)
        ctx.dev_free(pnorm)

    e2e = None
    if not args.no_end_to_end:
        my_bands = shard(list(range(B)), rank, world, contiguous=True)  # host rasters of this rank's bands only
        e2e = end_to_end_blocks(args, ctx, dist, bufs, len(my_bands), my_bands[0] if my_bands else 0, stride, band_stride)
    for ptr in bufs.values():
        ctx.dev_free(ptr)
    total_px = H * W * B
    return dict(
        value=total_px * args.steps / elapsed / 1e6, elapsed=elapsed, scaling='strong',
        workload=f'synthetic float32 {B}-band {H}x{W} resident in HBM, Model.{args.model}, kernel {k}x{k}, the reference\'s '
                 f'{len(all_blocks)} blocks (4096x4096 + {overlap[0]}-px halo, raster_pair.py:342-428) processed in place, '
                 f'each with its own block statistics; '
                 + (f'the {len(jobs)} block positions of the rank in {len(batches)} batched launches per kernel stage' if batches else f'the {B} bands of a block position share a launch')
                 + ' (BASELINE.json configs[3])',
        config=dict(bands=B, height=H, width=W, nodata_variant=args.nodata, blocks=len(all_blocks),
                    blocks_per_rank=B * len(mine),
                    parallelism=f'{world} rank(s) x 1 GPU, the {len(positions)} block positions dealt round-robin to the ranks, no collective'),
        roofline=dict(achieved_bytes=ALGO_BYTES_PER_PX * my_px, avg_launch_ms=elapsed / args.steps * 1e3, traffic=None, traffic_source=None,
                      copy_gbps=copy_med, copy_gbps_best=copy_best,
                      kernel=f'one step of this rank: {len(mine)} x (block statistics + hk::fit_apply_kernel over {B} bands), wall time on {n_streams} streams'),
        parity=parity, end_to_end=e2e, projection=projection, checksum=checksum)


def end_to_end_blocks(args, ctx, dist, bufs, nb, b0, stride, band_stride):
    """ The same blocks through RasterFuse.process from page-locked host rasters (one pass, PCIe-inclusive); the ranks
    split the bands. """
    import warnings
    from homonim_amd.fuse import RasterFuse
    H = W = args.size
    try:
        src = ctx.pinned_empty((nb, H, stride), np.float32)
        ref = ctx.pinned_empty((nb, H, stride), np.float32)
        out = ctx.pinned_empty((nb, H, W), np.float32)
    except Exception as ex:  # not enough lockable host memory on this box
        return dict(skipped=f'page-locked host rasters could not be allocated: {ex}')
    if nb == 0:  # more ranks than bands: keep the collective calls of the other ranks matched
        dist.barrier(), dist.barrier()
        dist.max_over_ranks(0.0)
        return dict(skipped='no band for this rank')
    ctx.d2h(src, bufs['src'] + 4 * band_stride * b0)
    ctx.d2h(ref, bufs['ref'] + 4 * band_stride * b0)
    nd = np.nan if args.nodata in (1, 2, 6) else None
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        rf = RasterFuse(src[:, :, :W], ref[:, :, :W], src_nodata=nd, ref_nodata=nd)
        kw = dict(model=args.model, kernel_shape=(args.kernel, args.kernel), model_config=dict(r2_inpaint_thresh=None),
                  block_config=dict(threads=4, max_block_mem=100), device_config=dict(devices=[ctx.device], streams=ctx.n_streams, pin=False),
                  corr_out=out)
        rf.process(**kw)  # warm-up: grows the per-stream device slabs
        dist.barrier()
        t0 = time.perf_counter()
        rf.process(**kw)
        dist.barrier()
        dt = dist.max_over_ranks(time.perf_counter() - t0)
    px = args.size * args.size * args.bands
    res = dict(value=round(px / dt / 1e6, 1), unit='Mpixels*bands/s', seconds=round(dt, 4),
               path='RasterFuse.process: page-locked host rasters -> H2D || statistics + fused kernel || D2H per block, 4 threads x 4 streams per GPU',
               pcie_gbps_in=round(8 * px / dt / 1e9, 1), pcie_gbps_out=round(4 * px / dt / 1e9, 1))
    del src, ref, out
    return res


# ----------------------------------------------------------------------------------------------------------------------
# config 4: mosaic of independent tiles, one tile per stream; ranks split the tile list
def run_tiles(args, ctx, dist, rank, world):
    from homonim_amd import _hk
    from homonim_amd.fuse import shard
    n = args.size
    B, k, T = args.bands, args.kernel, args.tiles
    thresh = 0.25 if (args.model == 'gain-offset' and not args.no_thresh) else None
    nd = np.nan if args.nodata in (1, 2, 6) else None
    desc = _hk.make_desc(args.model, (k, k), False, thresh, nd, nd)
    mine = shard(list(range(T)), rank, world, contiguous=True)
    stride = (n + 63) // 64 * 64
    band_stride = stride * n
    tile_bytes = 4 * band_stride * B
    n_streams = ctx.n_streams
    tiles = []
    # the counters and statistics of all tiles in one array each: a batched launch fetches / fills them with one copy
    fail_all = ctx.dev_alloc(8 * B * max(1, len(mine)))
    norm_all = ctx.dev_alloc(16 * B * max(1, len(mine)))
    ctx.memset(fail_all, 0, 8 * B * max(1, len(mine)))
    for j, t in enumerate(mine):
        d = {name: ctx.dev_alloc(tile_bytes) for name in ('src', 'ref', 'corr')}
        d['fail'] = fail_all + 8 * B * j
        d['norm'] = norm_all + 16 * B * j
        ctx.synth_fill_dev(d['src'], d['ref'], B, n, n, stride, band_stride, seed=5000 + t, nodata_variant=args.nodata, stream=0)
        job = _hk.DevJob()
        job.src, job.ref, job.corr = d['src'], d['ref'], d['corr']
        job.gain = job.offset = job.r2 = None
        job.fail_count = d['fail']
        job.norm = d['norm'] if args.model == 'gain-blk-offset' else None
        job.n_bands, job.height, job.width, job.stride, job.band_stride = B, n, n, stride, band_stride
        # launches of different streams overlap each other's tails, so a tile prefers longer wave segments than a launch that has
        # the GPU to itself (half the priming rows, half the waves): 128 rows 12.4 ms per step against 13.4 at the library's 64
        # (profiles/r03_c4_segrows.txt; alone, a tile's launch takes 0.300 ms at 64 rows and 0.335 at 128)
        job.seg_rows, job.stream = (args.seg_rows or (128 if k <= 5 else 0)), j % n_streams
        if thresh is not None and args.nodata in (3, 4):  # failures expected: the in-painting's inputs stay with the tile until its counters have been looked at
            job.scratch_bytes = ctx.job_scratch_bytes(job)
            d['scratch'] = ctx.dev_alloc(job.scratch_bytes)
            job.scratch = d['scratch']
        tiles.append((d, job, ctx.pinned_empty((B,), np.uint64), ctx.event()))
    ctx.stream_sync(0)
    copy_med = copy_best = None
    if tiles:
        d0 = tiles[0][0]
        copy_med, copy_best = probe_copy(ctx, d0['src'], d0['ref'], d0['corr'], tile_bytes)

    def make_step(my_tiles):
        """ the function that queues one step of a rank holding `my_tiles` (streams dealt within the rank) -> (step, batches) """
        for j, (d, job, counts, ev) in enumerate(my_tiles):
            job.stream = j % n_streams
        # Batched launches: the rank's tiles in `n_batches` groups, each one launch per kernel stage on its own stream
        n_batches = min(int(os.environ.get('HK_BENCH_BATCHES', str(args.batches))), len(my_tiles))
        batches = []
        if n_batches > 0:
            per = (len(my_tiles) + n_batches - 1) // n_batches
            for g in range(n_batches):
                group = my_tiles[g * per:(g + 1) * per]
                for d, job, counts, ev in group:
                    job.stream = g % n_streams
                if group:
                    batches.append((ctx.job_array([t[1] for t in group]), group, ctx.pinned_empty((B * len(group),), np.uint64), ctx.event()))

        def step_batched():
            n_fail = 0
            for arr, group, counts_all, ev in batches:
                if args.model == 'gain-blk-offset':
                    ctx.block_norm_batch_dev(desc, arr, group[0][0]['norm'])
                ctx.fit_apply_batch_dev(desc, arr)
                if thresh is not None:
                    ctx.fail_counts_batch_async(arr, counts_all, ev)
            if thresh is not None:
                for arr, group, counts_all, ev in batches:
                    ctx.event_sync(ev)
                    c_all = counts_all.copy()
                    if ctx.counts_pending(c_all):
                        for i, (d, job, counts, _) in enumerate(group):
                            c = c_all[B * i:B * (i + 1)]
                            if ctx.counts_pending(c):
                                n_fail += ctx.inpaint_dev_counts(desc, job, c)
            return n_fail

        def step():
            """ every tile's fused launch on its stream, then the host's look at the r2-mask counters of all of them """
            if batches:
                return step_batched()
            n_fail = 0
            for d, job, counts, ev in my_tiles:
                if args.model == 'gain-blk-offset':
                    ctx.block_norm_dev(desc, job, d['norm'])
                ctx.fit_apply_dev(desc, job)
                if thresh is not None:
                    ctx.fail_counts_async(job, counts, ev)
            if thresh is not None:
                for d, job, counts, ev in my_tiles:
                    ctx.event_sync(ev)
                    c = counts.copy()
                    if ctx.counts_pending(c):
                        n_fail += ctx.inpaint_dev_counts(desc, job, c)
            return n_fail
        return step, batches

    step, batches = make_step(tiles)

    for _ in range(args.warmup):
        step()
    ctx.sync()
    dist.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    n_fail = 0
    for _ in range(args.steps):
        n_fail += step()
    ctx.sync()
    dist.barrier()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0)

    parity = None
    if rank == 0 and not args.no_parity and tiles:
        d = tiles[0][0]
        parity = spot_check(ctx, args.model, k, thresh, args.nodata, d['src'], d['ref'], d['corr'], stride, n, n, n // 3, 1024,
                            min(n, 384), min(n - 1024, 1200), None, n_fail)
    checksum = None
    if getattr(args, 'checksum', False) or world > 1:   # this rank's tiles, every band
        checksum = windows_checksum(ctx, [(d['corr'] + 4 * b * band_stride, stride, n, n) for d, job, counts, ev in tiles for b in range(B)])

    projection = None
    if world == 1 and not args.no_projection and args.as_rank is None and tiles:
        # what rank r of an n-rank launch would run, alone on this GPU (tools: bench.py --config 4 --as-rank R/N)
        psteps = max(2, min(args.steps, 4))

        def shard_ms(r, n):
            sub = [tiles[j] for j in shard(list(range(T)), r, n, contiguous=True)]
            if not sub:
                return 0.0
            st, sub_batches = make_step(sub)
            ms = timed_steps(ctx, st, psteps, 1) * 1e3
            for arr, group, counts_all, ev in sub_batches:
                ctx.event_destroy(ev)
            return ms
        projection = project_scaling(elapsed / args.steps * 1e3, shard_ms)
        for j, (d, job, counts, ev) in enumerate(tiles):
            job.stream = j % n_streams

    e2e = None
    if not args.no_end_to_end:
        e2e = end_to_end_tiles(args, ctx, dist, tiles, stride, band_stride, thresh, nd)
    for d, job, counts, ev in tiles:
        ctx.event_destroy(ev)
        for name, ptr in d.items():
            if name not in ('fail', 'norm'):  # slices of fail_all / norm_all
                ctx.dev_free(ptr)
    for arr, group, counts_all, ev in batches:
        ctx.event_destroy(ev)
    ctx.dev_free(fail_all), ctx.dev_free(norm_all)
    my_px = len(mine) * n * n * B
    del tiles
    return dict(
        value=T * n * n * B * args.steps / elapsed / 1e6, elapsed=elapsed, scaling='strong',
        workload=f'mosaic of {T} independent {B}-band {n}x{n} float32 tiles resident in HBM, Model.{args.model}, kernel {k}x{k}, '
                 f'r2_inpaint_thresh {thresh}, one fused launch per tile, tiles dealt round the streams (BASELINE.json configs[4])',
        config=dict(bands=B, height=n, width=n, tiles=T, tiles_per_rank=len(mine), nodata_variant=args.nodata,
                    parallelism=f'{world} rank(s) x 1 GPU x {n_streams} streams, consecutive runs of the tile list per rank, no collective',
                    r2_mask_failures_per_step=n_fail // max(1, args.steps)),
        roofline=dict(achieved_bytes=ALGO_BYTES_PER_PX * my_px, avg_launch_ms=elapsed / args.steps * 1e3, traffic=None, traffic_source=None,
                      copy_gbps=copy_med, copy_gbps_best=copy_best,
                      kernel=f'one step of this rank: {len(mine)} x hk::fit_apply_kernel on {n_streams} streams, wall time'),
        parity=parity, end_to_end=e2e, projection=projection, checksum=checksum)


def end_to_end_tiles(args, ctx, dist, tiles, stride, band_stride, thresh, nd):
    """ The rank's tiles as host rasters through RasterFuse.process (one RasterFuse per tile, as one would process a
    directory of tiles): H2D || kernel || D2H over the context's streams; PCIe-inclusive. """
    import warnings
    from concurrent.futures import ThreadPoolExecutor
    from homonim_amd.fuse import RasterFuse
    n, B = args.size, args.bands
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 64 << 30
    per_tile = 3 * 4 * B * n * stride
    distinct = max(2, min(len(tiles), int(0.3 * avail // per_tile), getattr(args, 'e2e_distinct', None) or len(tiles)))
    host = []
    try:
        for j in range(distinct):
            s, r, o = (ctx.pinned_empty((B, n, stride), np.float32) for _ in range(3))
            ctx.d2h(s, tiles[j][0]['src'])
            ctx.d2h(r, tiles[j][0]['ref'])
            host.append((s, r, o))
    except Exception as ex:
        if len(host) < 2:
            return dict(skipped=f'page-locked host tiles could not be allocated: {ex}')
    distinct = len(host)
    # Four tiles in flight x two block threads each on a context of eight streams (round 5; two tiles on four streams left the link
    # idle at every tile's end: 5.2 -> 6.3 Gpixel*bands/s = 41.6 -> 50.3 GB/s host-to-device, same box).  The resident part of the
    # configuration keeps its four streams; this pass has a context of its own.
    in_flight = int(os.environ.get('HK_BENCH_E2E_TILES', '4'))
    blk_threads = int(os.environ.get('HK_BENCH_E2E_THREADS', '2'))
    n_e2e_streams = int(os.environ.get('HK_BENCH_E2E_STREAMS', '8'))
    kw = dict(model=args.model, kernel_shape=(args.kernel, args.kernel), model_config=dict(r2_inpaint_thresh=thresh),
              block_config=dict(threads=blk_threads, max_block_mem=100),
              device_config=dict(devices=[ctx.device], streams=n_e2e_streams, pin=False))

    def one(j):
        s, r, o = host[j % distinct]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            RasterFuse(s[:, :, :n], r[:, :, :n], src_nodata=nd, ref_nodata=nd).process(corr_out=o[:, :, :n] if stride == n else None, **kw)

    with ThreadPoolExecutor(in_flight) as ex:
        list(ex.map(one, range(min(len(tiles), 4))))  # warm-up: grows the per-stream device slabs
        dist.barrier()
        t0 = time.perf_counter()
        list(ex.map(one, range(len(tiles))))
        dist.barrier()
        dt = dist.max_over_ranks(time.perf_counter() - t0)
    px = args.tiles * n * n * B
    res = dict(value=round(px / dt / 1e6, 1), unit='Mpixels*bands/s', seconds=round(dt, 4), distinct_host_tiles=distinct,
               path=f'RasterFuse.process per tile: page-locked host rasters -> H2D || fused kernel || D2H, {in_flight} tiles x {blk_threads} block threads on {n_e2e_streams} streams per GPU',
               pcie_gbps_in=round(8 * px / dt / 1e9, 1), pcie_gbps_out=round(4 * px / dt / 1e9, 1))
    del host
    return res


# ----------------------------------------------------------------------------------------------------------------------
def launch_ranks(n: int) -> int:
    """ `python bench.py --gpus N` with no launcher around it: this process -- which has not touched a GPU and never will --
    starts the N ranks as CHILD processes (one per GPU, the environment torch.distributed.run would give them: RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_*), lets rank 0 print the one JSON line on the shared stdout and returns the worst
    exit code.  The shape of homonim/fuse.py:394-408: one call fans the work out over a pool of workers. """
    import socket
    import subprocess
    probe_failed = False
    try:   # HK_FIRST_PROCESS_PROBE=1: a child is the lease's first GPU process (see main); this launcher itself makes no GPU call
        from harness import first_process
        probe = first_process.gate()
        probe_failed = bool(probe and probe['fatal'])
    except Exception as ex:
        sys.stderr.write(f'bench.py: first-process probe not run: {ex}\n')
    port = os.environ.get('MASTER_PORT')
    if port is None:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = str(sk.getsockname()[1])
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 128 - rc)
                sys.stderr.write(f'bench.py: rank {r} exited with {rc}; stopping the other ranks\n')
                for q in pending:   # a rank that lost its peers would wait in a collective for ever
                    procs[q].terminate()
        time.sleep(0.05)
    return worst or (3 if probe_failed else 0)   # a process of this run died: the run fails, whatever the ranks printed


def main():
    args = parse_args()
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:  # plain `python bench.py --gpus N`: be the launcher (no GPU call in this process)
        sys.exit(launch_ranks(args.gpus))
    if int(env_world or '1') != args.gpus:  # a launcher started a different number of ranks than the command line names
        sys.stderr.write(
            f'bench.py: --gpus {args.gpus} but WORLD_SIZE is {env_world}: launch the ranks with\n'
            f'  python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 '
            f'--master-port 29500 bench.py --gpus {args.gpus} ...\n(or unset WORLD_SIZE: `python bench.py --gpus N` starts its own ranks)\n')
        sys.exit(2)
    try:   # a fatal signal names its sender, thread and native frames (harness/abort_trace.py)
        from harness import abort_trace
        abort_trace.install()
    except Exception:
        pass
    # HK_FIRST_PROCESS_PROBE=1 (off by default since round 5): a child is the lease's first GPU process (harness/first_process.py;
    # round 3's aborts only ever hit first processes).  If it dies, this run prints its line and then exits 3: a green run
    # means no process died.  (Ranks of a launch: the launcher did it.)
    first_probe = None
    if env_world is None:
        try:
            from harness import first_process
            first_probe = first_process.gate()
        except Exception as ex:
            sys.stderr.write(f'bench.py: first-process probe not run: {ex}\n')
    from homonim_amd import _hk, dist, topology
    rank, world, local_rank = dist.init()  # the ranks meet over loopback TCP, only when WORLD_SIZE > 1 (homonim_amd/dist.py)
    # Host placement: this rank's threads onto the cores of its GPU's NUMA node, BEFORE anything page-locked is allocated
    # (the staging rings of the context, the host rasters of `end_to_end`): SURVEY.md 8(e) "NUMA-local pinned buffers"
    affinity_before = os.sched_getaffinity(0)   # (the CPU baseline below gets the whole host back)
    placement = topology.bind_to_device(local_rank % max(1, _hk.device_count()))
    # one GPU per rank on a full node.  configs[3] deals its block positions to 8 streams: a position's statistics are a chain of
    # small latency-bound kernels, and with 4 streams the GPU still idled between them (12.1 -> 11.5 ms per step)
    n_streams = int(os.environ.get('HK_BENCH_STREAMS', '8' if args.config == 3 else '4'))
    ctx = _hk.Context(local_rank % max(1, _hk.device_count()), n_streams=n_streams)
    ctx.selftest()
    # The library's own RCCL communicator over the ranks of this launch (the one data-path collective of the hot path, the
    # split-block statistics, runs on it): joined here so that every N > 1 run also proves RCCL over xGMI up -- one in-place
    # all-reduce of a float64 word per rank, which must come back as the number of ranks.
    rccl_ranks, rccl_error, rccl_join_abandoned = None, None, False
    if dist.backend() == 'rccl':
        # Reported, never fatal, and never allowed to hang the run: the timed path has no collective, a rank's shard does not depend on
        # this communicator.  ncclCommInitRank returns only when EVERY rank has joined and has no timeout of its own, so a rank whose
        # librccl is missing (or whose bootstrap fails) would leave its peers waiting for ever: the join runs on a side thread with a
        # deadline, and the ranks agree over the launch's sockets (dist.sum_over_ranks) whether everybody got in before anybody queues
        # the all-reduce.
        # The id travels over the launch's sockets on THIS thread (they carry one conversation at a time); only the call that can
        # block without a deadline runs beside it.  A join that missed its deadline may still be inside ncclCommInitRank on `ctx`:
        # such a run reports, skips the collectives and leaves the context open at exit instead of tearing it down under the thread.
        import threading
        joined = {}
        try:
            uid = dist.exchange_comm_id()
        except Exception as ex:
            uid, joined['err'] = None, f'{type(ex).__name__}: {ex}'

        def join():
            try:
                ctx.comm_init(uid, rank, world)
                joined['ok'] = True
            except Exception as ex:
                joined['err'] = f'{type(ex).__name__}: {ex}'
        if uid is not None:
            th = threading.Thread(target=join, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get('HK_BENCH_RCCL_TIMEOUT', '120')))
            rccl_join_abandoned = th.is_alive()
        mine_ok = bool(joined.get('ok'))
        if not mine_ok:
            rccl_error = joined.get('err') or 'joining the communicator did not return within the deadline (a peer never joined?)'
        all_ok = dist.sum_over_ranks(1.0 if mine_ok else 0.0) == float(world)
        if all_ok:
            try:
                word = ctx.dev_alloc(8)
                ctx.h2d(word, np.ones(1, np.float64))
                ctx.comm_allreduce_f64_dev(word, 1, 0)
                ctx.stream_sync(0)
                back = np.zeros(1, np.float64)
                ctx.d2h(back, word)
                ctx.dev_free(word)
                rccl_ranks = ctx.comm_info()[1]
                if int(back[0]) != world or rccl_ranks != world:
                    rccl_error = f'all-reduce over {world} rank(s) returned {back[0]} (communicator of {rccl_ranks})'
            except Exception as ex:
                rccl_error = f'{type(ex).__name__}: {ex}'
        elif rccl_error is None:
            rccl_error = 'another rank could not join the communicator'
        if rccl_error:
            sys.stderr.write(f'bench.py: rank {rank}: the library\'s RCCL communicator is not usable: {rccl_error}\n')

    # every rank's host placement travels to rank 0's line (a rank bound to the wrong socket shows there, not in a lost stderr)
    placements = None
    if world > 1:
        placements = [json.loads(b.decode()) for b in dist.gather_bytes(json.dumps(topology.summary(placement)).encode())]
    # Fault injection for the launch tests (tests/test_gpu_multirank.py): HK_BENCH_DIE_RANK=R ends rank R abruptly once the ranks
    # have met -- the launch must end with a non-zero exit code within seconds, not sit in a barrier until a timeout.
    if os.environ.get('HK_BENCH_DIE_RANK') == str(rank) and world > 1:
        sys.stderr.write(f'bench.py: rank {rank}: HK_BENCH_DIE_RANK -- exiting abruptly\n')
        sys.stderr.flush()
        os._exit(17)

    runner = {1: run_resident, 2: run_resident, 3: run_blocks, 4: run_tiles}[args.config]
    if args.as_rank is not None:   # the shard of rank R of N, alone on this GPU: no group, no peers
        res = runner(args, ctx, SoloDist, args.as_rank[0], args.as_rank[1])
        my_px = res['roofline']['achieved_bytes'] / ALGO_BYTES_PER_PX
        res['value'] = my_px * args.steps / res['elapsed'] / 1e6   # what THIS GPU did, not a whole-job rate
        res['scaling'] = f'none: rank {args.as_rank[0]} of {args.as_rank[1]} run alone (projection aid)'
        res['config']['as_rank'] = dict(rank=args.as_rank[0], of=args.as_rank[1], shard_ms=round(res['elapsed'] / args.steps * 1e3, 4),
                                        note='one rank\'s shard of a strong-scaling launch on one GPU: not a scaling measurement')
    else:
        res = runner(args, ctx, dist, rank, world)
    # Through RasterFuse every raster has nodata = nan (raster_array.py:172-188), i.e. the product path runs the GENERAL
    # kernels; `value` is quoted on BASELINE.json's plain synthetic rasters.  The default run therefore adds a second,
    # shorter measurement of the same configuration on rasters with a NaN frame (reported beside, never as `value`).
    nan_variant = blob_variant = None
    if args.config == 2 and args.nodata == 0 and world == 1 and not args.no_nan_variant and not args.params:
        import copy

        def variant(nd, workload):
            a2 = copy.copy(args)
            a2.nodata, a2.steps, a2.warmup, a2.power_probe = nd, min(args.steps, 20), 3, False
            r2_ = run_resident(a2, ctx, dist, rank, world)
            rl2 = r2_['roofline']
            return {
                'workload': workload,
                'value': round(r2_['value'], 1), 'steps': a2.steps, 'ms_per_step': round(r2_['elapsed'] / a2.steps * 1e3, 4),
                'avg_launch_ms': round(rl2['avg_launch_ms'], 4),
                'frac': round(rl2['achieved_bytes'] / (rl2['avg_launch_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                'frac_of_copy': round(rl2['achieved_bytes'] / (rl2['avg_launch_ms'] * 1e-3) / 1e9 / rl2['copy_gbps'], 4) if rl2.get('copy_gbps') else None,
                'parity_spot_check': r2_['parity'],
            }
        nan_variant = variant(2, 'the same rasters with a 3-pixel NaN frame, src / ref nodata = nan (general kernels: the RasterFuse path)')
        # ... and what real nodata costs: the frame + ~1 % of the area in round holes 32 - 128 pixels across, source and reference
        # independently (cloud / shadow masks; raster_array.py:298-308, utils.py:54-56)
        blob_variant = variant(6, 'the same rasters with a 3-pixel NaN frame and ~1 % of the area in round NaN holes 32 - 128 px across '
                                  '(src and ref independently), src / ref nodata = nan')

    # The in-painting branch of the gain-offset model (kernel_model.py:361-371; r2_inpaint_thresh = 0.25 is the model's default)
    # on rasters whose reference is noisy: 35 % / 94 % of the pixels fail the r2 mask (--nodata 3 / 4).  Reported beside, never as `value`.
    inpaint_variants = None
    if args.config == 2 and args.nodata == 0 and world == 1 and not args.no_nan_variant and not args.params and not args.no_thresh \
            and (args.model, args.kernel, args.size, args.bands) == ('gain-offset', 5, 16384, 4):
        import copy
        inpaint_variants = {}
        for nd, steps in ((3, 6), (4, 4)):
            a4 = copy.copy(args)
            a4.nodata, a4.steps, a4.warmup, a4.power_probe = nd, steps, 2, False
            r4 = run_resident(a4, ctx, dist, rank, world)
            inpaint_variants['--nodata %d' % nd] = {
                'workload': 'the same rasters with a noisy reference: %s of the pixels fail the r2 mask and are in-painted' % ('35 %' if nd == 3 else '94 %'),
                'r2_mask_failures_per_step': r4['config'].get('r2_mask_failures_per_step'),
                'value': round(r4['value'], 1), 'steps': steps, 'ms_per_step': round(r4['elapsed'] / steps * 1e3, 4),
                'parity_spot_check': r4['parity'],
            }

    # BASELINE.json's other configurations, driver-timed beside the headline: a few steps each of configs[1], [3], [4] with
    # their own parity spot checks -- compact records, never `value` (their full lines: --config N)
    other = None
    if args.config == 2 and world == 1 and not args.no_other_configs and not args.params and args.nodata == 0 \
            and (args.model, args.kernel, args.size, args.bands) == ('gain-offset', 5, 16384, 4):
        import copy
        other = {}
        for cfg, (steps, warm) in {1: (40, 3), 3: (6, 1), 4: (4, 1)}.items():
            a3 = copy.copy(args)
            a3.config = cfg
            for key in ('model', 'kernel', 'size', 'bands', 'tiles'):
                setattr(a3, key, CONFIGS[cfg].get(key))
            a3.batches = CONFIGS[cfg].get('batches', 0)
            a3.steps, a3.warmup, a3.no_end_to_end, a3.no_thresh, a3.power_probe, a3.seg_rows = steps, warm, True, False, False, 0
            if cfg == 4:   # the pinned-host streaming pipeline (H2D || kernel || D2H through RasterFuse), driver-timed: 8 distinct host tiles
                a3.no_end_to_end, a3.e2e_distinct = False, 8
            c3 = ctx
            if cfg == 3:   # configs[3] deals its block positions to eight streams (see below)
                c3 = _hk.Context(ctx.device, n_streams=int(os.environ.get('HK_BENCH_STREAMS', '8')))
            try:
                r3 = {1: run_resident, 3: run_blocks, 4: run_tiles}[cfg](a3, c3, dist, rank, world)
            finally:
                if c3 is not ctx:
                    c3.close()
            rl3 = r3['roofline']
            ach3 = rl3['achieved_bytes'] / (rl3['avg_launch_ms'] * 1e-3) / 1e9
            other[str(cfg)] = {
                'workload': r3['workload'], 'value': round(r3['value'], 1), 'steps': steps, 'scaling': r3['scaling'],
                'ms_per_step': round(r3['elapsed'] / steps * 1e3, 4), 'avg_launch_ms': round(rl3['avg_launch_ms'], 4),
                'frac': round(ach3 / HBM_PEAK_GBPS, 4),
                'frac_of_copy': round(ach3 / rl3['copy_gbps'], 4) if rl3.get('copy_gbps') else None,
                'parity_passed': None if r3['parity'] is None else bool(r3['parity']['passed']),
                'parity_bitwise_mismatches': None if r3['parity'] is None else r3['parity']['bitwise_mismatches'],
            }
            if r3.get('end_to_end') is not None:   # PCIe-inclusive, host rasters: never `value`
                other[str(cfg)]['end_to_end'] = r3['end_to_end']
            if r3.get('projection') is not None:   # every rank's shard of an N-rank launch alone on this GPU: a projection
                other[str(cfg)]['projected_scaling_single_gpu'] = r3['projection']

    # the corrected pixels of the whole launch as an exact checksum: every rank's (sum, pixels), gathered in rank order
    shard_checksum = None
    if res.get('checksum') is not None:
        mine_ck = json.dumps(list(res['checksum'])).encode()
        parts = [json.loads(b.decode()) for b in (dist.gather_bytes(mine_ck) if args.as_rank is None else [mine_ck])]
        shard_checksum = dict(sum=sum(p[0] for p in parts) & 0xFFFFFFFFFFFFFFFF, pixels=sum(p[1] for p in parts),
                              per_rank=[dict(sum=p[0], pixels=p[1]) for p in parts],
                              what='sum of the float32 bit patterns of every corrected pixel the launch produced, modulo 2^64 (hk_debug_checksum_dev)')

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            # the baseline is the HOST's: its worker threads (created from this thread) may use every CPU the launcher allowed, not
            # only the GPU's NUMA node this rank was bound to for its staging work; the GPU part of the run is over
            os.sched_setaffinity(0, affinity_before)
            cpu = cpu_baseline(args.model, args.kernel, args.cpu_sample)
        rl = res['roofline']
        achieved = rl['achieved_bytes'] / (rl['avg_launch_ms'] * 1e-3) / 1e9
        k = args.kernel
        out = {
            'metric': 'Mpixels*bands/s fit+apply (5x5 gain-offset, float32)' if (args.model == 'gain-offset' and k == 5)
                      else f'Mpixels*bands/s fit+apply ({k}x{k} {args.model}, float32)',
            'value': round(res['value'], 1), 'unit': 'Mpixels*bands/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(res['elapsed'] / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': res['scaling'], 'vs_baseline': None, 'dtype': 'f32 in/out, f64 window sums', 'data': 'synthetic',
            'config': dict(workload=res['workload'], baseline_config=args.config, **res['config']),
            'roofline': {
                'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBPS, 4), 'traffic': rl['traffic'], 'traffic_source': rl['traffic_source'],
                'kernel': rl['kernel'], 'avg_launch_ms': round(rl['avg_launch_ms'], 4),
                'algorithmic_bytes_per_launch': rl['achieved_bytes'],
                # the same fraction against what THIS box's HBM gives a flat 2-read 1-write float4 stream of the run's planes
                # (hk_stream_probe_dev, measured before the timed region; SURVEY.md 8(d) "reports both fractions")
                'copy_gbps_measured': round(rl['copy_gbps'], 1) if rl.get('copy_gbps') else None,
                'copy_gbps_measured_best': round(rl['copy_gbps_best'], 1) if rl.get('copy_gbps_best') else None,
                'frac_of_copy': round(achieved / rl['copy_gbps'], 4) if rl.get('copy_gbps') else None,
            },
            'cpu_baseline': cpu,
            'parity_spot_check': res['parity'],
        }
        if res.get('end_to_end') is not None:
            out['end_to_end'] = res['end_to_end']
        if res.get('projection') is not None:
            out['projected_scaling_single_gpu'] = res['projection']
        out['host_placement'] = topology.summary(placement)   # of rank 0; every rank binds to its own GPU's node
        if placements is not None:
            out['host_placement_ranks'] = placements             # ... and says so: one record per rank, in rank order
        if shard_checksum is not None:
            out['shard_checksum'] = shard_checksum
        if nan_variant is not None:
            out['nodata_nan_variant'] = nan_variant
        if blob_variant is not None:
            out['nodata_clustered_holes_variant'] = blob_variant
        if inpaint_variants is not None:
            out['inpainting_variants'] = inpaint_variants
        if other is not None:
            out['other_configs'] = other
        if res.get('power') is not None:
            out['power'] = res['power']
        if first_probe is not None:
            out['first_gpu_process_probe'] = first_probe   # rc 0: the child that used the GPU before this process ended normally
        if dist.backend() is not None:
            out['dist_backend'] = dist.backend()   # 'rccl': one GPU per rank, the library's RCCL communicator joined; 'host': ranks share a GPU; absent for a single process
            out['rccl_ranks'] = rccl_ranks         # ranks of the library's own communicator (hk_comm_info); None under 'host'
            if rccl_error:
                out['rccl_error'] = rccl_error
        print(json.dumps(out), flush=True)

    if not rccl_join_abandoned:
        ctx.close()
    dist.finalize()
    if first_probe is not None and first_probe['fatal']:
        sys.exit(3)
    if rccl_join_abandoned:   # a thread is still inside ncclCommInitRank on the context: the line is printed, the run is not green
        sys.stdout.flush(), sys.stderr.flush()
        os._exit(4)


if __name__ == '__main__':
    main()
