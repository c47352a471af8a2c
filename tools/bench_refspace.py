""" Kernel timing of the RefSpace chain (hk_refspace_fit_apply: cast-in, down-sample, statistics + fit on the reference grid,
up-sample the parameters + apply on the source grid) on one large block; run under rocprofv3 --kernel-trace --stats.
usage: python3 tools/bench_refspace.py [ratio] [size] """
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homonim_amd import _hk
ratio = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
m = n // ratio
rng = np.random.default_rng(1)
ref = rng.uniform(0.05, 1, (m, m)).astype(np.float32)
src = (np.kron(ref, np.ones((ratio, ratio), np.float32)) * 0.8 + 0.05 + rng.normal(0, 0.01, (m * ratio, m * ratio))).astype(np.float32)
ctx = _hk.get_context(0)
desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, np.nan, np.nan)
down = (float(ratio), 0., float(ratio), 0.)     # src = ratio * ref_index
up = (1. / ratio, 0., 1. / ratio, 0.)
out = ctx.pinned_empty(src.shape)
ps, pr = ctx.pinned_empty(src.shape), ctx.pinned_empty(ref.shape)
ps[:], pr[:] = src, ref
for rep in range(4):
    t0 = time.perf_counter()
    ctx.refspace_fit_apply(desc, ps, pr, down, up, 5, 3, False, 2, False, out_corr=out)
    t = time.perf_counter() - t0
    print('%d x %d source, ratio %d: %.2f ms per call = %.2f Gpx/s (host to host)' % (src.shape[0], src.shape[1], ratio, t * 1e3, src.size / t / 1e9))
