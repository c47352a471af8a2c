"""
Block statistics of a gain-blk-offset block whose ROWS are spread over several ranks / devices -- the one optional
collective of the hot path (reference: KernelModel._fit_block_norm, homonim/kernel_model.py:216-229).

    norm[0] = std(ref[mask]) / std(src[mask]);  norm[1] = percentile(ref[mask], 1) - percentile(src[mask], 1) * norm[0]

Every ingredient is a sum over pixels: the shifted float64 moments, and the integer histograms of the exact three-level
radix select that finds the two order statistics behind the percentile.  So each rank runs the library's kernels on its
slab (``hk_block_norm_split_dev``, phases 0..5) and the ranks all-reduce (SUM) a small float64 exchange buffer between the
phases: 2 + 5 + 3 x (<= 8192) values per band, five all-reduces per block.  In production the library does that itself
-- RCCL over xGMI through the context's communicator (``hk_comm_init``, ``hk_block_norm_split_comm_dev``), queued on the
job's stream without host synchronisation and without torch.  A caller-supplied ``reducer`` keeps a host-driven phase loop
instead (tests in which several ranks share one GPU all-reduce over gloo: tests/_torch_reducer.py); the product has none.

The order statistics are exactly those of the whole block; the std ratio equals the single-device value up to the order of
the float64 sums (the slabs' partial sums are added rank by rank).
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np

from homonim_amd import _hk

N_PHASES = 6


def block_norm_split(ctx: '_hk.Context', desc: '_hk.FitDesc', job: '_hk.DevJob', reducer=None,
                     norm_dev: Optional[int] = None) -> np.ndarray:
    """
    Block statistics over all ranks; ``job`` describes THIS rank's slab of the block (device planes, any number of rows --
    none is allowed --, the block's width; ``n_bands`` equal on every rank).  Collective: every rank calls it with its slab.
    Returns norm (n_bands, 2) float64, identical on every rank; ``norm_dev`` (optional device pointer, n_bands x 2 float64)
    receives it too, ready to be ``job.norm`` of the rank's ``fit_apply_dev``.

    ``reducer`` None (the production path): the context's own RCCL communicator (``dist.init_comm`` / ``Context.comm_init``)
    -- the library queues the six phases and the five all-reduces on the job's stream (``hk_block_norm_split_comm_dev``);
    no torch, no host synchronisation until the result is read.  A ``reducer`` -- any callable object with ``world_size``,
    ``ptr`` (device address of a float64 exchange buffer) and ``n_doubles`` that all-reduces (SUM) that buffer over the ranks
    when called -- keeps the phase loop on the host instead.
    """
    nb = int(job.n_bands)
    own = norm_dev is None
    if own:
        norm_dev = ctx.dev_alloc(16 * nb)
    if reducer is None:
        try:
            ctx.block_norm_split_comm_dev(desc, job, norm_dev)
            ctx.stream_sync(job.stream)
            out = np.zeros((nb, 2), np.float64)
            ctx.d2h(out, norm_dev)
            return out
        finally:
            if own:
                ctx.dev_free(norm_dev)
    if reducer.n_doubles < ctx.split_exchange_doubles(nb):
        if own:
            ctx.dev_free(norm_dev)
        raise ValueError('exchange buffer smaller than hk_block_norm_split_exchange_doubles(n_bands)')
    try:
        for phase in range(N_PHASES):
            ctx.block_norm_split_phase(desc, job, phase, reducer.world_size, reducer.ptr, norm_dev)
            ctx.stream_sync(job.stream)
            if phase < N_PHASES - 1:
                reducer()
        out = np.zeros((nb, 2), np.float64)
        ctx.d2h(out, norm_dev)
        return out
    finally:
        if own:
            ctx.dev_free(norm_dev)


def block_norm_split_local(parts: Sequence[Tuple['_hk.Context', '_hk.DevJob']], desc: '_hk.FitDesc') -> List[np.ndarray]:
    """
    The same protocol inside ONE process: ``parts`` = (context, slab job) per device (fuse.create_device_config(devices=[...])
    fan-out), the exchange buffers are summed on the host.  Returns every part's norm (they are identical).
    """
    world = len(parts)
    nb = int(parts[0][1].n_bands)
    n = parts[0][0].split_exchange_doubles(nb)
    bufs = [c.dev_alloc(8 * n) for c, _ in parts]
    norms = [c.dev_alloc(16 * nb) for c, _ in parts]
    try:
        for phase in range(N_PHASES):
            for (c, job), b, nd in zip(parts, bufs, norms):
                c.block_norm_split_phase(desc, job, phase, world, b, nd)
            host = []
            for (c, job), b in zip(parts, bufs):
                c.stream_sync(job.stream)
                h = np.zeros(n, np.float64)
                if phase < N_PHASES - 1:
                    c.d2h(h, b)
                host.append(h)
            if phase < N_PHASES - 1:
                total = np.sum(host, axis=0)  # rank order, like a ring all-reduce
                for (c, _), b in zip(parts, bufs):
                    c.h2d(b, total)
        out = []
        for (c, _), nd in zip(parts, norms):
            o = np.zeros((nb, 2), np.float64)
            c.d2h(o, nd)
            out.append(o)
        return out
    finally:
        for (c, _), b, nd in zip(parts, bufs, norms):
            c.dev_free(b), c.dev_free(nd)
