"""
Worker of tests/test_gpu_split_norm.py for the library's own RCCL path: plain processes (no torch.distributed.run, no torch
at all), one per GPU, ranks from RANK / WORLD_SIZE / LOCAL_RANK, the communicator id handed over through a file
(homonim_amd.dist.init_comm).  Every rank holds a slab of rows of one 3-band block -- with more than two ranks the last one
holds NO rows -- and takes part in hk_block_norm_split_comm_dev; every rank saves its result.  Not a test module.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    out_dir, variant = sys.argv[1], sys.argv[2]
    from homonim_amd import _hk, dist, split_norm
    from oracle import oracle_np as onp  # input generator only (test infrastructure)
    rank, world, local_rank = dist.env_ranks()
    ctx = _hk.Context(local_rank % _hk.device_count(), n_streams=2)
    assert ctx.comm_info() == (-1, 0)
    try:
        joined = dist.init_comm(ctx, os.path.join(out_dir, 'comm_id.bin'))
    except Exception as ex:   # noqa: BLE001 -- the RCCL bootstrap of THIS box (interfaces, IPC): not what the test is about
        sys.stderr.write(f'COMM_INIT_FAILED rank {rank}: {ex}\n')
        sys.exit(77)
    assert joined == (rank, world)
    assert ctx.comm_info() == (rank, world)
    h, w, nb = 613, 1003, 3
    pairs = [onp.synth_pair(h, w, 700 + b, variant) for b in range(nb)]
    src, ref = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    # uneven slabs of rows; with three or more ranks the last one has none
    holders = world if world < 3 else world - 1
    edges = [0] + [int(h * (0.37 + 0.63 * (r + 1) / holders)) if r + 1 < holders else h for r in range(holders)]
    edges += [h] * (world - holders)
    r0, r1 = edges[rank], edges[rank + 1]
    rows = r1 - r0
    stride = (w + 63) // 64 * 64
    pad = lambda a: np.ascontiguousarray(np.pad(a[:, r0:r1], ((0, 0), (0, 0), (0, stride - w))), np.float32)  # noqa: E731
    d_src, d_ref = ctx.dev_alloc(4 * stride * max(rows, 1) * nb), ctx.dev_alloc(4 * stride * max(rows, 1) * nb)
    if rows:
        ctx.h2d(d_src, pad(src)), ctx.h2d(d_ref, pad(ref))
    nd = np.nan if variant != 'none' else None
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, nd, nd)
    job = _hk.DevJob()
    job.src, job.ref = d_src, d_ref
    job.corr = job.gain = job.offset = job.r2 = job.norm = job.fail_count = None
    job.n_bands, job.height, job.width, job.stride, job.band_stride = nb, rows, w, stride, stride * max(rows, 1)
    job.seg_rows, job.stream = 0, 1
    norm = split_norm.block_norm_split(ctx, desc, job)            # reducer None: hk_block_norm_split_comm_dev
    norm2 = split_norm.block_norm_split(ctx, desc, job)           # the communicator and its buffer are reusable
    assert (norm == norm2).all()
    # the exported primitive: in-place SUM of a device buffer over the ranks
    d = ctx.dev_alloc(8 * 5)
    ctx.h2d(d, np.arange(5, dtype=np.float64) + rank)
    ctx.comm_allreduce_f64_dev(d, 5, stream=0)
    ctx.stream_sync(0)
    got = np.zeros(5)
    ctx.d2h(got, d)
    assert (got == world * np.arange(5) + sum(range(world))).all(), got
    ctx.dev_free(d)
    np.save(os.path.join(out_dir, f'norm_{rank}.npy'), norm)
    assert 'torch' not in sys.modules, 'the RCCL path of the split-block statistics must not import torch'
    ctx.dev_free(d_src), ctx.dev_free(d_ref)
    ctx.comm_destroy()
    assert ctx.comm_info() == (-1, 0)
    ctx.close()


if __name__ == '__main__':
    main()
