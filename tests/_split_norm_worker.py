"""
Worker of tests/test_gpu_split_norm.py, started by `python -m torch.distributed.run --nproc-per-node N`: every rank holds a
slab of rows of one 3-band block and takes part in the split-block statistics (homonim_amd/split_norm.py); rank 0 saves
the result.  On a 1-GPU box the ranks share device 0 (torch gloo group, or a torch RCCL group of one).  Not a test module.
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
    # the all-reduce of this test is torch.distributed's (gloo when the ranks share a GPU).  torch brings a HIP runtime of its own:
    # it has to open the device BEFORE the library's runtime makes its first call (hk_device_count included) -- the other order
    # leaves torch with "No HIP GPUs are available"
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from _torch_reducer import TorchReducer, init_torch_group
    init_torch_group(dist.env_ranks()[2])
    rank, world, local_rank = dist.init()
    dev = local_rank % _hk.device_count()
    ctx = _hk.Context(dev, n_streams=2)
    h, w, nb = 613, 1003, 3
    pairs = [onp.synth_pair(h, w, 700 + b, variant) for b in range(nb)]
    src, ref = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    # uneven slabs of rows: rank r holds rows [edges[r], edges[r + 1])
    edges = [0] + [int(h * (0.37 + 0.63 * (r + 1) / world)) if r + 1 < world else h for r in range(world)]
    r0, r1 = edges[rank], edges[rank + 1]
    rows = r1 - r0
    stride = (w + 63) // 64 * 64
    pad = lambda a: np.ascontiguousarray(np.pad(a[:, r0:r1], ((0, 0), (0, 0), (0, stride - w))), np.float32)  # noqa: E731
    d_src, d_ref = ctx.dev_alloc(4 * stride * rows * nb), ctx.dev_alloc(4 * stride * rows * nb)
    ctx.h2d(d_src, pad(src)), ctx.h2d(d_ref, pad(ref))
    nd = np.nan if variant != 'none' else None
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, nd, nd)
    job = _hk.DevJob()
    job.src, job.ref = d_src, d_ref
    job.corr = job.gain = job.offset = job.r2 = job.norm = job.fail_count = None
    job.n_bands, job.height, job.width, job.stride, job.band_stride = nb, rows, w, stride, stride * rows
    job.seg_rows, job.stream = 0, 1
    reducer = TorchReducer(ctx.split_exchange_doubles(nb), dev)
    norm = split_norm.block_norm_split(ctx, desc, job, reducer)
    norm2 = split_norm.block_norm_split(ctx, desc, job, reducer)   # the buffers are reusable
    assert (norm == norm2).all()
    np.save(os.path.join(out_dir, f'norm_{rank}.npy'), norm)
    if rank == 0:
        with open(os.path.join(out_dir, 'backend.txt'), 'w') as f:
            f.write(f'{reducer.backend} {world}\n')
    ctx.dev_free(d_src), ctx.dev_free(d_ref)
    del reducer
    ctx.close()
    dist.barrier()
    dist.finalize()
    import torch.distributed as tdist
    tdist.destroy_process_group()


if __name__ == '__main__':
    main()
