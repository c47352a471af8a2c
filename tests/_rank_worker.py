"""
Worker of tests/test_gpu_multirank.py, started by `python -m torch.distributed.run --nproc-per-node N`: every rank runs
its shard of RasterFuse.process (one process per GPU; on a 1-GPU box the ranks share device 0 and rendezvous over gloo)
and saves what it produced.  Not a test module.
"""
import os
import sys
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    out_dir, model, k, contiguous = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4] == '1'
    from homonim_amd import _hk, dist
    from homonim_amd.fuse import RasterFuse
    from oracle import oracle_np as onp  # input generator only (test infrastructure)
    rank, world, local_rank = dist.init()
    n_dev = _hk.device_count()
    pairs = [onp.synth_pair(520, 700, 300 + b, 'frame+holes') for b in range(3)]
    src, ref = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        corr, params = RasterFuse(src, ref).process(
            None, model, (k, k), param_filename=True, model_config=dict(r2_inpaint_thresh=0.25),
            block_config=dict(threads=2, max_block_mem=0.3),
            device_config=dict(devices=[local_rank % n_dev], rank=rank, world_size=world, contiguous=contiguous))
    np.save(os.path.join(out_dir, f'corr_{rank}.npy'), corr)
    np.save(os.path.join(out_dir, f'params_{rank}.npy'), params)
    # the bookkeeping collectives of a sharded run: every rank learns the global number of pixels it takes part in
    n_mine = float((~np.isnan(corr)).sum())
    total = dist.sum_over_ranks(n_mine)
    slowest = dist.max_over_ranks(float(rank))
    dist.barrier()
    if rank == 0:
        with open(os.path.join(out_dir, 'summary.txt'), 'w') as f:
            f.write(f'{world} {int(total)} {int(slowest)}\n')
    dist.finalize()


if __name__ == '__main__':
    main()
