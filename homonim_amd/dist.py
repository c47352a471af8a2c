"""
Multi-process plumbing for one-process-per-GPU runs (``python -m torch.distributed.run ... bench.py``).

The hot path has NO data-path collective: (band x block) work items are independent, every rank
works on its own shard and the only exchanges are a barrier and scalar reductions for timing / bookkeeping.  Those go
through ``torch.distributed`` -- backend "nccl" (= RCCL over xGMI on ROCm) when this rank has a GPU, "gloo" otherwise
(CPU tests).  torch is imported lazily and only when WORLD_SIZE > 1.
"""
import os
from typing import Optional, Tuple

_state = dict(initialised=False, world=1, rank=0, local_rank=0, backend=None)


def env_ranks() -> Tuple[int, int, int]:
    """ (rank, world_size, local_rank) from the torchrun environment (1-process defaults). """
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """ Join the process group named by the environment; a no-op for single-process runs. """
    rank, world, local_rank = env_ranks()
    _state.update(world=world, rank=rank, local_rank=local_rank)
    # HOMONIM_AMD_DIST_FORCE=1: join the group even alone (exercises the RCCL plumbing on a 1-GPU box)
    if (world > 1 or os.environ.get('HOMONIM_AMD_DIST_FORCE') == '1') and not _state['initialised']:
        import torch
        import torch.distributed as dist
        if backend is None:
            # HOMONIM_AMD_DIST_BACKEND=gloo: several ranks sharing one GPU (smoke tests on a 1-GPU box)
            backend = os.environ.get('HOMONIM_AMD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend=backend)
        _state.update(initialised=True, backend=backend)
    return rank, world, local_rank


def _reduce(value: float, op_name: str) -> float:
    if not _state['initialised']:
        return float(value)
    import torch
    import torch.distributed as dist
    device = f"cuda:{_state['local_rank']}" if _state['backend'] == 'nccl' else 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op_name))
    return float(t.item())


def backend() -> Optional[str]:
    """ Backend of the joined process group (None: single process, no group). """
    return _state['backend'] if _state['initialised'] else None


def barrier():
    if _state['initialised']:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value: float) -> float:
    return _reduce(value, 'MAX')


def sum_over_ranks(value: float) -> float:
    return _reduce(value, 'SUM')


def init_comm(ctx, id_file: Optional[str] = None) -> Tuple[int, int]:
    """
    Give ``ctx`` (a ``homonim_amd._hk.Context``) the library's own RCCL communicator over the ranks of this launch: rank 0
    makes the id (``hk_comm_unique_id``) and the others receive it -- through the torch.distributed process group when one
    was joined (``init``; an object broadcast, launcher plumbing only), else through ``id_file`` (or ``$HOMONIM_AMD_COMM_FILE``:
    rank 0 writes it atomically, the others wait for it).  The collectives themselves (``Context.block_norm_split_comm_dev``)
    never touch torch.  -> (rank, world_size)
    """
    import time
    from homonim_amd import _hk
    rank, world, _ = env_ranks()
    id_file = id_file or os.environ.get('HOMONIM_AMD_COMM_FILE')
    if _state['initialised'] and id_file is None:
        import torch.distributed as dist
        box = [_hk.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    elif world == 1 and id_file is None:
        uid = _hk.comm_unique_id()
    else:
        if id_file is None:
            raise RuntimeError('init_comm needs a joined process group (dist.init) or an id file (HOMONIM_AMD_COMM_FILE)')
        # The file carries a header -- magic, the launch's token (MASTER_PORT / TORCHELASTIC_RUN_ID / HOMONIM_AMD_LAUNCH_ID),
        # rank 0's clock -- so that a file left behind by another or an earlier launch is not taken for this one's: rank 0
        # removes whatever is there before it makes the id, the other ranks skip files with a foreign token or older than
        # STALE_S, and rank 0 removes the file once the communicator stands (ncclCommInitRank returns when every rank has
        # joined, i.e. has read it).  Use a fresh path per launch where a crashed launch may be restarted within STALE_S.
        import struct
        STALE_S = 600.0
        token = '|'.join(os.environ.get(k, '') for k in ('MASTER_PORT', 'TORCHELASTIC_RUN_ID', 'HOMONIM_AMD_LAUNCH_ID')).encode()[:64]
        head = struct.Struct('<8s64sd')
        if rank == 0:
            try:
                os.unlink(id_file)
            except FileNotFoundError:
                pass
            uid = _hk.comm_unique_id()
            tmp = f'{id_file}.tmp{os.getpid()}'
            with open(tmp, 'wb') as f:
                f.write(head.pack(b'HKCOMM01', token, time.time()) + uid)
            os.replace(tmp, id_file)
        else:
            t0 = time.time()
            uid = None
            while uid is None:
                try:
                    with open(id_file, 'rb') as f:
                        blob = f.read()
                    if len(blob) > head.size:
                        magic, tok, stamp = head.unpack(blob[:head.size])
                        if magic == b'HKCOMM01' and tok.rstrip(b'\0') == token and stamp >= t0 - STALE_S:
                            uid = blob[head.size:]
                except FileNotFoundError:
                    pass
                if uid is None:
                    if time.time() - t0 > 120:
                        raise RuntimeError(f'rank {rank}: no communicator id of this launch in {id_file} after 120 s')
                    time.sleep(0.02)
        ctx.comm_init(uid, rank, world)
        if rank == 0:
            try:
                os.unlink(id_file)
            except OSError:
                pass
        return rank, world
    ctx.comm_init(uid, rank, world)
    return rank, world


def finalize():
    if _state['initialised']:
        import torch.distributed as dist
        dist.destroy_process_group()
        _state['initialised'] = False
