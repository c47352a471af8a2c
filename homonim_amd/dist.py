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
        uid = _file_rendezvous(id_file, rank, world, _hk.comm_unique_id)
        ctx.comm_init(uid, rank, world)
        if rank == 0:   # ncclCommInitRank returns when every rank has joined, i.e. has read the id
            for path in [id_file] + [f'{id_file}.{kind}{r}' for kind in ('hello', 'ack') for r in range(1, world)]:
                try:
                    os.unlink(path)
                except OSError:
                    pass
        return rank, world
    ctx.comm_init(uid, rank, world)
    return rank, world


def _file_rendezvous(id_file: str, rank: int, world: int, make_uid, timeout: float = 120.0) -> bytes:
    """
    The communicator id from rank 0 to the others through files, safe against whatever an earlier (crashed) launch left at the
    same path: every other rank announces itself with a fresh random nonce (``<id_file>.hello<r>``), rank 0 publishes the id
    together with the nonces it saw, a rank takes only an id file that names ITS nonce and acknowledges it
    (``<id_file>.ack<r>``), and rank 0 goes on -- into ncclCommInitRank, which has no timeout of its own -- only when every
    rank has acknowledged the file it last wrote.  A stale hello makes rank 0 publish once more when the fresh one arrives; a
    stale id or ack file never matches a fresh nonce.  No launcher token is needed (rounds 3-4 keyed the file on MASTER_PORT /
    TORCHELASTIC_RUN_ID and accepted any young file when neither was set).  Every wait ends with a clear error after ``timeout``.
    """
    import json
    import struct
    import time
    magic = b'HKCOMM02'

    def put(path, blob):
        tmp = f'{path}.tmp{os.getpid()}.{rank}'
        with open(tmp, 'wb') as f:
            f.write(blob)
        os.replace(tmp, path)

    def get(path):
        try:
            with open(path, 'rb') as f:
                return f.read()
        except OSError:
            return None

    t0 = time.time()
    if rank != 0:
        nonce = os.urandom(8).hex()
        put(f'{id_file}.hello{rank}', nonce.encode())
        while True:
            blob = get(id_file)
            if blob and blob[:8] == magic and len(blob) > 12:
                n = struct.unpack('<I', blob[8:12])[0]
                try:
                    seen = json.loads(blob[12:12 + n].decode())
                except ValueError:
                    seen = {}
                if seen.get(str(rank)) == nonce:
                    put(f'{id_file}.ack{rank}', nonce.encode())
                    return blob[12 + n:]
            if time.time() - t0 > timeout:
                raise RuntimeError(f'rank {rank}: no communicator id of this launch in {id_file} after {timeout:.0f} s')
            time.sleep(0.01)
    try:
        os.unlink(id_file)
    except OSError:
        pass
    uid = make_uid()
    written = None
    while True:
        cur = {}
        for r in range(1, world):
            blob = get(f'{id_file}.hello{r}')
            if blob:
                cur[str(r)] = blob.decode(errors='replace')
        if len(cur) == world - 1 and cur != written:
            head = json.dumps(cur).encode()
            put(id_file, magic + struct.pack('<I', len(head)) + head + uid)
            written = cur
        if written is not None and all((get(f'{id_file}.ack{r}') or b'').decode(errors='replace') == written[str(r)] for r in range(1, world)):
            return uid
        if time.time() - t0 > timeout:
            missing = [r for r in range(1, world) if str(r) not in cur]
            raise RuntimeError(f'rank 0: ranks {missing or "(all announced, not all acknowledged)"} did not join through {id_file} within {timeout:.0f} s')
        time.sleep(0.01)


def finalize():
    if _state['initialised']:
        import torch.distributed as dist
        dist.destroy_process_group()
        _state['initialised'] = False
