"""
Multi-process plumbing for one-process-per-GPU runs (``python -m torch.distributed.run ... bench.py``, ``bench.py --gpus N``,
or any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE).

The hot path has NO data-path collective: (band x block) work items are independent, every rank
works on its own shard and the only exchanges are a barrier and scalar reductions for timing / bookkeeping.  Since round 5 they
need no tensor library: the ranks of ONE node meet over loopback TCP --

* rank 0 listens on an ephemeral port of 127.0.0.1 and publishes it through a small file in a private (0700) directory under the
  temporary directory, named after
  the launch (MASTER_PORT / TORCHELASTIC_RUN_ID / HOMONIM_AMD_LAUNCH_ID) and exchanged with the nonce handshake of
  ``_file_rendezvous`` (a file an earlier, crashed launch left behind is never taken for this launch's);
* ``barrier`` / ``max_over_ranks`` / ``sum_over_ranks`` are a gather to rank 0 and a reply to everybody (N - 1 small messages each
  way; ~0.1 ms for 8 ranks -- they bracket timed regions, they are never inside one);
* the one real exchange of the hot path, the statistics of a gain-blk-offset block whose rows are spread over GPUs, runs on the
  LIBRARY's RCCL communicator (``init_comm``: the id travels over the same sockets; the collectives are queued by
  libhomonim_hk.so on the job's stream, homonim_amd/split_norm.py).

``backend()`` says which of the two this launch is: 'rccl' (one GPU per rank; ``init_comm`` is expected to work) or 'host' (several
ranks share a GPU or have none -- smoke tests; HOMONIM_AMD_DIST_BACKEND=host, the legacy value 'gloo' means the same).
"""
import os
import socket
import struct
import tempfile
import time
from typing import List, Optional, Tuple

_state = dict(initialised=False, world=1, rank=0, local_rank=0, backend=None, server=None, peers=None, sock=None)
_TIMEOUT = 900.0   # a rank may sit in a barrier while rank 0 times its CPU baseline; a lost peer still ends the wait


def env_ranks() -> Tuple[int, int, int]:
    """ (rank, world_size, local_rank) from the launcher's environment (1-process defaults). """
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def _launch_token() -> str:
    """ What names THIS launch among the launches of this user on this node: the launcher's MASTER_PORT / TORCHELASTIC_RUN_ID /
    HOMONIM_AMD_LAUNCH_ID; a launcher that sets none of them (mpirun, srun: RANK and WORLD_SIZE only) is named by the process that
    started the ranks -- the ranks of one launch share their parent, two concurrent launches do not. """
    parts = [os.environ.get(k, '') for k in ('MASTER_PORT', 'TORCHELASTIC_RUN_ID', 'HOMONIM_AMD_LAUNCH_ID')]
    if not any(parts):
        parts = ['ppid', str(os.getppid())]
    return '_'.join(''.join(c if c.isalnum() else '-' for c in p) for p in parts)


def _check_one_node(world: int):
    """ The rendezvous is loopback TCP + a file of the local temporary directory: ONE node.  A launch that spans nodes is refused at
    once, with the reason, instead of waiting out the rendezvous timeout (the hot path has no cross-node exchange either: its work
    items are independent; run one launch per node on a shard of the block list, homonim_amd.fuse.shard). """
    def env_int(key):
        try:
            return int(os.environ[key])
        except (KeyError, ValueError):
            return None
    local_world, nnodes, group_world = env_int('LOCAL_WORLD_SIZE'), env_int('NNODES'), env_int('GROUP_WORLD_SIZE')
    spans = (local_world is not None and local_world != world) or (nnodes or 1) > 1 or (group_world or 1) > 1
    if spans:
        raise RuntimeError(
            f'homonim_amd.dist: this launch spans nodes (WORLD_SIZE={world}, LOCAL_WORLD_SIZE={local_world}, NNODES={nnodes}, '
            f'GROUP_WORLD_SIZE={group_world}) -- the ranks of a launch meet over loopback TCP and must share one node; start one '
            f'launch per node, each on its shard of the work (homonim_amd.fuse.shard)')


def _private_dir() -> str:
    """ A directory of this user's own (mode 0700, owner checked) under the temporary directory: the port hand-over of a launch must
    not be readable or pre-creatable by another user of the box. """
    d = os.path.join(tempfile.gettempdir(), f'homonim_amd_{os.getuid()}')
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError(f'{d} is not a private directory of uid {os.getuid()} (mode {oct(st.st_mode & 0o777)}): refusing to rendezvous through it')
    return d


def _send_msg(sock: socket.socket, payload: bytes):
    sock.sendall(struct.pack('<I', len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError('a peer rank closed its connection')
        buf += chunk
    return bytes(buf)


def _recv_msg(sock: socket.socket) -> bytes:
    (n,) = struct.unpack('<I', _recv_exact(sock, 4))
    return _recv_exact(sock, n)


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """ Meet the other ranks of the launch named by the environment; a no-op for single-process runs. """
    rank, world, local_rank = env_ranks()
    _state.update(world=world, rank=rank, local_rank=local_rank)
    # HOMONIM_AMD_DIST_FORCE=1: set the plumbing up even alone (exercises it on a 1-GPU box)
    if (world > 1 or os.environ.get('HOMONIM_AMD_DIST_FORCE') == '1') and not _state['initialised']:
        if backend is None:
            backend = os.environ.get('HOMONIM_AMD_DIST_BACKEND')
        if backend is None:
            from homonim_amd import _hk
            try:
                n_dev = _hk.device_count()
            except Exception:
                n_dev = 0
            backend = 'rccl' if n_dev >= world else 'host'   # one GPU per rank, else the ranks share what there is
        backend = {'gloo': 'host', 'nccl': 'rccl'}.get(backend, backend)
        if backend not in ('host', 'rccl'):
            raise ValueError(f"unknown HOMONIM_AMD_DIST_BACKEND {backend!r}: 'rccl' (one GPU per rank) or 'host'")
        _check_one_node(world)
        path = os.path.join(_private_dir(), f'rdzv_{_launch_token()}')
        if rank == 0:
            server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            server.bind(('127.0.0.1', 0))
            server.listen(max(8, world))
            port = server.getsockname()[1]
            if world > 1:
                _file_rendezvous(path, 0, world, lambda: struct.pack('<I', port), timeout=_TIMEOUT)
            server.settimeout(_TIMEOUT)
            peers: List[Optional[socket.socket]] = [None] * world
            for _ in range(world - 1):
                conn, _addr = server.accept()
                conn.settimeout(_TIMEOUT)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                (r,) = struct.unpack('<I', _recv_msg(conn))
                if not (0 < r < world) or peers[r] is not None:
                    raise RuntimeError(f'rank 0: unexpected peer announced itself as rank {r}')
                peers[r] = conn
            for p in (path, *(f'{path}.{kind}{r}' for kind in ('hello', 'ack') for r in range(1, world))):
                try:
                    os.unlink(p)
                except OSError:
                    pass
            _state.update(server=server, peers=peers)
        else:
            (port,) = struct.unpack('<I', _file_rendezvous(path, rank, world, None, timeout=_TIMEOUT))
            sock = socket.create_connection(('127.0.0.1', port), timeout=_TIMEOUT)
            sock.settimeout(_TIMEOUT)
            sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send_msg(sock, struct.pack('<I', rank))
            _state.update(sock=sock)
        _state.update(initialised=True, backend=backend)
        barrier()   # everybody is in
    return rank, world, local_rank


def _exchange(payload: bytes, reduce_fn) -> bytes:
    """ Gather every rank's payload at rank 0, reduce, reply the result to all. """
    if not _state['initialised'] or _state['world'] == 1:
        return reduce_fn([payload])
    if _state['rank'] == 0:
        parts = [payload] + [_recv_msg(_state['peers'][r]) for r in range(1, _state['world'])]
        out = reduce_fn(parts)
        for r in range(1, _state['world']):
            _send_msg(_state['peers'][r], out)
        return out
    _send_msg(_state['sock'], payload)
    return _recv_msg(_state['sock'])


def _reduce(value: float, op_name: str) -> float:
    fn = {'MAX': max, 'SUM': sum}[op_name]
    out = _exchange(struct.pack('<d', float(value)), lambda parts: struct.pack('<d', fn(struct.unpack('<d', p)[0] for p in parts)))
    return struct.unpack('<d', out)[0]


def backend() -> Optional[str]:
    """ 'rccl' / 'host' once the ranks have met (None: single process, nothing to meet). """
    return _state['backend'] if _state['initialised'] else None


def barrier():
    if _state['initialised']:
        _exchange(b'', lambda parts: b'')


def max_over_ranks(value: float) -> float:
    return _reduce(value, 'MAX')


def sum_over_ranks(value: float) -> float:
    return _reduce(value, 'SUM')


def gather_bytes(payload: bytes) -> List[bytes]:
    """ every rank's ``payload``, in rank order, on every rank. """
    out = _exchange(payload, lambda parts: b''.join(struct.pack('<I', len(p)) + p for p in parts))
    parts, pos = [], 0
    while pos < len(out):
        (n,) = struct.unpack('<I', out[pos:pos + 4])
        parts.append(out[pos + 4:pos + 4 + n])
        pos += 4 + n
    return parts


def broadcast_bytes(payload: Optional[bytes]) -> bytes:
    """ rank 0's ``payload`` on every rank. """
    return _exchange(payload if (_state['rank'] == 0 and payload is not None) else b'', lambda parts: parts[0])


def exchange_comm_id() -> bytes:
    """ The id of the library's RCCL communicator for this launch, made by rank 0 (``hk_comm_unique_id``) and handed to every rank
    over the launch's sockets -- ON THE CALLING THREAD: the sockets carry one conversation at a time.  What may block without a
    deadline is ``Context.comm_init(uid, rank, world)`` (ncclCommInitRank returns when every rank has joined); a caller that wants
    a deadline runs only that on a side thread (bench.py). """
    from homonim_amd import _hk
    rank, world, _ = env_ranks()
    if not _state['initialised']:
        if world == 1:
            return _hk.comm_unique_id()
        raise RuntimeError('exchange_comm_id needs the ranks to have met (dist.init)')
    payload = None
    if rank == 0:   # a rank 0 that cannot make the id (no librccl) tells the others instead of leaving them in the exchange
        try:
            payload = b'\x01' + _hk.comm_unique_id()
        except Exception as ex:
            payload = b'\x00' + f'{type(ex).__name__}: {ex}'.encode()
    blob = broadcast_bytes(payload)
    if blob[:1] != b'\x01':
        raise RuntimeError('rank 0 could not make the RCCL communicator id: ' + blob[1:].decode(errors='replace'))
    return blob[1:]


def init_comm(ctx, id_file: Optional[str] = None) -> Tuple[int, int]:
    """
    Give ``ctx`` (a ``homonim_amd._hk.Context``) the library's own RCCL communicator over the ranks of this launch: rank 0
    makes the id (``hk_comm_unique_id``) and the others receive it -- over the sockets of ``init`` when the ranks have met, else
    through ``id_file`` (or ``$HOMONIM_AMD_COMM_FILE``; ``_file_rendezvous``).  The collectives themselves
    (``Context.block_norm_split_comm_dev``) are queued by the library.  -> (rank, world_size)
    """
    from homonim_amd import _hk
    rank, world, _ = env_ranks()
    id_file = id_file or os.environ.get('HOMONIM_AMD_COMM_FILE')
    if _state['initialised'] and id_file is None:
        uid = exchange_comm_id()
    elif world == 1 and id_file is None:
        uid = _hk.comm_unique_id()
    else:
        if id_file is None:
            raise RuntimeError('init_comm needs the ranks to have met (dist.init) or an id file (HOMONIM_AMD_COMM_FILE)')
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
    A small blob from rank 0 to the others through files, safe against whatever an earlier (crashed) launch left at the
    same path: every other rank announces itself with a fresh random nonce (``<id_file>.hello<r>``), rank 0 publishes the blob
    together with the nonces it saw, a rank takes only a file that names ITS nonce and acknowledges it
    (``<id_file>.ack<r>``), and rank 0 goes on -- e.g. into ncclCommInitRank, which has no timeout of its own -- only when every
    rank has acknowledged the file it last wrote.  A stale hello makes rank 0 publish once more when the fresh one arrives; a
    stale blob or ack file never matches a fresh nonce.  No launcher token is needed (rounds 3-4 keyed the file on MASTER_PORT /
    TORCHELASTIC_RUN_ID and accepted any young file when neither was set).  Every wait ends with a clear error after ``timeout``.
    """
    import json
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
                raise RuntimeError(f'rank {rank}: nothing of this launch in {id_file} after {timeout:.0f} s')
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
        try:
            barrier()   # nobody closes a socket somebody else still reads
        except Exception:
            pass
        for s in (_state.get('peers') or []):
            if s is not None:
                s.close()
        for key in ('sock', 'server'):
            if _state.get(key) is not None:
                _state[key].close()
        _state.update(initialised=False, backend=None, server=None, peers=None, sock=None)
