"""
CPU tests of the block scheduler: the partition rule against tables produced by the reference's own
raster_pair.py (tests/golden/block_pairs.json, oracle/gen_golden.py), sharding, and the N>1 rank path under gloo.
"""
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest

from conftest import GOLDEN_DIR, REPO
from homonim_amd import fuse, utils
from homonim_amd.errors import BlockSizeError


def _golden_blocks():
    with open(os.path.join(GOLDEN_DIR, 'block_pairs.json')) as f:
        return json.load(f)['cases']


@pytest.mark.parametrize('case', _golden_blocks(), ids=lambda c: f"{c['height']}x{c['width']}x{c['n_bands']}-k{c['kernel_shape']}-m{c['max_block_mem']}-{c['proc_crs']}")
def test_block_partition_matches_reference(case):
    mem = float('inf') if case['max_block_mem'] is None else case['max_block_mem']
    shape = (case['height'], case['width'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert list(fuse.auto_block_shape(shape, mem)) == case['block_shape']
        overlap = utils.overlap_for_kernel(case['kernel_shape'])
        assert list(overlap) == case['overlap']
        bps = list(fuse.block_pairs(shape, case['n_bands'], overlap, mem))
    assert len(bps) == case['n_blocks']
    rows = [[bp.band_i, *bp.src_in_block, *bp.src_out_block, *bp.ref_in_block, *bp.ref_out_block, bp.outer] for bp in bps]
    exp = case['block_pairs']
    if len(rows) > 70:
        rows = rows[:35] + rows[-35:]
    assert rows == exp
    # out-blocks tile each band exactly once
    cover = np.zeros((case['n_bands'], *shape), np.uint8) if case['height'] * case['width'] <= 4096 * 4096 else None
    if cover is not None:
        for bp in bps:
            rs, cs = bp.src_out_block.toslices()
            cover[bp.band_i][rs, cs] += 1
        assert (cover == 1).all()


def test_baseline_sizes_appendix_b():
    """ 16384^2 at max_block_mem=100 -> 4096^2 blocks, 16 per band; in-block rows with k=5. """
    assert fuse.auto_block_shape((16384, 16384), 100) == (4096, 4096)
    assert fuse.auto_block_shape((8192, 8192), 100) == (4096, 4096)
    bps = list(fuse.block_pairs((16384, 16384), 1, (3, 3), 100))
    assert len(bps) == 16
    rows = sorted({(bp.src_in_block.row_off, bp.src_in_block.row_off + bp.src_in_block.height) for bp in bps})
    assert rows == [(0, 4099), (4093, 8195), (8189, 12291), (12285, 16384)]


def test_block_size_errors():
    with pytest.raises(BlockSizeError):
        fuse.auto_block_shape((100, 100), 1e-9)
    with pytest.raises(BlockSizeError), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        list(fuse.block_pairs((64, 64), 1, (16, 16), 64 * 8 * 4 / 2**20))


def test_shard_is_a_partition():
    items = list(range(37))
    for n in (1, 2, 3, 8):
        parts = [fuse.shard(items, i, n) for i in range(n)]
        assert sorted(sum(parts, [])) == items
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        fuse.shard(items, 2, 2)


def test_config_factories():
    assert fuse.RasterFuse.create_block_config()['max_block_mem'] == 100
    assert fuse.RasterFuse.create_block_config(threads=1)['threads'] == 1
    prof = fuse.RasterFuse.create_out_profile()
    assert prof['driver'] == 'GTiff' and prof['dtype'] == 'float32' and np.isnan(prof['nodata'])
    assert prof['creation_options']['compress'] == 'deflate'
    assert fuse.RasterFuse.create_model_config()['r2_inpaint_thresh'] == 0.25
    with pytest.raises(TypeError):
        fuse.RasterFuse.create_block_config(bogus=1)


def test_convert_dtype_round_clip_nodata():
    """ reference tests/test_raster_array.py:297-358 semantics: round, clip, nan -> nodata. """
    a = np.array([[-5.4, 0.5, 1.5, 2.5], [254.6, 300., np.nan, 7.49]], np.float32)
    out = fuse.convert_dtype(a, 'uint8', 0)
    assert out.dtype == np.uint8
    assert out.tolist() == [[0, 0, 2, 2], [255, 255, 0, 7]]
    out16 = fuse.convert_dtype(a, 'int16', -9999)
    assert out16.tolist() == [[-5, 0, 2, 2], [255, 300, -9999, 7]]


_WORKER = r'''
import os, sys, json
sys.path.insert(0, {repo!r})
from homonim_amd import dist, fuse, utils
rank, world, local_rank = dist.init()          # loopback TCP between the ranks (backend 'host': no GPU here)
blocks = list(fuse.block_pairs((3000, 2000), 3, utils.overlap_for_kernel((5, 5)), 4))
mine = fuse.shard(blocks, rank, world)
px = sum(b.src_out_block.width * b.src_out_block.height for b in mine)
dist.barrier()
total = dist.sum_over_ranks(px)
slowest = dist.max_over_ranks(float(rank + 1))
n_blocks = dist.sum_over_ranks(len(mine))
if rank == 0:
    print(json.dumps(dict(world=world, total=total, slowest=slowest, n_blocks=n_blocks, all_blocks=len(blocks))))
dist.finalize()
'''


@pytest.mark.timeout(180)
def test_rank_sharding_gloo_world2(tmp_path):
    """ The N>1 path of bench.py / RasterFuse under the driver's launcher: every rank takes a disjoint shard; barrier and scalar
    reductions over the ranks' own loopback rendezvous (homonim_amd/dist.py; HOMONIM_AMD_DIST_BACKEND=gloo is the legacy name of
    'host').  The torch gloo all-reduce of two ranks is tests/test_split_norm_cpu.py::test_protocol_with_two_gloo_ranks. """
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER.format(repo=REPO))
    # 'host' explicitly (legacy spelling): on a box WITH GPUs the default is 'rccl', which wants one device per rank
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', HOMONIM_AMD_DIST_BACKEND='gloo')
    res = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
         '--master-port', '29617', str(script)], capture_output=True, text=True, env=env, timeout=170
    )
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['world'] == 2 and out['total'] == 3000 * 2000 * 3 and out['slowest'] == 2.0
    assert out['n_blocks'] == out['all_blocks']


def test_convert_dtype_matches_reference_goldens():
    """ host-side statement of RasterArray._convert_array_dtype vs outputs of the reference's own code
    (tests/golden/convert_dtype.npz, oracle/gen_golden.py). """
    g = np.load(os.path.join(GOLDEN_DIR, 'convert_dtype.npz'))
    a = g['input']
    for key in g.files:
        if key == 'input':
            continue
        dtype, nd = key.rsplit('_', 1)
        nodata = float('nan') if nd == 'nan' else float(nd)
        out = fuse.convert_dtype(a.copy(), dtype, nodata)
        exp = g[key]
        assert out.dtype == exp.dtype, key
        np.testing.assert_array_equal(out, exp, err_msg=key)


def _golden_multires():
    with open(os.path.join(GOLDEN_DIR, 'block_pairs_multires.json')) as f:
        return json.load(f)['cases']


@pytest.mark.parametrize('case', _golden_multires(), ids=lambda c: f"src{c['src']['res']}-ref{c['ref']['res']}-{c['proc_crs']}-k{c['kernel_shape']}")
def test_multires_block_partition_matches_reference(case):
    """ Windows + blocks for source / reference pairs of different resolution and origin vs tables from the reference's
    own RasterPairReader (tests/golden/block_pairs_multires.json). """
    from homonim_amd.geo import Affine
    mk = lambda g: fuse.Grid(Affine(g['res'], 0., g['origin'][0], 0., -g['res'], g['origin'][1]), g['height'], g['width'])
    src, ref = mk(case['src']), mk(case['ref'])
    src_win, ref_win = fuse.pair_windows(src, ref)
    assert list(src_win) == case['src_win'] and list(ref_win) == case['ref_win']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        bps = list(fuse.block_pairs_multires(src, ref, case['proc_crs'], case['n_bands'], case['overlap'],
                                             case['max_block_mem']))
    assert len(bps) == case['n_blocks']
    rows = [[bp.band_i, *bp.src_in_block, *bp.src_out_block, *bp.ref_in_block, *bp.ref_out_block, bp.outer] for bp in bps]
    if len(rows) > 60:
        rows = rows[:30] + rows[-30:]
    assert rows == case['block_pairs']


def test_window_helpers():
    from homonim_amd.geo import Window
    assert fuse.expand_window_to_grid(Window(1.2, 3.7, 4.1, 2.0)) == Window(1, 3, 5, 3)
    assert fuse.expand_window_to_grid(Window(-0.5, 2.0, 3.0, 3.0), (1, 2)) == Window(-3, 1, 8, 5)
    assert fuse.round_window_to_grid(Window(1.5, 2.5, 3.0, 3.0)) == Window(2, 2, 2, 4)   # half to even, like np.round


def test_communicator_id_travels_through_a_file(tmp_path, monkeypatch):
    """ dist.init_comm without a torch process group: rank 0 makes the RCCL communicator id (hk_comm_unique_id) and writes it
    atomically, the other ranks wait for the file; every rank then joins with the same 128 bytes (Context.comm_init).  The
    library calls are replaced by recorders here (no GPU); the GPU side is tests/test_gpu_split_norm.py. """
    import threading
    from homonim_amd import _hk, dist
    uid = bytes(range(128))
    made = []
    monkeypatch.setattr(_hk, 'comm_unique_id', lambda: made.append(1) or uid)
    monkeypatch.setattr(dist, '_state', dict(dist._state, initialised=False))
    joined = {}
    all_in = threading.Barrier(3)

    class FakeCtx:
        def __init__(self, rank, collective=True):
            self.rank, self.collective = rank, collective

        def comm_init(self, unique_id, rank, world):
            joined[rank] = (unique_id, world)
            if self.collective and world > 1:
                all_in.wait(timeout=30)   # ncclCommInitRank returns once every rank has joined

    path = str(tmp_path / 'comm_id.bin')
    # a file left behind by an earlier launch (another MASTER_PORT) is not this launch's id: rank 0 replaces it, the others skip it
    # ... whatever its format and however young, with no launcher token in the environment at all: an id file, a hello and an
    # acknowledgement of a crashed launch of THIS protocol (their nonces are not this launch's)
    import json
    import struct
    head = json.dumps({'1': 'feedfacefeedface', '2': 'deadbeefdeadbeef'}).encode()
    with open(path, 'wb') as f:
        f.write(b'HKCOMM02' + struct.pack('<I', len(head)) + head + bytes(128))
    for name, text in (('hello1', 'feedfacefeedface'), ('ack1', 'feedfacefeedface'), ('ack2', 'deadbeefdeadbeef')):
        with open(f'{path}.{name}', 'w') as f:
            f.write(text)
    for key in ('MASTER_PORT', 'TORCHELASTIC_RUN_ID', 'HOMONIM_AMD_LAUNCH_ID'):
        monkeypatch.delenv(key, raising=False)
    local = threading.local()
    monkeypatch.setattr(dist, 'env_ranks', lambda: (local.rank, 3, local.rank))

    def run(rank):
        local.rank = rank
        assert dist.init_comm(FakeCtx(rank), path) == (rank, 3)

    waiters = [threading.Thread(target=run, args=(r,)) for r in (1, 2)]
    [t.start() for t in waiters]          # they poll for the file first
    run(0)
    [t.join(timeout=30) for t in waiters]
    assert made == [1]                     # only rank 0 asked the library for an id
    assert joined == {0: (uid, 3), 1: (uid, 3), 2: (uid, 3)}
    assert not os.path.exists(path)        # rank 0 removes the id once the communicator stands
    # a world of one needs neither a group nor a file
    monkeypatch.setattr(dist, 'env_ranks', lambda: (0, 1, 0))
    assert dist.init_comm(FakeCtx(0)) == (0, 1) and joined[0] == (uid, 1)
    # several ranks without either: a clear error, before any collective call
    monkeypatch.setattr(dist, 'env_ranks', lambda: (1, 2, 1))
    monkeypatch.delenv('HOMONIM_AMD_COMM_FILE', raising=False)
    with pytest.raises(RuntimeError):
        dist.init_comm(FakeCtx(1))


_WORKER8 = r'''
import os, sys, json, time
sys.path.insert(0, {repo!r})
from homonim_amd import dist
rank, world, local_rank = dist.init()
assert dist.backend() == 'host'
t0 = time.perf_counter()
for _ in range(200):
    dist.barrier()
per_barrier_us = (time.perf_counter() - t0) / 200 * 1e6
assert dist.max_over_ranks(float(rank)) == world - 1
assert dist.sum_over_ranks(float(rank)) == world * (world - 1) / 2
blob = dist.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
assert blob == bytes(range(128))
parts = dist.gather_bytes(b'rank %d' % rank + b'!' * rank)      # every rank's payload on every rank, in rank order
assert parts == [b'rank %d' % r + b'!' * r for r in range(world)]
slowest = dist.max_over_ranks(per_barrier_us)
if rank == 0:
    print(json.dumps(dict(world=world, barrier_us=slowest)))
dist.finalize()
'''


@pytest.mark.timeout(300)
@pytest.mark.parametrize('launcher', ['torchrun', 'plain'])
def test_eight_ranks_meet_without_a_tensor_library(tmp_path, launcher):
    """ The driver's 8-GPU launch (python -m torch.distributed.run --nproc-per-node 8) and a plain launcher that only sets RANK /
    WORLD_SIZE / MASTER_PORT: eight ranks meet over loopback TCP (homonim_amd/dist.py), barrier, reduce and broadcast -- the
    plumbing bench.py --gpus 8 brackets its timed region with, here without GPUs.  A barrier of 8 ranks stays well under a
    millisecond (it never sits inside a timed region, only around one). """
    script = tmp_path / 'worker8.py'
    script.write_text(_WORKER8.format(repo=REPO))
    env = dict(os.environ, OMP_NUM_THREADS='1', HOMONIM_AMD_DIST_BACKEND='host')
    for key in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'TORCHELASTIC_RUN_ID'):
        env.pop(key, None)
    if launcher == 'torchrun':
        res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=8', '--master-addr', '127.0.0.1',
                              '--master-port', '29641', str(script)], capture_output=True, text=True, env=env, timeout=280)
        assert res.returncode == 0, res.stderr[-2000:]
        out = res.stdout
    else:
        procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='8', MASTER_PORT='29642'),
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(8)]
        outs = [p.communicate(timeout=280) for p in procs]
        assert all(p.returncode == 0 for p in procs), [e[-500:] for _, e in outs]
        out = outs[0][0]
    rec = json.loads([ln for ln in out.splitlines() if ln.startswith('{')][-1])
    assert rec['world'] == 8 and rec['barrier_us'] < 5000, rec


def test_bench_launcher_reports_a_failing_rank():
    """ `python bench.py --gpus 2` starts its own ranks; without a GPU both fail loudly (no CPU fallback) and the launcher --
    which itself makes no GPU call -- stops the others and hands back a non-zero exit code, no JSON line. """
    env = dict(os.environ, PYTHONPATH=REPO, HIP_VISIBLE_DEVICES='-1', HOMONIM_AMD_DIST_BACKEND='gloo')
    for key in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(key, None)
    run = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--config', '1'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode not in (0, 2), run.stderr[-2000:]
    assert 'rank' in run.stderr and not [ln for ln in run.stdout.splitlines() if ln.startswith('{')]


def test_projection_arithmetic_and_the_as_rank_argument(monkeypatch):
    """ bench.py's single-GPU projection of strong scaling (no GPU needed for its arithmetic): an N-rank step lasts as long as its
    slowest shard; speed-up = t(1) / max shard, efficiency = speed-up / N.  `--as-rank R/N` is refused outside configs 3 / 4. """
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(REPO, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    table = {2: [5.0, 5.5], 4: [2.6, 2.9, 2.7, 2.5], 8: [1.4] * 7 + [1.6]}
    proj = bench.project_scaling(10.0, lambda r, n: table[n][r])
    assert 'PROJECTION' in proj['kind'] and proj['t1_ms'] == 10.0
    by = proj['by_world_size']
    assert by['2']['max_shard_ms'] == 5.5 and by['2']['projected_speedup'] == pytest.approx(10 / 5.5, abs=1e-3)
    assert by['4']['projected_efficiency'] == pytest.approx(10 / (4 * 2.9), abs=1e-4)
    assert by['8']['shard_ms'] == table[8] and by['8']['projected_speedup'] == pytest.approx(6.25)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--config', '3', '--as-rank', '3/8'])
    assert bench.parse_args().as_rank == (3, 8)
    for bad in (['--config', '2', '--as-rank', '0/2'], ['--config', '3', '--as-rank', '8/8'], ['--config', '4', '--as-rank', 'x'],
                ['--config', '3', '--gpus', '2', '--as-rank', '0/2']):
        monkeypatch.setattr(sys, 'argv', ['bench.py'] + bad)
        with pytest.raises(SystemExit):
            bench.parse_args()
    assert bench.SoloDist.max_over_ranks(3.5) == 3.5 and bench.SoloDist.backend() is None


_WORKER_NOID = r'''
import os, sys, time
sys.path.insert(0, {repo!r})
from homonim_amd import _hk, dist
rank, world, _ = dist.init()
def no_rccl():
    raise _hk.DeviceError('librccl not found (stand-in)')
_hk.comm_unique_id = no_rccl
class Ctx:
    def comm_init(self, uid, r, w):
        raise AssertionError('nobody may get as far as ncclCommInitRank')
t0 = time.time()
try:
    dist.init_comm(Ctx())
    sys.exit(5)
except RuntimeError as ex:
    assert 'could not make the RCCL communicator id' in str(ex) and 'librccl not found' in str(ex), str(ex)
assert time.time() - t0 < 30
assert dist.sum_over_ranks(1.0) == world      # the launch's sockets are still in step
dist.finalize()
'''


@pytest.mark.timeout(120)
def test_a_rank_zero_without_rccl_tells_the_others(tmp_path):
    """ dist.init_comm over the launch's sockets: if rank 0 cannot make the communicator id, EVERY rank gets the error at once -- nobody
    is left waiting in the exchange (or, worse, inside ncclCommInitRank, which has no timeout). """
    script = tmp_path / 'worker_noid.py'
    script.write_text(_WORKER_NOID.format(repo=REPO))
    env = dict(os.environ, OMP_NUM_THREADS='1', HOMONIM_AMD_DIST_BACKEND='host', WORLD_SIZE='3', MASTER_PORT='29651')
    env.pop('HOMONIM_AMD_COMM_FILE', None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=100) for p in procs]
    assert all(p.returncode == 0 for p in procs), [e[-800:] for _, e in outs]


def test_a_launch_that_spans_nodes_is_refused_at_once(monkeypatch):
    """ The rendezvous is loopback TCP + a local file: a launch whose environment says it spans nodes (torchrun --nnodes 2:
    LOCAL_WORLD_SIZE < WORLD_SIZE, GROUP_WORLD_SIZE 2) fails in dist.init with the reason, not after the 900 s rendezvous timeout
    (round-5 advisor finding). """
    from homonim_amd import dist
    for env in (dict(WORLD_SIZE='16', LOCAL_WORLD_SIZE='8'), dict(WORLD_SIZE='8', LOCAL_WORLD_SIZE='8', GROUP_WORLD_SIZE='2'),
                dict(WORLD_SIZE='8', NNODES='2')):
        for k in ('WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_WORLD_SIZE', 'NNODES'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv('RANK', '1')
        monkeypatch.setenv('HOMONIM_AMD_DIST_BACKEND', 'host')
        with pytest.raises(RuntimeError, match='spans nodes'):
            dist.init()
    # one node, as torchrun describes it: accepted by the check
    for k in ('LOCAL_WORLD_SIZE', 'GROUP_WORLD_SIZE', 'NNODES'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8'), monkeypatch.setenv('GROUP_WORLD_SIZE', '1')
    dist._check_one_node(8)


def test_a_launcher_that_names_no_launch_is_named_by_the_ranks_parent(monkeypatch):
    """ mpirun / srun set RANK and WORLD_SIZE only: the rendezvous file is then named after the process that started the ranks, so
    that two such launches of one user do not adopt each other's ranks (round-5 advisor finding). """
    from homonim_amd import dist
    for k in ('MASTER_PORT', 'TORCHELASTIC_RUN_ID', 'HOMONIM_AMD_LAUNCH_ID'):
        monkeypatch.delenv(k, raising=False)
    assert dist._launch_token() == f'ppid_{os.getppid()}'
    monkeypatch.setenv('MASTER_PORT', '29500')
    assert dist._launch_token().startswith('29500')
