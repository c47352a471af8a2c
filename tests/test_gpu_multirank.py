"""
GPU tests of the multi-process / multi-context paths: one process per GPU under torch.distributed.run, every rank its
shard of the (band x block) list, no data-path collective.  On a 1-GPU box the ranks share
device 0 (HOMONIM_AMD_DIST_BACKEND=host); on a full node the same worker runs one rank per GPU and joins the library's RCCL
communicator.  The ranks meet over loopback TCP (homonim_amd/dist.py): no tensor library on either path.
"""
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

from homonim_amd import _hk  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402


def _inputs():
    pairs = [onp.synth_pair(520, 700, 300 + b, 'frame+holes') for b in range(3)]
    return np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])


def _same(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize('model, k, contiguous', [('gain-offset', 5, False), ('gain-blk-offset', 5, True)])
def test_two_ranks_process_their_shards_and_the_union_is_the_single_rank_result(tmp_path, model, k, contiguous):
    from homonim_amd.fuse import RasterFuse
    env = dict(os.environ, HOMONIM_AMD_DIST_BACKEND='host', MASTER_ADDR='127.0.0.1', PYTHONPATH=REPO)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(29600 + (os.getpid() + k + contiguous) % 300), os.path.join(REPO, 'tests', '_rank_worker.py'),
           str(tmp_path), model, str(k), '1' if contiguous else '0']
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    src, ref = _inputs()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        full_c, full_p = RasterFuse(src, ref).process(None, model, (k, k), param_filename=True,
                                                      model_config=dict(r2_inpaint_thresh=0.25),
                                                      block_config=dict(threads=2, max_block_mem=0.3))
    parts_c = [np.load(tmp_path / f'corr_{r}.npy') for r in range(2)]
    parts_p = [np.load(tmp_path / f'params_{r}.npy') for r in range(2)]
    filled = [~np.isnan(p) for p in parts_c]
    assert filled[0].any() and filled[1].any() and not (filled[0] & filled[1]).any()      # disjoint, both non-empty
    assert _same(np.where(filled[0], parts_c[0], parts_c[1]), full_c)                     # and complete
    fp = [~np.isnan(p) for p in parts_p]
    assert not (fp[0] & fp[1]).any() and _same(np.where(fp[0], parts_p[0], parts_p[1]), full_p)
    world, total, slowest = (int(v) for v in (tmp_path / 'summary.txt').read_text().split())
    assert world == 2 and total == int((~np.isnan(full_c)).sum()) and slowest == 1


def test_two_contexts_in_one_process_share_the_block_list(ctx_unused=None):
    """ device_config(devices=[d0, d1]): one model + context per entry, blocks dealt round the models by the thread pool
    (fuse.py process()).  With one GPU both entries are device 0 -- two separate contexts (stream pools, staging slabs,
    certificate / in-painting state) working on one raster at once -- and the result must not change. """
    from homonim_amd.fuse import RasterFuse
    src, ref = _inputs()
    kw = dict(model='gain-offset', kernel_shape=(5, 5), param_filename=True, model_config=dict(r2_inpaint_thresh=0.6),
              block_config=dict(threads=4, max_block_mem=0.3))
    n_dev = _hk.device_count()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        one_c, one_p = RasterFuse(src, ref).process(device_config=dict(devices=[0]), **kw)
        two_c, two_p = RasterFuse(src, ref).process(device_config=dict(devices=[0, 1 % n_dev], separate_contexts=True), **kw)
    assert _same(one_c, two_c) and _same(one_p, two_p)
    assert np.isnan(one_c).sum() < one_c.size


def test_bench_joins_an_rccl_group_of_one():
    """ The driver's N > 1 launch is `python -m torch.distributed.run ... bench.py --gpus N` over RCCL.  A 1-GPU box can
    still run that plumbing with a launch of one rank: the rank sets the rendezvous up alone, joins the library's RCCL
    communicator (a communicator of one) and all-reduces a word on it. """
    import json
    env = dict(os.environ, HOMONIM_AMD_DIST_FORCE='1', MASTER_ADDR='127.0.0.1', PYTHONPATH=REPO)
    env.pop('HOMONIM_AMD_DIST_BACKEND', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(29300 + os.getpid() % 200), os.path.join(REPO, 'bench.py'), '--gpus', '1', '--config', '1',
           '--steps', '3', '--warmup', '1', '--no-cpu-baseline']
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['parity_spot_check']['passed']
    assert line.get('dist_backend') == 'rccl'
    assert line.get('rccl_ranks') == 1   # the library's own communicator (hk_comm_info) stands and its all-reduce ran


@pytest.mark.parametrize('config', [1, 3])
def test_bench_with_two_ranks(config):
    """ `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` as the driver launches it, the two ranks sharing
    this box's GPU (backend 'host'): config 1 = one raster per rank (weak scaling, the value counts both), config 3 = the block
    positions split between the ranks (strong scaling).  Rank 0 prints the one JSON line. """
    import json
    env = dict(os.environ, HOMONIM_AMD_DIST_BACKEND='host', MASTER_ADDR='127.0.0.1', PYTHONPATH=REPO)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(29700 + (os.getpid() + config) % 200), os.path.join(REPO, 'bench.py'), '--gpus', '2',
           '--config', str(config), '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-end-to-end']
    if config == 3:
        cmd += ['--size', '8192', '--bands', '2']
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['parity_spot_check']['passed']
    assert line['scaling'] == ('weak' if config == 1 else 'strong') and line['dist_backend'] == 'host'


def test_bench_refuses_a_rank_count_that_disagrees_with_the_launch():
    """ A launcher that started ONE rank (WORLD_SIZE=1) for a command line that names two: refused before any GPU call. """
    run = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1'],
                         env=dict(os.environ, PYTHONPATH=REPO, WORLD_SIZE='1'), capture_output=True, text=True, timeout=120)
    assert run.returncode == 2 and 'torch.distributed.run' in run.stderr


def test_bench_launches_its_own_ranks():
    """ Plain `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset): the parent process -- which never touches a GPU --
    starts the two ranks as child processes with the torchrun environment and relays rank 0's ONE JSON line
    (homonim/fuse.py:394-408: one call fans out over the workers).  The ranks share this box's GPU (backend 'host'). """
    import json
    env = dict(os.environ, HOMONIM_AMD_DIST_BACKEND='host', PYTHONPATH=REPO)
    for key in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(key, None)
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--config', '1', '--steps', '3', '--warmup', '1',
           '--no-cpu-baseline']
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['parity_spot_check']['passed']
    assert line['scaling'] == 'weak' and line['dist_backend'] == 'host' and line['rccl_ranks'] is None


@pytest.mark.parametrize('config', [3, 4])
def test_bench_as_rank_and_the_single_gpu_projection(config):
    """ `bench.py --as-rank R/N` runs exactly the shard rank R of an N-rank launch would run, alone (no group, no peers), and the plain
    single-rank line of configs 3 / 4 carries `projected_scaling_single_gpu`: every shard of an N = 2 / 4 / 8 launch timed alone --
    labelled a projection, never `value`, never a scaling claim.  (homonim/fuse.py:394-408, raster_pair.py:379-428: the work items the
    ranks split.) """
    import json
    env = dict(os.environ, PYTHONPATH=REPO)
    for key in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(key, None)
    small = ['--size', '8192', '--bands', '2'] if config == 3 else ['--size', '1024', '--tiles', '16']
    base = [sys.executable, os.path.join(REPO, 'bench.py'), '--config', str(config), '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
            '--no-end-to-end'] + small
    run = subprocess.run(base, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads([ln for ln in run.stdout.strip().splitlines() if ln.startswith('{')][-1])
    proj = line['projected_scaling_single_gpu']
    assert 'PROJECTION' in proj['kind'] and set(proj['by_world_size']) == {'2', '4', '8'}
    for n, rec in proj['by_world_size'].items():
        assert len(rec['shard_ms']) == int(n) and rec['max_shard_ms'] == max(rec['shard_ms'])
        assert rec['projected_speedup'] == pytest.approx(proj['t1_ms'] / rec['max_shard_ms'], rel=1e-3)
    assert line['scaling'] == 'strong' and line['n_gpus'] == 1 and 'host_placement' in line
    # one rank of four, alone
    run = subprocess.run(base + ['--as-rank', '1/4'], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    solo = json.loads([ln for ln in run.stdout.strip().splitlines() if ln.startswith('{')][-1])
    assert solo['config']['as_rank']['rank'] == 1 and solo['config']['as_rank']['of'] == 4
    assert 'alone' in solo['scaling'] and 'projected_scaling_single_gpu' not in solo
    assert solo['parity_spot_check'] is None or solo['parity_spot_check']['passed']
    # its rate is what THIS GPU did on a quarter of the work, not a whole-job figure
    whole = line['value'] * line['ms_per_step']
    assert solo['value'] * solo['ms_per_step'] == pytest.approx(whole / 4, rel=0.3)
    # and the argument is refused where it has no meaning
    bad = subprocess.run(base[:2] + ['--config', '2', '--as-rank', '0/2'], env=env, capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2


def test_rank_binds_to_the_numa_node_of_its_gpu():
    """ homonim_amd/topology.py on the box: the library names the GPU's PCI address, sysfs its NUMA node; when sysfs says, every thread
    of the process ends up on that node's CPUs (a box that does not say is left alone, with the reason in the record). """
    from homonim_amd import topology
    before = os.sched_getaffinity(0)
    try:
        bus = _hk.device_pci_bus_id(0)
        assert len(bus) >= 12 and bus == bus.lower() and bus.count(':') == 2
        rec = topology.bind_to_device(0)
        assert rec['bus_id'] == bus
        if rec['bound']:
            assert rec['numa_node'] is not None and set(os.sched_getaffinity(0)) == set(rec['cpus']) and rec['threads'] >= 1
            for tid in os.listdir('/proc/self/task'):
                assert set(os.sched_getaffinity(int(tid))) <= set(rec['cpus'])
        else:
            assert rec['reason']
        assert topology.summary(rec)['bus_id'] == bus
    finally:
        for tid in os.listdir('/proc/self/task'):
            try:
                os.sched_setaffinity(int(tid), before)
            except OSError:
                pass


# ----------------------------------------------------------------------------------------------------------------------
# round 6: the 8-rank dress rehearsal.  No 8-GPU node has been available to this project; the first scaling run will be a one-shot on a
# machine nobody has touched.  Eight processes sharing this box's GPU (backend 'host') run everything of that launch but RCCL and a
# second HBM stack: eight contexts, eight pinned rings, eight NUMA bindings, the rendezvous, the shard lists, the one JSON line.
# (homonim/fuse.py:394-408: the pool the ranks replace; raster_pair.py:379-428: the work items they split.)
def _bench(n, launcher, args, extra_env=None, timeout=900):
    import json
    import time
    env = dict(os.environ, HOMONIM_AMD_DIST_BACKEND='host', PYTHONPATH=REPO, **(extra_env or {}))
    for key in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'LOCAL_WORLD_SIZE', 'GROUP_WORLD_SIZE'):
        env.pop(key, None)
    common = ['--gpus', str(n), '--no-cpu-baseline', '--no-end-to-end', '--no-nan-variant', '--no-other-configs', '--no-power-probe',
              '--no-projection', '--checksum'] + args
    if launcher == 'torchrun' and n > 1:   # the driver's launch
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
               '--master-port', str(29100 + (os.getpid() + 7 * n + len(args)) % 400), os.path.join(REPO, 'bench.py')] + common
        env['MASTER_ADDR'] = '127.0.0.1'
    else:                                   # plain: bench.py starts its own ranks
        cmd = [sys.executable, os.path.join(REPO, 'bench.py')] + common
    t0 = time.time()
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in run.stdout.strip().splitlines() if ln.startswith('{')]
    return run, [json.loads(ln) for ln in lines], time.time() - t0


@pytest.mark.parametrize('config, launcher', [(1, 'plain'), (2, 'torchrun'), (3, 'torchrun'), (4, 'plain')])
def test_eight_ranks_dress_rehearsal(config, launcher):
    """ `bench.py --gpus 8` at the BASELINE sizes, both ways it can be launched: configs 1 / 2 hold one full raster per rank (8 x 12.9 GB
    of this GPU's 288 for config 2), configs 3 / 4 split the real lists (the reference's 128 blocks of 8 x 16384^2; 64 tiles of
    4 x 4096^2).  One JSON line from rank 0, n_gpus 8, parity spot check green, every rank's host placement in the line -- and the
    UNION of the eight shards is the single-rank result: exact checksum and pixel count of every corrected pixel (configs 3 / 4: the
    totals; configs 1 / 2: rank 0's raster, the ranks' rasters differ by their seed). """
    args = ['--config', str(config), '--steps', '2', '--warmup', '1']
    run, lines, _ = _bench(8, launcher, args)
    assert run.returncode == 0, run.stderr[-3000:]
    assert len(lines) == 1, run.stdout[-2000:]
    line = lines[0]
    assert line['n_gpus'] == 8 and line['value'] > 0 and line['dist_backend'] == 'host'
    assert line['parity_spot_check']['passed']
    assert line['scaling'] == ('weak' if config in (1, 2) else 'strong')
    ranks = line['host_placement_ranks']
    assert len(ranks) == 8 and all(r['bus_id'] == ranks[0]['bus_id'] for r in ranks)      # (one GPU here; eight on the node)
    ck = line['shard_checksum']
    assert len(ck['per_rank']) == 8 and all(p['pixels'] > 0 for p in ck['per_rank'])
    solo_run, solo_lines, _ = _bench(1, 'plain', args)
    assert solo_run.returncode == 0, solo_run.stderr[-3000:]
    solo = solo_lines[-1]['shard_checksum']
    if config in (1, 2):
        assert ck['per_rank'][0] == solo['per_rank'][0]
        assert ck['pixels'] == 8 * solo['pixels']
    else:
        assert ck['pixels'] == solo['pixels'] and ck['sum'] == solo['sum'], 'the union of the eight shards is not the single-rank result'


@pytest.mark.parametrize('launcher', ['plain', 'torchrun'])
def test_a_rank_that_dies_ends_the_launch(launcher):
    """ A rank that dies mid-run (HK_BENCH_DIE_RANK: gone without a word once the ranks have met) must end the whole launch with a
    non-zero exit code within seconds -- its peers see the closed socket at their next exchange -- not leave seven processes in a
    barrier until the 900 s rendezvous timeout. """
    run, lines, seconds = _bench(8, launcher, ['--config', '1', '--steps', '2', '--warmup', '1'], extra_env={'HK_BENCH_DIE_RANK': '5'},
                                 timeout=300)
    assert run.returncode != 0, (run.stdout[-1000:], run.stderr[-2000:])
    assert seconds < 150, f'the launch took {seconds:.0f} s to notice a dead rank'
    assert 'HK_BENCH_DIE_RANK' in run.stderr
