"""
GPU tests of the optional collective of the hot path: the block statistics of a gain-blk-offset block whose rows are spread
over ranks / devices (homonim_amd/split_norm.py, hk_block_norm_split_dev; reference KernelModel._fit_block_norm,
homonim/kernel_model.py:216-229).  The split result must be the single-device result: order statistics exactly, the std
ratio up to the order of the float64 sums.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

from homonim_amd import _hk, split_norm  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402


@pytest.fixture(scope='module')
def ctx():
    c = _hk.default_context()
    c.selftest()
    return c


def _block(variant, h=613, w=1003, nb=3):
    pairs = [onp.synth_pair(h, w, 700 + b, variant) for b in range(nb)]
    return np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])


def _single(ctx, src, ref, nd):
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, nd, nd)
    return np.stack([ctx.block_norm(desc, src[b], ref[b]) for b in range(src.shape[0])])


def _slab_job(c, src, ref, r0, r1, stream=0):
    nb, _, w = src.shape
    rows = r1 - r0
    stride = (w + 63) // 64 * 64
    pad = lambda a: np.ascontiguousarray(np.pad(a[:, r0:r1], ((0, 0), (0, 0), (0, stride - w))), np.float32)  # noqa: E731
    d_src, d_ref = c.dev_alloc(4 * stride * max(rows, 1) * nb), c.dev_alloc(4 * stride * max(rows, 1) * nb)
    c.h2d(d_src, pad(src)), c.h2d(d_ref, pad(ref))
    job = _hk.DevJob()
    job.src, job.ref = d_src, d_ref
    job.corr = job.gain = job.offset = job.r2 = job.norm = job.fail_count = None
    job.n_bands, job.height, job.width, job.stride, job.band_stride = nb, rows, w, stride, stride * max(rows, 1)
    job.seg_rows, job.stream = 0, stream
    return job, (d_src, d_ref)


@pytest.mark.parametrize('variant, edges', [('frame+holes', (0, 300, 613)), ('none', (0, 37, 38, 400, 613)),
                                            ('frame+holes', (0, 3, 613)), ('frame+holes', (0, 200, 420, 613))])
@pytest.mark.oracle
def test_split_statistics_in_one_process_equal_the_whole_block(ctx, variant, edges):
    """ two to four contexts on this GPU, each with a slab of rows (one of them a single row / an all-nodata strip): the
    phases run in step, the exchange buffers are summed on the host. """
    src, ref = _block(variant)
    nd = None if variant == 'none' else np.nan
    exp = _single(ctx, src, ref, nd)
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, nd, nd)
    ctxs = [_hk.Context(0, n_streams=1) for _ in range(len(edges) - 1)]
    parts, bufs = [], []
    try:
        for c, r0, r1 in zip(ctxs, edges[:-1], edges[1:]):
            job, b = _slab_job(c, src, ref, r0, r1)
            parts.append((c, job)), bufs.append((c, b))
        norms = split_norm.block_norm_split_local(parts, desc)
        for n in norms[1:]:
            assert (n == norms[0]).all()                       # identical on every "rank"
        np.testing.assert_allclose(norms[0], exp, rtol=1e-12, atol=0)
        # the percentile term is exact: with the single-device gain the offsets agree to the last bit or two
        assert np.abs(norms[0][:, 1] - exp[:, 1]).max() <= 1e-12 * np.abs(exp[:, 1]).max() + 1e-15
    finally:
        for c, (a, b) in bufs:
            c.dev_free(a), c.dev_free(b)
        for c in ctxs:
            c.close()


@pytest.mark.oracle
def test_split_statistics_of_a_block_without_valid_pixels(ctx):
    src, ref = _block('frame+holes', 64, 200, 2)
    src[:] = np.nan
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, np.nan, np.nan)
    c2 = _hk.Context(0, n_streams=1)
    j0, b0 = _slab_job(ctx, src, ref, 0, 30)
    j1, b1 = _slab_job(c2, src, ref, 30, 64)
    try:
        norms = split_norm.block_norm_split_local([(ctx, j0), (c2, j1)], desc)
        assert (norms[0] == 0).all() and (norms[1] == 0).all()   # kernel_model.py:223-226
    finally:
        ctx.dev_free(b0[0]), ctx.dev_free(b0[1]), c2.dev_free(b1[0]), c2.dev_free(b1[1])
        c2.close()


def _run_worker(tmp_path, nproc, variant, env_extra):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', PYTHONPATH=REPO, **env_extra)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr',
           '127.0.0.1', '--master-port', str(29100 + (os.getpid() + nproc) % 150),
           os.path.join(REPO, 'tests', '_split_norm_worker.py'), str(tmp_path), variant]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    return (tmp_path / 'backend.txt').read_text().split()


@pytest.mark.parametrize('nproc', [2, 3])
def test_split_statistics_over_a_process_group(ctx, tmp_path, nproc):
    """ one process per slab under torch.distributed.run; on this 1-GPU box the ranks share device 0 and all-reduce
    through gloo (the exchange buffer goes through the host), on a full node the same worker runs over RCCL. """
    backend, world = _run_worker(tmp_path, nproc, 'frame+holes', dict(HOMONIM_AMD_DIST_BACKEND='gloo'))
    assert backend == 'gloo' and int(world) == nproc
    src, ref = _block('frame+holes')
    exp = _single(ctx, src, ref, np.nan)
    norms = [np.load(tmp_path / f'norm_{r}.npy') for r in range(nproc)]
    for n in norms[1:]:
        assert (n == norms[0]).all()
    np.testing.assert_allclose(norms[0], exp, rtol=1e-12, atol=0)


def _run_comm_workers(tmp_path, world, variant='frame+holes'):
    """ `world` plain processes, rank r on GPU r, joined through the library's own RCCL communicator (no torch) """
    procs = []
    for r in range(world):
        env = dict(os.environ, PYTHONPATH=REPO, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r))
        env.pop('HOMONIM_AMD_COMM_FILE', None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, 'tests', '_split_norm_comm_worker.py'), str(tmp_path), variant],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    results = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        results.append((p.returncode, err))
    if world > 1 and any(rc == 77 for rc, _ in results):
        # the communicator could not be formed on this box (RCCL bootstrap / IPC of the environment): nothing was computed
        pytest.skip('RCCL communicator over several GPUs could not be initialised here: ' + results[0][1][-300:])
    for rc, err in results:
        assert rc == 0, err[-3000:]
    return [np.load(tmp_path / f'norm_{r}.npy') for r in range(world)]


def test_split_statistics_through_rccl(ctx, tmp_path):
    """ The production path: the LIBRARY binds RCCL (hk_comm_init) and queues the five all-reduces between the six phases on
    the job's stream itself (hk_block_norm_split_comm_dev) -- no torch in the process (the worker asserts it), no host
    synchronisation between the phases.  A communicator of one rank on this 1-GPU box: the collectives really run through
    RCCL and the result must be the whole-block statistics of the single slab. """
    norms = _run_comm_workers(tmp_path, 1)
    src, ref = _block('frame+holes')
    exp = _single(ctx, src, ref, np.nan)
    np.testing.assert_allclose(norms[0], exp, rtol=1e-12, atol=0)


@pytest.mark.parametrize('world', [2, 3, 4])
def test_split_statistics_through_rccl_over_several_gpus(ctx, tmp_path, world):
    """ ... and with one process per GPU where the box has them (the driver's 8-GPU node; skipped on a 1-GPU box: RCCL does
    not put two ranks on one device).  With three or more ranks the last one holds no rows of the block. """
    if _hk.device_count() < world:
        pytest.skip(f'needs {world} GPUs')
    norms = _run_comm_workers(tmp_path, world)
    for n in norms[1:]:
        assert (n == norms[0]).all()
    src, ref = _block('frame+holes')
    exp = _single(ctx, src, ref, np.nan)
    np.testing.assert_allclose(norms[0], exp, rtol=1e-12, atol=0)


@pytest.mark.oracle
def test_split_statistics_with_a_slab_of_no_rows(ctx):
    """ a rank without rows of the block takes part with zeros (it used to be refused, leaving the others
    waiting inside the all-reduce) """
    src, ref = _block('frame+holes')
    exp = _single(ctx, src, ref, np.nan)
    desc = _hk.make_desc('gain-blk-offset', (5, 5), False, None, np.nan, np.nan)
    ctxs = [_hk.Context(0, n_streams=1) for _ in range(3)]
    parts, bufs = [], []
    try:
        for c, r0, r1 in zip(ctxs, (0, 250, 250), (250, 250, 613)):
            job, b = _slab_job(c, src, ref, r0, r1)
            parts.append((c, job)), bufs.append((c, b))
        norms = split_norm.block_norm_split_local(parts, desc)
        assert (norms[1] == norms[0]).all() and (norms[2] == norms[0]).all()
        np.testing.assert_allclose(norms[0], exp, rtol=1e-12, atol=0)
    finally:
        for c, (a, b) in bufs:
            c.dev_free(a), c.dev_free(b)
        for c in ctxs:
            c.close()


def test_split_statistics_through_a_torch_nccl_group_of_one(ctx, tmp_path):
    """ the host-driven phase loop with the all-reduce on a torch tensor through RCCL (backend nccl) -- a group of one on this
    box: same plumbing, and the result must then be the whole-block statistics of the single slab. """
    env = dict(HOMONIM_AMD_DIST_FORCE='1')
    os.environ.pop('HOMONIM_AMD_DIST_BACKEND', None)
    backend, world = _run_worker(tmp_path, 1, 'frame+holes', env)
    assert backend == 'nccl' and int(world) == 1
    src, ref = _block('frame+holes')
    exp = _single(ctx, src, ref, np.nan)
    np.testing.assert_allclose(np.load(tmp_path / 'norm_0.npy'), exp, rtol=1e-12, atol=0)
