"""
CPU tests of the split-block statistics PROTOCOL (homonim_amd/split_norm.py; the device side is hk_norm.hip
launch_block_norm_split, tested on the GPU in tests/test_gpu_split_norm.py): the numpy statement of what every rank
contributes per phase and of the select on the reduced histograms (oracle/oracle_np.py split_norm_*) must give the block
statistics of the reference (KernelModel._fit_block_norm, homonim/kernel_model.py:216-229) -- in one process over several
slabs, and with two ranks that all-reduce their contributions over gloo.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from oracle import oracle_np as onp


def _protocol(vals_per_rank, allreduce):
    """ the six phases with `allreduce` standing for the SUM all-reduce of a numpy array (every rank's copy in, sums out) """
    world = len(vals_per_rank)
    shift = allreduce([np.array([v[0].mean() if v[0].size else 0.0, v[1].mean() if v[1].size else 0.0]) for v in vals_per_rank]) / world
    mom = allreduce([onp.split_norm_moments(v, shift) for v in vals_per_rank])
    n = int(mom[0])
    if n == 0:
        return np.zeros(2), None
    ranks, prefixes = onp.split_norm_ranks(n), [[0, 0], [0, 0]]
    for level in range(3):
        hist = allreduce([onp.split_norm_hist(v, level, prefixes) for v in vals_per_rank])
        prefixes, ranks = onp.split_norm_select(hist, level, prefixes, ranks)
    return onp.split_norm_finish(mom, shift, prefixes), prefixes


@pytest.mark.parametrize('edges', [(0, 200), (0, 70, 71, 200), (0, 3, 100, 150, 200)])
@pytest.mark.parametrize('kind', ['continuous', 'ties'])
def test_protocol_over_slabs_equals_the_whole_block(edges, kind):
    rng = np.random.default_rng(len(edges))
    if kind == 'continuous':
        src, ref = onp.synth_pair(200, 300, 3, 'frame+holes')
    else:
        src = rng.integers(0, 7, (200, 300)).astype(np.float32)
        ref = (2 * src + rng.integers(0, 3, src.shape)).astype(np.float32)
        src[rng.random(src.shape) < 0.05] = np.nan
    vals = [onp.split_norm_slab_values(src[a:b], np.nan, ref[a:b], np.nan) for a, b in zip(edges[:-1], edges[1:])]
    norm, prefixes = _protocol(vals, lambda parts: np.sum(parts, axis=0))
    exp = onp.fit_block_norm(src, np.nan, ref, np.nan)
    assert norm == pytest.approx(exp, rel=2e-6, abs=1e-7)       # numpy's own statistics run in float32
    # the order statistics are EXACT: the selected keys are elements k0, k0 + 1 of the sorted valid values
    s_all, r_all = onp.split_norm_slab_values(src, np.nan, ref, np.nan)
    k0, k1 = onp.split_norm_ranks(s_all.size)[0]
    for q, allv in enumerate((s_all, r_all)):
        srt = np.sort(allv)
        assert onp._key2f(prefixes[q][0]) == srt[k0] and onp._key2f(prefixes[q][1]) == srt[k1]


def test_protocol_without_valid_pixels():
    src = np.full((20, 30), np.nan, np.float32)
    ref = np.ones((20, 30), np.float32)
    vals = [onp.split_norm_slab_values(src[:9], np.nan, ref[:9], np.nan), onp.split_norm_slab_values(src[9:], np.nan, ref[9:], np.nan)]
    norm, _ = _protocol(vals, lambda parts: np.sum(parts, axis=0))
    assert (norm == 0).all()     # kernel_model.py:223-226


_WORKER = r'''
import os, sys, json
sys.path.insert(0, {repo!r})
sys.path.insert(0, os.path.join({repo!r}, 'tests'))
import numpy as np, torch, torch.distributed as dist
from homonim_amd import dist as hdist
from oracle import oracle_np as onp
from test_split_norm_cpu import _protocol
rank, world, _ = hdist.init()          # the product's rendezvous (loopback TCP)
dist.init_process_group(backend='gloo')  # ... and torch's gloo group for this test's all-reduce
src, ref = onp.synth_pair(200, 300, 3, 'frame+holes')
edges = [0, 83, 200]
mine = onp.split_norm_slab_values(src[edges[rank]:edges[rank + 1]], np.nan, ref[edges[rank]:edges[rank + 1]], np.nan)

def allreduce(parts):           # this rank holds ONE part; the sum comes from the collective
    t = torch.from_numpy(np.ascontiguousarray(parts[0], np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()

# _protocol divides the shift by len(vals_per_rank): pass a list of the world's length whose own entry is first
norm, prefixes = _protocol([mine] + [mine] * (world - 1), lambda parts: allreduce(parts))
with open(os.path.join({out!r}, 'rank_%d.json' % rank), 'w') as f:
    json.dump(dict(rank=rank, norm=[float(v) for v in norm], prefixes=[[int(x) for x in row] for row in prefixes]), f)
hdist.finalize()
dist.destroy_process_group()
'''


@pytest.mark.timeout(180)
def test_protocol_with_two_gloo_ranks(tmp_path):
    """ world_size 2 over gloo: each rank contributes its slab, the all-reduce is torch.distributed's. """
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER.format(repo=REPO, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', HOMONIM_AMD_DIST_BACKEND='gloo')
    res = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
         '--master-port', '29633', str(script)], capture_output=True, text=True, env=env, timeout=170)
    assert res.returncode == 0, res.stderr[-2000:]
    outs = [json.loads((tmp_path / f'rank_{r}.json').read_text()) for r in range(2)]
    assert len(outs) == 2 and outs[0]['norm'] == outs[1]['norm'] and outs[0]['prefixes'] == outs[1]['prefixes']
    src, ref = onp.synth_pair(200, 300, 3, 'frame+holes')
    assert outs[0]['norm'] == pytest.approx(list(onp.fit_block_norm(src, np.nan, ref, np.nan)), rel=2e-6, abs=1e-7)
