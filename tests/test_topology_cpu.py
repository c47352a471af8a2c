""" Host placement (homonim_amd/topology.py): the sysfs parsers on a canned tree of a two-socket, eight-GPU node. """
import os
import sys

import pytest

from homonim_amd import topology

GPUS = {  # PCI address -> NUMA node (the layout of an 8 x MI355X node: four GPUs per socket)
    '0000:05:00.0': 0, '0000:15:00.0': 0, '0000:65:00.0': 0, '0000:75:00.0': 0,
    '0000:85:00.0': 1, '0000:95:00.0': 1, '0000:e5:00.0': 1, '0000:f5:00.0': 1,
}


@pytest.fixture
def sysfs(tmp_path):
    root = tmp_path / 'sys'
    for node, cpus in ((0, '0-63,128-191\n'), (1, '64-127,192-255\n')):
        d = root / 'devices' / 'system' / 'node' / f'node{node}'
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(cpus)
    for i, (bdf, node) in enumerate(GPUS.items()):
        d = root / 'bus' / 'pci' / 'devices' / bdf
        d.mkdir(parents=True)
        (d / 'numa_node').write_text(f'{node}\n')
        (d / 'uevent').write_text(f'DRIVER=amdgpu\nPCI_CLASS=38000\nPCI_SLOT_NAME={bdf}\n')
        c = root / 'class' / 'drm' / f'card{i}'
        c.mkdir(parents=True)
        os.symlink(d, c / 'device')
        (root / 'class' / 'drm' / f'card{i}-DP-1').mkdir()   # connectors are not cards
    # a device whose node the firmware does not name, and an integrated VGA controller without the file
    d = root / 'bus' / 'pci' / 'devices' / '0000:03:00.0'
    d.mkdir(parents=True)
    (d / 'numa_node').write_text('-1\n')
    (root / 'bus' / 'pci' / 'devices' / '0000:04:00.0').mkdir()
    return str(root)


def test_cpulist():
    assert topology.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert topology.parse_cpulist('5') == [5]
    assert topology.parse_cpulist('') == []
    assert topology.parse_cpulist('3,1-2,2') == [1, 2, 3]


def test_gpu_node_and_cpus(sysfs):
    assert topology.pci_numa_node('0000:05:00.0', sysfs) == 0
    assert topology.pci_numa_node('0000:E5:00.0', sysfs) == 1     # HIP spells the address in upper case
    assert topology.pci_numa_node('0000:03:00.0', sysfs) is None  # -1
    assert topology.pci_numa_node('0000:04:00.0', sysfs) is None  # no file
    assert topology.pci_numa_node('0000:aa:00.0', sysfs) is None  # no device
    cpus = topology.node_cpus(1, sysfs)
    assert len(cpus) == 128 and cpus[0] == 64 and cpus[63] == 127 and cpus[64] == 192 and cpus[-1] == 255
    assert topology.node_cpus(7, sysfs) == []


def test_placement_respects_the_allowed_cpus(sysfs):
    p = topology.placement_for('0000:95:00.0', sysfs)
    assert p['numa_node'] == 1 and len(p['cpus']) == 128
    p = topology.placement_for('0000:95:00.0', sysfs, allowed=list(range(60, 70)))
    assert p['cpus'] == [64, 65, 66, 67, 68, 69]
    p = topology.placement_for('0000:95:00.0', sysfs, allowed=[0, 1])   # a container pinned to the other socket
    assert p['cpus'] == []
    assert topology.placement_for('0000:03:00.0', sysfs) == dict(bus_id='0000:03:00.0', numa_node=None, cpus=[])


def test_cards_and_summary(sysfs):
    cards = topology.drm_cards(sysfs)
    assert cards == GPUS
    rec = dict(topology.placement_for('0000:05:00.0', sysfs), bound=True, threads=3, reason=None)
    s = topology.summary(rec)
    assert s['cpus'] == '0-63,128-191' and s['n_cpus'] == 128 and s['numa_node'] == 0 and s['bound'] and s['threads'] == 3


def test_bind_without_a_library_or_gpu_is_a_record_not_an_error(monkeypatch):
    monkeypatch.setenv('HOMONIM_AMD_NO_BIND', '1')
    assert topology.bind_to_device(0)['bound'] is False
    monkeypatch.delenv('HOMONIM_AMD_NO_BIND')
    before = os.sched_getaffinity(0)
    rec = topology.bind_to_device(0, sysfs_root='/nonexistent')   # no GPU here: the record says why, the affinity stays
    assert rec['bound'] is False and rec['reason']
    assert os.sched_getaffinity(0) == before


def test_a_worker_thread_binds_to_its_device_and_remembers_it(monkeypatch):
    """ Several GPUs in one process (RasterFuse with a device list): a worker thread is bound per block to the CPUs of the block's GPU;
    the look-up happens once per device, a thread already on the device's node makes no system call. """
    import threading
    calls = []
    allowed = sorted(os.sched_getaffinity(0))
    monkeypatch.setattr(topology, '_device_cpus', {0: allowed[:1], 1: allowed[-1:], 2: []})
    monkeypatch.setattr(topology, '_thread_device', None)
    real = os.sched_setaffinity
    monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, cpus: (calls.append((pid, tuple(cpus))), real(pid, cpus))[1])
    out = {}

    def worker():
        out['a'] = topology.bind_current_thread(0), sorted(os.sched_getaffinity(0))
        out['b'] = topology.bind_current_thread(0)                      # already there: no call
        out['c'] = topology.bind_current_thread(1), sorted(os.sched_getaffinity(0))
        out['d'] = topology.bind_current_thread(2)                      # a device whose node sysfs does not name: left alone
    before = os.sched_getaffinity(0)
    t = threading.Thread(target=worker)
    t.start(), t.join()
    assert out['a'] == (True, allowed[:1]) and out['b'] is True and out['c'] == (True, allowed[-1:]) and out['d'] is False
    assert calls == [(0, tuple(allowed[:1])), (0, tuple(allowed[-1:]))]
    assert os.sched_getaffinity(0) == before      # the calling (main) thread is untouched
    monkeypatch.setenv('HOMONIM_AMD_NO_BIND', '1')
    assert topology.bind_current_thread(0) is False


def test_a_worker_bound_to_one_gpu_still_finds_the_cpus_of_another(monkeypatch, sysfs):
    """ RasterFuse deals blocks round the models: a pool worker already bound to GPU A's node can be the first to ask for GPU B.  The
    CPUs of B's node are taken from the PROCESS's baseline affinity, not from the calling thread's current one (which holds none of
    them), and an empty answer is not remembered (round-5 advisor finding). """
    import threading
    import types
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip('needs two CPUs')
    lo, hi = allowed[:1], allowed[-1:]
    monkeypatch.setattr(topology, '_device_cpus', {})
    monkeypatch.setattr(topology, '_thread_device', None)
    monkeypatch.setattr(topology, '_BASELINE', allowed)
    # devices 0 / 1 on two "nodes" made of the first / the last allowed CPU
    monkeypatch.setattr(topology, 'placement_for', lambda bus, root='/sys', allowed_=None: dict(
        bus_id=bus, numa_node=int(bus[-1]), cpus=[c for c in (lo if bus.endswith('0') else hi) if allowed_ is None or c in allowed_]))
    fake_hk = types.SimpleNamespace(device_pci_bus_id=lambda d: f'0000:00:00.{d}')
    import homonim_amd
    monkeypatch.setattr(homonim_amd, '_hk', fake_hk, raising=False)
    monkeypatch.setitem(sys.modules, 'homonim_amd._hk', fake_hk)
    out = {}

    def worker():
        out['a'] = topology.bind_current_thread(0), sorted(os.sched_getaffinity(0))
        out['b'] = topology.bind_current_thread(1), sorted(os.sched_getaffinity(0))   # asked from a thread that holds only `lo`
    t = threading.Thread(target=worker)
    t.start(), t.join()
    assert out['a'] == (True, lo) and out['b'] == (True, hi), out
