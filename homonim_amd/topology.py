"""
Host placement of a rank: run the process's threads on the cores of the NUMA node its GPU hangs off.

One process drives one GPU (bench.py --gpus N, RasterFuse with a device list per process).  Its host side moves every byte
twice -- pageable caller rasters are packed into the pinned staging ring (a memcpy), the ring is read by the GPU's DMA -- so at
8 ranks x (44 in + 22 out) GB/s a rank whose threads and pinned pages sit on the other socket pays the inter-socket link for
all of it.  `bind_to_device` reads the GPU's PCI address from the library (hk_device_pci_bus_id), its NUMA node and that node's
CPU list from sysfs, and sets the affinity of EVERY thread the process has so far (the HIP runtime started some) -- threads
created later inherit it, and page-locked memory allocated from then on (hipHostMalloc: first touch by the caller) is local.

The reference has nothing to place: its workers are threads of one host process (homonim/fuse.py:396-401).

Everything that reads sysfs takes the tree's root as an argument, so the parsers are tested on a canned tree without a GPU
(tests/test_topology_cpu.py).
"""
import os
from typing import Dict, List, Optional


def parse_cpulist(text: str) -> List[int]:
    """ '0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist). """
    cpus = []
    for part in text.strip().split(','):
        part = part.strip()
        if not part:
            continue
        if '-' in part:
            lo, hi = part.split('-', 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def pci_numa_node(bus_id: str, sysfs_root: str = '/sys') -> Optional[int]:
    """ NUMA node of the PCI device `bus_id` ('0000:c1:00.0'); None when sysfs does not say (file missing, or -1: a
    single-node host or a VM without the ACPI proximity information). """
    try:
        with open(os.path.join(sysfs_root, 'bus', 'pci', 'devices', bus_id.lower(), 'numa_node')) as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def node_cpus(node: int, sysfs_root: str = '/sys') -> List[int]:
    """ CPUs of NUMA node `node` ([] when the node directory is missing). """
    try:
        with open(os.path.join(sysfs_root, 'devices', 'system', 'node', f'node{int(node)}', 'cpulist')) as f:
            return parse_cpulist(f.read())
    except OSError:
        return []


def drm_cards(sysfs_root: str = '/sys') -> Dict[str, Optional[int]]:
    """ {PCI address: NUMA node} of every DRM card sysfs lists (/sys/class/drm/card*/device -> the PCI device; its `uevent`
    names PCI_SLOT_NAME).  Diagnostics: what the box looks like, whichever GPU this rank got. """
    out = {}
    base = os.path.join(sysfs_root, 'class', 'drm')
    try:
        names = sorted(os.listdir(base))
    except OSError:
        return out
    for name in names:
        if not name.startswith('card') or '-' in name:
            continue
        dev = os.path.join(base, name, 'device')
        slot = None
        try:
            with open(os.path.join(dev, 'uevent')) as f:
                for line in f:
                    if line.startswith('PCI_SLOT_NAME='):
                        slot = line.split('=', 1)[1].strip().lower()
        except OSError:
            continue
        if slot is None:
            continue
        try:
            with open(os.path.join(dev, 'numa_node')) as f:
                node = int(f.read().strip())
        except (OSError, ValueError):
            node = -1
        out[slot] = node if node >= 0 else None
    return out


def placement_for(bus_id: str, sysfs_root: str = '/sys', allowed: Optional[List[int]] = None) -> dict:
    """ Where a rank driving the GPU at `bus_id` should run: {'bus_id', 'numa_node', 'cpus'} -- `cpus` is the node's CPU list
    intersected with `allowed` (the affinity the launcher / container gave the process), empty when there is nothing to do
    (unknown node, or no allowed CPU on it). """
    node = pci_numa_node(bus_id, sysfs_root)
    cpus = node_cpus(node, sysfs_root) if node is not None else []
    if allowed is not None:
        cpus = [c for c in cpus if c in set(allowed)]
    return dict(bus_id=bus_id.lower(), numa_node=node, cpus=cpus)


def _threads_of_process() -> List[int]:
    try:
        return sorted(int(t) for t in os.listdir('/proc/self/task'))
    except OSError:
        return [0]


def bind_to_device(device: int, sysfs_root: str = '/sys') -> dict:
    """ Run this process on the cores next to HIP device `device`.  Call it BEFORE page-locked memory is allocated (a
    Context's staging rings come on first use, RasterFuse pins on request).  Returns the placement record
    {'bus_id', 'numa_node', 'cpus', 'bound', 'threads', 'reason'}; never raises: a box that does not say is left alone.
    HOMONIM_AMD_NO_BIND=1 turns it off. """
    rec = dict(bus_id=None, numa_node=None, cpus=[], bound=False, threads=0, reason=None)
    if os.environ.get('HOMONIM_AMD_NO_BIND') == '1':
        rec['reason'] = 'HOMONIM_AMD_NO_BIND=1'
        return rec
    try:
        from homonim_amd import _hk
        bus_id = _hk.device_pci_bus_id(device)
    except Exception as ex:
        rec['reason'] = f'no PCI address for device {device}: {ex}'
        return rec
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = None
    rec.update(placement_for(bus_id, sysfs_root, allowed))
    if not rec['cpus']:
        rec['reason'] = 'sysfs names no NUMA node for the GPU' if rec['numa_node'] is None else 'no allowed CPU on the GPU\'s node'
        return rec
    n = 0
    for tid in _threads_of_process():
        try:
            os.sched_setaffinity(tid, rec['cpus'])
            n += 1
        except OSError:
            pass   # a thread that ended meanwhile
    rec.update(bound=n > 0, threads=n)
    return rec


_device_cpus: Dict[int, List[int]] = {}
_thread_device = None   # threading.local, made on first use


def _process_baseline() -> Optional[List[int]]:
    """ The CPUs the launcher / container allows this PROCESS, read from the main thread (thread-group leader) -- not from the calling
    thread, which an earlier bind_current_thread() may have narrowed to another GPU's node -- and captured when this module is first
    imported, i.e. before any binding of its own (bind_to_device narrows the main thread too). """
    try:
        return sorted(os.sched_getaffinity(os.getpid()))
    except (AttributeError, OSError):
        return None


_BASELINE = _process_baseline()


def bind_current_thread(device: int, sysfs_root: str = '/sys') -> bool:
    """ Run the CALLING thread on the cores next to HIP device `device` (one process driving several GPUs: RasterFuse with a device
    list deals its blocks to worker threads; the thread that packs a block into a GPU's pinned staging ring should run on that GPU's
    socket).  The placement of a device is looked up once; a thread that is already bound to the device costs nothing.  -> whether
    the thread is bound now.  Never raises; HOMONIM_AMD_NO_BIND=1 turns it off. """
    global _thread_device
    import threading
    if os.environ.get('HOMONIM_AMD_NO_BIND') == '1':
        return False
    if _thread_device is None:
        _thread_device = threading.local()
    if getattr(_thread_device, 'device', None) == device:
        return True
    cpus = _device_cpus.get(device)
    if cpus is None:
        # `allowed` = the process's baseline, not this thread's current affinity: a pool worker already bound to GPU A's node
        # that is the first to ask for GPU B would otherwise find no allowed CPU on B's node (round-5 advisor finding)
        try:
            from homonim_amd import _hk
            cpus = placement_for(_hk.device_pci_bus_id(device), sysfs_root, _BASELINE)['cpus']
        except Exception:
            cpus = []
        if cpus:   # an empty answer is not remembered: the next block asks again
            _device_cpus[device] = cpus
    if not cpus:
        return False
    try:
        os.sched_setaffinity(0, cpus)   # 0: the calling thread
    except OSError:
        return False
    _thread_device.device = device
    return True


def summary(rec: dict) -> dict:
    """ The placement record as bench.py prints it: the CPU list as a range string. """
    cpus = rec.get('cpus') or []
    spans, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        spans.append(str(cpus[i]) if i == j else f'{cpus[i]}-{cpus[j]}')
        i = j + 1
    return dict(bus_id=rec.get('bus_id'), numa_node=rec.get('numa_node'), cpus=','.join(spans), n_cpus=len(cpus),
                bound=bool(rec.get('bound')), threads=rec.get('threads', 0), reason=rec.get('reason'))
