""" What the compiled kernels must NOT contain, checked in the gfx950 code objects of the in-tree build (no GPU needed).

Round 6 found that the batched builds of the fused kernel (BASELINE configs[3]) and the statistics' streaming pass had run on
``flat_load`` / ``flat_store`` since round 3: their plane pointers are read from tables in device memory, which gives the compiler
no address space to infer, and a flat access also counts on the LDS counter -- a wait for the row ring then waits for the global
stream too (HISTORY.md 71).  Nothing in the test-suite could see that: results are the same.  This test disassembles every object
of ``homonim_amd/lib`` and fails on any ``flat_`` or ``scratch_`` instruction outside the self-test kernel (no generic-pointer
accesses, no register spills), so that the next table pointer without an address space is caught at build time. """
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, 'homonim_amd', 'lib')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
# kernels that may use scratch: the self-test (a diagnostic kernel that indexes local arrays dynamically; never on a hot path)
ALLOWED = ('selftest_kernel',)


def _device_disassembly(obj):
    """ disassembly of the gfx950 code object bundled in a host object file """
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, '--offloading', local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        cos = [f for f in os.listdir(tmp) if 'gfx950' in f]
        if not cos:
            return None
        return subprocess.run([OBJDUMP, '-d', os.path.join(tmp, cos[0])], capture_output=True, text=True, check=True).stdout


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain is not installed')
def test_no_flat_or_scratch_instruction_in_any_kernel():
    objs = sorted(glob.glob(os.path.join(LIB, '*.o')))
    if not objs:
        pytest.skip('no in-tree objects (python -m homonim_amd.build leaves them beside the library)')
    offenders = {}
    n_kernels = 0
    for obj in objs:
        text = _device_disassembly(obj)
        if text is None:      # a host-only object (hk_api.o holds no kernels)
            continue
        kernel = None
        for line in text.splitlines():
            m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
            if m:
                kernel = m.group(1)
                n_kernels += 1
                continue
            if kernel is None or any(a in kernel for a in ALLOWED):
                continue
            m = re.match(r'^\s+((?:flat|scratch)_\w+)', line)
            if m:
                offenders.setdefault((os.path.basename(obj), kernel), []).append(m.group(1))
    assert n_kernels > 300, f'only {n_kernels} kernels found: the disassembly did not work'
    report = '\n'.join(f'{o}: {k}: {len(v)} x {sorted(set(v))}' for (o, k), v in sorted(offenders.items()))
    assert not offenders, ('generic-pointer (flat_) or spilling (scratch_) instructions in the library -- a pointer read from a table '
                           'in device memory needs its address space (hk_fit_kernel.h table_pointer, hk_norm.hip plane_at):\n' + report)
