""" Register / scratch / LDS figures of every kernel in a HIP object or library, from the code object's metadata notes.

    python tools/kernel_regs.py [OBJECT ...] [--spills] [--grep TEXT]      (default: homonim_amd/lib/hk_fit_m*.o)

Lists the builds that use scratch memory (private_segment_fixed_size > 0 or vgpr_spill_count > 0) with --spills. """
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def kernels(obj):
    obj = os.path.abspath(obj)
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', obj], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [f for f in os.listdir(tmp) if 'gfx950' in f]
        # llvm-objdump writes the bundles next to the INPUT on some versions
        src_dir = os.path.dirname(obj)
        bundles = [f for f in os.listdir(src_dir) if f.startswith(os.path.basename(obj) + '.0.')]   # (some versions write them beside the INPUT)
        stray = [f for f in bundles if 'gfx950' in f]
        path = os.path.join(tmp, cos[0]) if cos else os.path.join(src_dir, stray[0])
        notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', path], capture_output=True, text=True).stdout
        for f in bundles:
            os.unlink(os.path.join(src_dir, f))
    out = []
    for e in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
        g = lambda key: int(re.search(r'\.' + key + r':\s+(\d+)', e).group(1))  # noqa: E731
        out.append(dict(name=re.search(r'\.name:\s+(\S+)', e).group(1), vgpr=g('vgpr_count'), sgpr=g('sgpr_count'),
                        scratch=g('private_segment_fixed_size'), spills=g('vgpr_spill_count'), lds=g('group_segment_fixed_size')))
    return out


def demangle(names):
    res = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    return res[:len(names)]


if __name__ == '__main__':
    argv = list(sys.argv[1:])
    grep_text = None
    if '--grep' in argv:
        i = argv.index('--grep')
        grep_text = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith('--')]
    lib_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'homonim_amd', 'lib')
    # default: the translation units of the fused kernel (homonim_amd/build.py FIT_TUS)
    objs = args if args else sorted(os.path.join(lib_dir, f) for f in os.listdir(lib_dir) if f.startswith('hk_fit_m') and f.endswith('.o'))
    ks = [k for obj in objs for k in kernels(obj)]
    grep = grep_text
    names = demangle([k['name'] for k in ks])
    fit = [(k, n) for k, n in zip(ks, names) if 'fit_apply_kernel' in n or 'fit_list_kernel' in n]
    bad = [(k, n) for k, n in fit if k['scratch'] or k['spills']]
    print(f'{len(ks)} kernels, {len(fit)} fused-kernel builds, {len(bad)} of them use scratch')
    for k, n in (bad if '--spills' in sys.argv else zip(ks, names)):
        if grep and grep not in n:
            continue
        print(f"{n.replace('hk::', '').split('(')[0]:70s} vgpr {k['vgpr']:3d} sgpr {k['sgpr']:3d} scratch {k['scratch']:4d} B spilled {k['spills']:3d}")
