"""
Build script for the native pieces (no cmake needed; hipcc cross-compiles gfx950 without a GPU).

    python -m homonim_amd.build            # libhomonim_hk.so (HIP, gfx950) + oracle/_build/libhk_oracle.so (gcc)

The HIP library is the product; the oracle library is test infrastructure (see oracle/README in DESIGN.md).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB_DIR = os.path.join(PKG, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libhomonim_hk.so')
HIP_SOURCES = ['hk_kernels.hip', 'hk_norm.hip', 'hk_convert.hip', 'hk_mask.hip', 'hk_inpaint.hip', 'hk_resample.hip', 'hk_compare.hip', 'hk_api.hip']
# The fused fit(+apply) kernel (template: csrc/hk_fit_kernel.h) is instantiated in one translation unit per MODEL x R2: the same
# source compiled six times side by side (object name, extra flags).  The heaviest first, so that it does not start last.
FIT_TUS = [(f'hk_fit_m{m}_r{r}.o', [f'-DHK_TU_MODEL={m}', f'-DHK_TU_R2={r}']) for m, r in ((2, 1), (0, 1), (1, 1), (2, 0), (0, 0), (1, 0))]
FIT_TU_SOURCE = 'hk_fit_tu.hip'
# -ffp-contract=off: numpy never fuses a*b+c; the kernels must round exactly where the reference does.
# -Wno-bitwise-instead-of-logical: `a | b` / `a & b` on booleans is deliberate in the kernels (no short-circuit branch per pixel).
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function',
               '-Wno-bitwise-instead-of-logical']

ORACLE_DIR = os.path.join(REPO, 'oracle')
ORACLE_BUILD = os.path.join(ORACLE_DIR, '_build')
ORACLE_LIB = os.path.join(ORACLE_BUILD, 'libhk_oracle.so')


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'build step failed: {" ".join(cmd)}\n{res.stdout}')
    if res.stdout.strip():   # warnings are not swallowed: the build is meant to be warning-free
        print(res.stdout, flush=True)
    return res.stdout


def build_hip(force: bool = False, verbose: bool = True) -> str:
    """ Compile homonim_amd/csrc/*.hip for gfx950 into homonim_amd/lib/libhomonim_hk.so (in-tree). """
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, 'hk_kernels.h'), os.path.join(CSRC, 'hk_fit_kernel.h'), os.path.join(REPO, 'include', 'homonim_hk.h'),
               os.path.join(REPO, 'include', 'homonim_hk_devtools.h')]
    objs = []
    jobs = []
    extra = os.environ.get('HK_EXTRA_HIPCC_FLAGS', '').split()   # development: e.g. -DHK_DEV_SUBSET
    for obj_name, flags in FIT_TUS:
        src_path = os.path.join(CSRC, FIT_TU_SOURCE)
        obj = os.path.join(LIB_DIR, obj_name)
        objs.append(obj)
        if force or _stale(obj, [src_path] + headers):
            jobs.append([_hipcc(), *HIPCC_FLAGS, *extra, *flags, '-c', src_path, '-o', obj])
    for src in HIP_SOURCES:
        src_path = os.path.join(CSRC, src)
        obj = os.path.join(LIB_DIR, src.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src_path] + headers):
            jobs.append([_hipcc(), *HIPCC_FLAGS, '-c', src_path, '-o', obj])
    if jobs:
        if verbose:
            print(f'[homonim_amd.build] compiling {len(jobs)} HIP source(s) for gfx950 ...', flush=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        _run([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH, *objs])
        if verbose:
            print(f'[homonim_amd.build] linked {LIB_PATH}', flush=True)
    return LIB_PATH


def build_oracle(force: bool = False, verbose: bool = True) -> str:
    """ Compile the plain-C oracle (test infrastructure; never loaded by the product). """
    src = os.path.join(ORACLE_DIR, 'hk_oracle.c')
    if not os.path.exists(src):
        return ''
    os.makedirs(ORACLE_BUILD, exist_ok=True)
    if force or _stale(ORACLE_LIB, [src]):
        _run(['gcc', '-O2', '-std=c11', '-fPIC', '-shared', '-fopenmp', '-ffp-contract=off', '-fno-fast-math',
              '-o', ORACLE_LIB, src, '-lm'])
        if verbose:
            print(f'[homonim_amd.build] built {ORACLE_LIB}', flush=True)
    return ORACLE_LIB


if __name__ == '__main__':
    force = '--force' in sys.argv
    build_hip(force)
    build_oracle(force)
