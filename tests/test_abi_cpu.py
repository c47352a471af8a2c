"""
CPU tests of the boundary: the C-ABI library builds, loads and exports every symbol include/homonim_hk.h declares
(no compute calls -- there is no GPU here), the host-side classes validate like the reference, and the product never
imports the oracle.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO
from homonim_amd import (Affine, ConfigWarning, CRS, DeviceError, KernelModel, Model, RasterArray, RefSpaceModel,
                         Resampling, _hk, utils)


def _header(which='both'):
    """ Text of the public header(s): include/homonim_hk.h (the drop-in boundary) and include/homonim_hk_devtools.h (aids). """
    names = {'main': ['homonim_hk.h'], 'devtools': ['homonim_hk_devtools.h'], 'both': ['homonim_hk.h', 'homonim_hk_devtools.h']}
    return '\n'.join(open(os.path.join(REPO, 'include', n)).read() for n in names[which])


@pytest.fixture(scope='module')
def lib():
    from homonim_amd import build
    build.build_hip(verbose=False)
    return _hk.load_library()


def test_library_exports_every_declared_symbol(lib):
    strip = lambda text: re.sub(r'/\*.*?\*/', '', text, flags=re.S)   # noqa: E731
    declared = set(re.findall(r'\b(hk_[a-z0-9_]+)\s*\(', strip(_header('both'))))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in the headers but not exported'
    # and the ctypes table covers the headers exactly
    assert declared == set(_hk.SIGNATURES)
    # the aids live in their own header: the boundary header declares none of them, the devtools header nothing else
    aids = set(re.findall(r'\b(hk_[a-z0-9_]+)\s*\(', strip(_header('devtools'))))
    assert aids == set(_hk.DEVTOOLS)
    assert not aids & set(re.findall(r'\b(hk_[a-z0-9_]+)\s*\(', strip(_header('main'))))
    assert lib.hk_abi_version() == _hk.ABI_VERSION == int(re.search(r'#define HK_ABI_VERSION (\d+)', _header('main')).group(1))


def test_header_is_plain_c_and_a_c_program_can_drive_the_library(lib, tmp_path):
    """ The boundary is a C ABI: include/homonim_hk.h compiles as strict C99 and a C program (tests/c/abi_consumer.c)
    dlopens the library, checks struct layout and error behaviour -- and, without a GPU, that contexts are refused. """
    import subprocess
    exe = tmp_path / 'abi_consumer'
    subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(REPO, 'include'),
                    os.path.join(REPO, 'tests', 'c', 'abi_consumer.c'), '-o', str(exe), '-ldl'], check=True)
    run = subprocess.run([str(exe), _hk.lib_path()], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert 'abi_consumer: ok' in run.stdout
    # the aids' header is plain C as well
    src = tmp_path / 'devtools.c'
    src.write_text('#include "homonim_hk_devtools.h"\nint main(void) { return HK_ABI_VERSION > 0 ? 0 : 1; }\n')
    subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(REPO, 'include'), str(src), '-o',
                    str(tmp_path / 'devtools')], check=True)


def test_ctypes_structs_have_the_offsets_the_c_compiler_gives_the_header(tmp_path):
    """ Field by field: the ctypes mirrors in homonim_amd/_hk.py against offsetof() / sizeof() of include/homonim_hk.h as gcc
    lays the structs out (same field names on both sides; a renamed, re-ordered or re-typed field fails here, not on the GPU). """
    import subprocess
    pairs = [('hk_fit_desc', _hk.FitDesc), ('hk_io_desc', _hk.IoDesc), ('hk_space_desc', _hk.SpaceDesc),
             ('hk_out_window', _hk.OutWindow), ('hk_dev_job', _hk.DevJob)]
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "homonim_hk.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'    printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for field in cls._fields_:
            lines.append(f'    printf("{cname} {field[0]} %zu %zu\\n", offsetof({cname}, {field[0]}), '
                         f'sizeof((({cname}*)0)->{field[0]}));')
    lines += ['    return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines) + '\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(REPO, 'include'), str(src), '-o', str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split('\n')
    seen = {}
    for ln in out:
        parts = ln.split()
        if len(parts) == 3:
            seen[(parts[0], 'sizeof')] = int(parts[2])
        elif len(parts) == 4:
            seen[(parts[0], parts[1])] = (int(parts[2]), int(parts[3]))
    for cname, cls in pairs:
        assert ctypes.sizeof(cls) == seen[(cname, 'sizeof')], cname
        for field in cls._fields_:
            desc = getattr(cls, field[0])
            assert (desc.offset, desc.size) == seen[(cname, field[0])], f'{cname}.{field[0]}'
    # the header declares no field the mirrors lack
    header = _header('both')
    for cname, cls in pairs:
        body = re.search(r'typedef struct[^{]*\{([^}]*)\}\s*' + cname + ';', header).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        names = set()
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(','):
                names.add(re.sub(r'\[.*\]', '', part.strip().split()[-1]).lstrip('*'))
        assert names == {f[0] for f in cls._fields_}, cname


def test_ctypes_prototypes_agree_with_the_header_argument_by_argument():
    """ Every prototype of include/homonim_hk.h against its entry in _hk.SIGNATURES: number of arguments, and per argument
    pointer / 4-byte int / 8-byte int / float / double (a struct pointer: the mirrored struct).  A scalar passed at the wrong
    width through ctypes is silent on x86-64 until it is not. """
    header = re.sub(r'/\*.*?\*/', '', _header('both'), flags=re.S)
    protos = re.findall(r'\b([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(hk_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', header)
    assert len(protos) == len(_hk.SIGNATURES)
    scalar = {'int': 'i4', 'int32_t': 'i4', 'uint32_t': 'i4', 'int64_t': 'i8', 'uint64_t': 'i8', 'size_t': 'i8',
              'float': 'f4', 'double': 'f8'}
    structs = {'hk_fit_desc': _hk.FitDesc, 'hk_io_desc': _hk.IoDesc, 'hk_space_desc': _hk.SpaceDesc,
               'hk_out_window': _hk.OutWindow, 'hk_dev_job': _hk.DevJob}

    def c_kind(decl):
        decl = decl.strip()
        if '*' in decl or '[' in decl:   # (an array parameter is a pointer)
            base = decl.replace('const', ' ').replace('struct', ' ').split('*')[0].split()[0]
            return ('ptr', structs.get(base))
        words = [w for w in decl.replace('const', ' ').split()]
        return (scalar[words[0]], None)

    def py_kind(t):
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, 'contents'):
            pointee = getattr(t, '_type_', None)
            return ('ptr', pointee if isinstance(pointee, type) and issubclass(pointee, ctypes.Structure) else None)
        return ({ctypes.c_int: 'i4', ctypes.c_int32: 'i4', ctypes.c_uint32: 'i4', ctypes.c_int64: 'i8', ctypes.c_uint64: 'i8',
                 ctypes.c_size_t: 'i8', ctypes.c_float: 'f4', ctypes.c_double: 'f8'}[t], None)

    for ret, name, args in protos:
        restype, argtypes = _hk.SIGNATURES[name]
        assert py_kind(restype)[0] == ('ptr' if '*' in ret else scalar[ret.split()[-1]]), name
        decls = [] if args.strip() in ('', 'void') else args.split(',')
        assert len(decls) == len(argtypes), f'{name}: {len(decls)} arguments in the header, {len(argtypes)} in _hk.SIGNATURES'
        for i, (decl, t) in enumerate(zip(decls, argtypes)):
            ck, pk = c_kind(decl), py_kind(t)
            assert ck[0] == pk[0], f'{name} argument {i} ({decl.strip()}): header {ck[0]}, ctypes {pk[0]}'
            if ck[1] is not None and pk[1] is not None:
                assert ck[1] is pk[1], f'{name} argument {i}: {decl.strip()} mirrored by {pk[1].__name__}'


def test_the_binding_shown_in_integration_md_matches_the_library():
    """ INTEGRATION.md shows the ctypes stub a homonim maintainer would add; its struct and its argtypes lines are executed
    here and compared with the library's own table (same widths argument by argument, same struct layout). """
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    block = next(b for b in re.findall(r'```python\n(.*?)```', text, flags=re.S) if 'class _FitDesc' in b)
    struct_src = re.search(r'(class _FitDesc\(C\.Structure\):.*?\n)\n', block, flags=re.S).group(1)
    ns = {'C': ctypes}
    exec(struct_src, ns)
    exec(re.search(r'(_f32p, _f64p = .*)\n', block).group(1), ns)
    shown = ns['_FitDesc']
    assert [(n, t) for n, t in shown._fields_] == [(n, t) for n, t in _hk.FitDesc._fields_]

    class _Fn:
        pass

    class _Lib:
        def __getattr__(self, name):
            fn = self.__dict__.setdefault(name, _Fn())
            return fn

    ns['_lib'] = _Lib()
    stmts = re.findall(r'(_lib\.hk_[a-z_]+\.argtypes = \[.*?\])\n', block, flags=re.S)
    assert len(stmts) >= 2
    for stmt in stmts:
        exec(stmt, ns)

    def width(t):
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, 'contents'):
            return 'ptr'
        return ctypes.sizeof(t), t in (ctypes.c_float, ctypes.c_double)

    for name, fn in ns['_lib'].__dict__.items():
        mine = _hk.SIGNATURES[name][1]
        assert len(fn.argtypes) == len(mine), name
        for i, (a, b) in enumerate(zip(fn.argtypes, mine)):
            assert width(a) == width(b), f'{name} argument {i}'


def test_backend_name_and_struct_layout(lib):
    assert lib.hk_backend_name() == b'hip-gfx950'
    assert ctypes.sizeof(_hk.FitDesc) == 40  # 10 x 4-byte fields, matches hk_fit_desc
    assert ctypes.sizeof(_hk.DevJob) == 8 * 8 + 3 * 4 + 4 + 2 * 8 + 2 * 4 + 4 * 4 + 2 * 8   # ... + the store window + scratch
    assert ctypes.sizeof(_hk.OutWindow) == 2 * 8 + 4 * 4 + 8   # ... + the parameter rasters' own row stride


def test_host_only_size_helpers(lib):
    """ The two size helpers of the header are plain host arithmetic (no device needed): scratch of a device job = 5 bytes
    per element of the span of its planes; exchange buffer of the split-block statistics = 4 x 2048 float64 per band. """
    assert lib.hk_dev_job_scratch_bytes(1, 96, 320, 0) == 5 * 96 * 320
    assert lib.hk_dev_job_scratch_bytes(3, 96, 320, 40000) == 5 * (2 * 40000 + 96 * 320)
    assert lib.hk_dev_job_scratch_bytes(0, 96, 320, 0) == 0
    assert lib.hk_block_norm_split_exchange_doubles(1) == 4 * 2048
    assert lib.hk_block_norm_split_exchange_doubles(8) == 8 * 4 * 2048
    assert lib.hk_block_norm_split_exchange_doubles(0) == 0


def test_no_gpu_fails_loudly(lib):
    """ On a GPU-less host the context cannot be created: DeviceError, never a silent CPU path. """
    if _hk.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(DeviceError):
        _hk.Context(0)
    ra = RasterArray(np.ones((8, 8), np.float32), CRS(), Affine.identity())
    with pytest.raises(DeviceError):
        KernelModel(Model.gain, (3, 3)).fit(ra, ra.copy())


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(REPO, 'homonim_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(root, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'hk_oracle' not in text.replace(
                    'build_oracle', '').replace('libhk_oracle', '').replace("'hk_oracle.c'", ''), f


def test_product_never_imports_the_harness_diagnostics():
    """ harness/ (abort tracer, first-GPU-process probe) serves tests/conftest.py, bench.py and smoke() only. """
    for root, _, files in os.walk(os.path.join(REPO, 'homonim_amd')):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(root, f)).read()
                assert 'import harness' not in text and 'from harness' not in text, f
    from harness import abort_trace
    assert os.path.exists(abort_trace.build())   # plain C, gcc only


# -- reference tests/test_kernel_model.py:296-334 (error behaviour, config) -------------------------------------------
@pytest.mark.parametrize('model, kernel_shape', [
    (Model.gain, (0, 0)), (Model.gain_blk_offset, (0, 0)), (Model.gain_offset, (4, 5)), (Model.gain_offset, (1, 1)),
])
def test_kernel_shape_exception(model, kernel_shape):
    with pytest.raises(ValueError):
        RefSpaceModel(model=model, kernel_shape=kernel_shape)


def test_small_gain_offset_kernel_warns():
    with pytest.warns(ConfigWarning):
        KernelModel(Model.gain_offset, (3, 3))


def test_config():
    config = dict(r2_inpaint_thresh=0.1, mask_partial=True, downsampling=Resampling.bilinear,
                  upsampling=Resampling.nearest)
    km = RefSpaceModel(Model.gain, (5, 5), find_r2=True, **config)
    for key, val in config.items():
        assert getattr(km, '_' + key) == val
    assert KernelModel.create_config() == dict(r2_inpaint_thresh=0.25, mask_partial=False,
                                               downsampling=Resampling.average, upsampling=Resampling.cubic_spline)
    assert KernelModel.default_kernel_shape == (5, 5) and KernelModel.default_model == Model.gain_blk_offset
    assert km.model == Model.gain and km.kernel_shape == (5, 5) and km.find_r2


def test_config_exception():
    with pytest.raises(TypeError):
        RefSpaceModel(Model.gain, (5, 5), find_r2=True, unknown='value')


def test_fit_apply_shape_mismatch_is_value_error():
    a = RasterArray(np.ones((8, 8), np.float32), CRS(), Affine.identity())
    b = RasterArray(np.ones((8, 9), np.float32), CRS(), Affine.identity())
    c = RasterArray(np.ones((8, 8), np.float32), CRS(), Affine.translation(1, 0))
    km = KernelModel(Model.gain, (3, 3))
    for other in (b, c):
        with pytest.raises(ValueError):
            km.fit(a, other)
        with pytest.raises(ValueError):
            km.apply(a, other)


def test_utils():
    assert utils.overlap_for_kernel((5, 5)) == (3, 3) and utils.overlap_for_kernel((15, 15)) == (8, 8)
    assert utils.overlap_for_kernel((1, 3)) == (1, 2)
    assert utils.validate_kernel_shape((5, 7), Model.gain) == (5, 7)
    assert utils.validate_threads(0) >= 1
    with pytest.raises(ValueError):
        utils.validate_threads(10**6)
    assert utils.nan_equals(np.array([np.nan, 1., 2.]), np.nan).tolist() == [True, False, False]


# -- RasterArray carrier (reference tests/test_raster_array.py semantics for the in-memory part) -----------------------
def test_raster_array_mask_and_nodata():
    arr = np.arange(12, dtype=np.float32).reshape(3, 4)
    arr[0, 0] = np.nan
    ra = RasterArray(arr.copy(), CRS(), Affine.identity())
    assert ra.shape == (3, 4) and ra.count == 1 and ra.dtype == 'float32' and np.isnan(ra.nodata)
    assert ra.mask.sum() == 11 and not ra.mask[0, 0]
    ra.nodata = 5  # re-label the masked pixel
    assert ra.array[0, 0] == 5 and ra.mask.sum() == 10  # the pixel that already equalled 5 is now masked too
    ra.nodata = None
    assert ra.mask.all()
    ra2 = RasterArray(arr.copy(), CRS(), Affine.identity(), nodata=3)
    m = ra2.mask.copy()
    m[2, 2] = False
    ra2.mask = m
    assert ra2.array[2, 2] == 3 and ra2.mask.sum() == 10  # nan pixel is valid under numeric nodata
    cp = ra2.copy()
    cp.array[1, 1] = -1
    assert ra2.array[1, 1] != -1
    with pytest.raises(ValueError):
        ra2.array = np.zeros((2, 2), np.float32)
    with pytest.raises(ValueError):
        RasterArray(np.zeros(3, np.float32), CRS(), Affine.identity())
    with pytest.raises(TypeError):
        RasterArray(arr, 'EPSG:3857', Affine.identity())
    with pytest.raises(TypeError):
        RasterArray(arr, CRS(), (1, 0, 0, 0, 1, 0))


def test_raster_array_from_profile():
    prof = dict(crs=CRS(), transform=Affine.identity(), nodata=float('nan'), count=3, width=4, height=2, dtype='float32')
    ra = RasterArray.from_profile(None, prof)
    assert ra.array.shape == (3, 2, 4) and np.isnan(ra.array).all() and ra.count == 3
    assert ra.profile['count'] == 3 and ra.proj_profile['shape'] == (2, 4)
    from homonim_amd.errors import ImageProfileError
    with pytest.raises(ImageProfileError):
        RasterArray.from_profile(None, dict(crs=CRS(), transform=Affine.identity()))


def test_counts_pending_is_host_only(lib):
    """ hk_counts_pending: callers of the two-halves r2-mask protocol need not interpret the raw counters (the retry bit of the
    certificate-only build included) -- and it runs without a device. """
    import ctypes as C
    u64p = C.POINTER(C.c_uint64)
    a = np.zeros(5, np.uint64)
    assert lib.hk_counts_pending(a.ctypes.data_as(u64p), 5) == 0
    a[3] = 17
    assert lib.hk_counts_pending(a.ctypes.data_as(u64p), 5) == 1 and lib.hk_counts_pending(a.ctypes.data_as(u64p), 3) == 0
    a[3], a[0] = 0, 1 << 63          # HK_COUNT_RETRY: the band must be run again with the complete build
    assert lib.hk_counts_pending(a.ctypes.data_as(u64p), 5) == 1
    assert lib.hk_counts_pending(None, 5) == 0


def test_every_entry_point_of_the_boundary_header_has_a_row_in_integration_md():
    """ include/homonim_hk.h holds the drop-in boundary only: each of its entry points is named in INTEGRATION.md's table of what
    it replaces in the reference (`hk_event_*` covers a family, `hk_host_alloc/free` a pair); the aids of
    include/homonim_hk_devtools.h are named there as aids. """
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    named, families = set(), []
    for tok in re.findall(r'hk_[a-z0-9_]+\*?(?:/[a-z0-9_]+)*', text):
        head, *alts = tok.split('/')
        if head.endswith('*'):
            families.append(head[:-1])
            continue
        named.add(head)
        prefix = head[:head.rfind('_') + 1]
        named.update(prefix + a for a in alts)
    strip = lambda t: re.sub(r'/\*.*?\*/', '', t, flags=re.S)   # noqa: E731
    declared = set(re.findall(r'\b(hk_[a-z0-9_]+)\s*\(', strip(_header('main'))))
    missing = sorted(n for n in declared if n not in named and not any(n.startswith(f) for f in families))
    assert not missing, f'entry points of homonim_hk.h without a row in INTEGRATION.md: {missing}'
    assert all(n in named for n in _hk.DEVTOOLS)
