""" pytest configuration: markers + shared golden-vector loaders. """
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'oracle: the test compares the results of the kernels it launches with the oracle, a golden '
                                       'vector or a reference-held number (tests/test_zz_build_ledger.py counts only these)')
    # a fatal signal names its sender, thread and native frames (harness/abort_trace.py); in front of faulthandler
    try:
        from harness import abort_trace
        log_dir = os.path.join(REPO, 'gpurun_out')
        path = os.path.join(log_dir, f'abort_trace_{os.getpid()}.log') if os.path.isdir(log_dir) else None
        abort_trace.install(path)
    except Exception as ex:  # diagnostics must never stop a run
        print(f'[conftest] abort tracer not installed: {ex}', file=sys.stderr)


# the library checks every page of every caller array it is about to copy directly (hk_api.hip check_rows_pinned): a host pointer
# that is not page-locked fails its call in the test-suite instead of being pinned on the fly by the runtime
os.environ.setdefault('HK_ASSERT_PINNED', '1')

_probe = {}


def pytest_sessionstart(session):
    """ HK_FIRST_PROCESS_PROBE=1: a run that selects the GPU tests first lets a CHILD process be the lease's first GPU process
    (harness/first_process.py: round 3's aborts only ever hit first processes; the cause was removed in round 4 and the probe is
    off by default since round 5).  When it runs, its death is the run's failure: pytest_sessionfinish turns the exit status
    non-zero -- a green run means no process died. """
    expr = session.config.getoption('markexpr', '') or ''
    if 'gpu' not in expr or 'not gpu' in expr or os.environ.get('HK_FIRST_PROCESS_PROBE') != '1':
        return
    try:
        from harness import first_process
        _probe.update(first_process.run())
    except Exception as ex:
        _probe.update(rc=None, seconds=0, output=f'probe not run: {ex}')
    if _probe.get('rc') == 0:   # (`-q` prints no report header: say it here)
        last = (_probe.get('output') or '').splitlines()[-1:] or ['']
        sys.__stderr__.write(f"[conftest] first GPU process probe: rc 0 in {_probe.get('seconds')} s -- {last[0]}\n")
        sys.__stderr__.flush()
    else:   # loud, on the real stderr, whatever pytest captures
        sys.__stderr__.write(f"\n[conftest] THE FIRST GPU PROCESS OF THIS RUN DIED OR FAILED (rc {_probe.get('rc')}):\n{_probe.get('output')}\n\n")
        sys.__stderr__.flush()


def pytest_sessionfinish(session, exitstatus):
    if _probe and _probe.get('rc') != 0:
        sys.__stderr__.write(f"[conftest] the first-GPU-process probe ended with rc {_probe.get('rc')}: this run FAILS whatever its tests did\n")
        sys.__stderr__.flush()
        if session.exitstatus == 0:
            session.exitstatus = 3


def pytest_report_header(config):
    if _probe:
        last = (_probe.get('output') or '').splitlines()[-1:] or ['']
        return f"first GPU process probe: rc {_probe.get('rc')} in {_probe.get('seconds')} s -- {last[0]}"
    return None


# ----------------------------------------------------------------------------------------------------------------------
# Launch ledger (round 6).  The library counts the launches of every kernel build it holds (hk_debug_build_ledger: each instantiation
# of the fused kernel's template, each kernel of the other units, launched or not).  Around every GPU test the counts are read; a
# build counts as CHECKED only when it was launched inside a test that (a) carries the `oracle` marker -- its assertions compare the
# launched kernels' results with the oracle, a golden vector or a reference-held number -- and (b) passed.  The last GPU test
# (tests/test_zz_build_ledger.py) fails on any build that never was.  Kernels launched in child processes are not seen (their
# tests are not marked).
BUILD_LEDGER = {'checked': {}, 'unchecked': {}, 'gpu_tests': 0, 'gpu_tests_failed': 0}


@pytest.hookimpl(hookwrapper=True, tryfirst=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    if rep.when == 'call':
        item._hk_call_passed = rep.passed


@pytest.fixture(autouse=True)
def _build_ledger(request):
    node = request.node
    if node.get_closest_marker('gpu') is None:
        yield
        return
    from homonim_amd import _hk
    before = _hk.build_ledger()
    yield
    after = _hk.build_ledger()
    passed = bool(getattr(node, '_hk_call_passed', False))
    BUILD_LEDGER['gpu_tests'] += 1
    BUILD_LEDGER['gpu_tests_failed'] += 0 if passed else 1
    book = BUILD_LEDGER['checked' if (passed and node.get_closest_marker('oracle') is not None) else 'unchecked']
    test_name = node.nodeid.split('::', 1)[-1].split('[')[0]
    for build, n in after.items():
        d = n - before.get(build, 0)
        if d > 0:
            entry = book.setdefault(build, {'launches': 0, 'tests': set()})
            entry['launches'] += d
            entry['tests'].add(test_name)


def _nodata(v):
    return None if v is None else (float('nan') if v == 'nan' else float(v))


def load_golden_cases():
    """ [(case dict with decoded nodata, lazily-indexed npz key prefix)] from tests/golden/manifest.json """
    with open(os.path.join(GOLDEN_DIR, 'manifest.json')) as f:
        manifest = json.load(f)
    cases = []
    for c in manifest['cases']:
        c = dict(c)
        c['src_nodata'] = _nodata(c['src_nodata'])
        c['ref_nodata'] = _nodata(c['ref_nodata'])
        c['kernel_shape'] = tuple(c['kernel_shape'])
        cases.append(c)
    return cases


GOLDEN_CASES = load_golden_cases()


def case_id(c):
    k = c['kernel_shape']
    return f"{c['name']}-{c['model']}-k{k[0]}x{k[1]}-r2{int(c['find_r2'])}-t{c['r2_inpaint_thresh']}-{c['variant']}"


@pytest.fixture(scope='session')
def goldens():
    return np.load(os.path.join(GOLDEN_DIR, 'kernel_model_goldens.npz'))


def assert_same_f32(actual, expected, what=''):
    """ Bit-exact float32 comparison with NaN == NaN (any payload). """
    actual = np.asarray(actual)
    expected = np.asarray(expected)
    assert actual.shape == expected.shape, f'{what}: shape {actual.shape} != {expected.shape}'
    assert actual.dtype == expected.dtype, f'{what}: dtype {actual.dtype} != {expected.dtype}'
    both_nan = np.isnan(actual) & np.isnan(expected)
    same = (actual == expected) | both_nan
    if not same.all():
        idx = np.argwhere(~same)
        i = tuple(idx[0])
        raise AssertionError(
            f'{what}: {idx.shape[0]} of {actual.size} elements differ; first at {i}: {actual[i]!r} != {expected[i]!r}'
        )
