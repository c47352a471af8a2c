""" The LAST GPU test (files run in name order): every kernel build the library holds must have been launched by a test that
compared its results with the oracle (tests/conftest.py BUILD_LEDGER; include/homonim_hk_devtools.h hk_debug_build_ledger).

"922 passed" does not say that each of the several hundred builds of the fused kernel's template -- MODEL x R2 x width x nodata
specialisation x ring mode x certificate-only x lock-step x batched, picked by hk_api.hip fill_args from measured thresholds -- ever
met the oracle; round 5 found a bug in shipped builds (kernels taller than 129 rows) only because someone happened to add a case.
This test turns the question into a failure, and writes the ledger (build, launches, checking tests) next to the run's other
outputs.  It only judges a COMPLETE run of the GPU suite: with -k, a file selection or failed tests before it, it skips / reports. """
import os

import pytest

from conftest import BUILD_LEDGER, REPO

pytestmark = pytest.mark.gpu

# measurement / test aids of include/homonim_hk_devtools.h: no reference arithmetic to compare with
AIDS = {'synth_kernel', 'stream_probe_kernel', 'selftest_kernel', 'checksum_kernel'}


def _write_report(path, universe):
    checked, unchecked = BUILD_LEDGER['checked'], BUILD_LEDGER['unchecked']
    rows = []
    for build in sorted(universe):
        c, u = checked.get(build), unchecked.get(build)
        state = 'checked' if c else ('launched-unchecked' if u else 'never-launched')
        tests = sorted(c['tests']) if c else (sorted(u['tests']) if u else [])
        rows.append(f"{build}\t{state}\t{(c or {}).get('launches', 0)}\t{(u or {}).get('launches', 0)}\t{','.join(tests)}")
    with open(path, 'w') as f:
        f.write('# kernel build\tstate\tlaunches in oracle-checking tests\tother launches\ttests\n')
        f.write('\n'.join(rows) + '\n')


def test_every_kernel_build_met_the_oracle(request):
    from homonim_amd import _hk
    universe = set(_hk.build_ledger()) - AIDS
    out_dir = os.path.join(REPO, 'gpurun_out')
    if os.path.isdir(out_dir):
        _write_report(os.path.join(out_dir, 'build_coverage.txt'), universe)
    cfg = request.config
    narrowed = bool(cfg.getoption('keyword')) or any(os.path.isfile(a.split('::')[0]) for a in cfg.args)
    if narrowed:
        pytest.skip('a selection of the GPU suite ran: the ledger only judges complete runs')
    checked = set(BUILD_LEDGER['checked'])
    missing = sorted(universe - checked)
    n_fit = sum(1 for b in universe if b.startswith('fit_apply_kernel'))
    print(f'[ledger] {len(universe & checked)} of {len(universe)} kernel builds ({n_fit} of the fused kernel) were launched by a passing '
          f'test that compares with the oracle; {BUILD_LEDGER["gpu_tests"]} GPU tests, {BUILD_LEDGER["gpu_tests_failed"]} not passed')
    assert not missing, (f'{len(missing)} of {len(universe)} kernel builds never met the oracle (launched only by unmarked / failed '
                         f'tests, or not at all): {missing[:40]}{" ..." if len(missing) > 40 else ""}')
