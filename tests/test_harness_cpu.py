""" The harness diagnostics (harness/, never imported by the product): the abort tracer names a fatal signal raised on a NATIVE
thread -- sender, thread, native frames -- and rescues what the process last wrote to a captured stderr.  This is the tool that
showed the round-3 abort to be a GPU memory access fault (profiles/r04_abort_caught.txt). """
import os
import subprocess
import sys

from conftest import REPO

_CHILD = r'''
import ctypes, os, sys, tempfile, time
sys.path.insert(0, %(repo)r)
from harness import abort_trace
real = os.dup(2)
cap = tempfile.TemporaryFile()
os.dup2(cap.fileno(), 2)                      # what pytest's fd capture does during a test
assert abort_trace.install(fd=real)
os.write(2, b"Memory access fault by GPU node-2 (stand-in for the runtime's last words)\n")
libc = ctypes.CDLL(None)
t = ctypes.c_ulong()
libc.pthread_create(ctypes.byref(t), None, ctypes.c_void_p(ctypes.cast(libc.abort, ctypes.c_void_p).value), None)
time.sleep(5)
'''


def test_abort_on_a_native_thread_is_named_and_the_captured_stderr_rescued():
    run = subprocess.run([sys.executable, '-c', _CHILD % dict(repo=REPO)], capture_output=True, text=True, timeout=60)
    assert run.returncode == -6, run.returncode            # SIGABRT, default action after the report
    err = run.stderr
    assert '[hk_abort_trace] signal 6' in err and 'raised by this process (abort/raise)' in err
    assert 'abort+0x' in err or 'abort' in err              # native frames of the aborting thread
    assert 'tail of the captured stderr' in err and "stand-in for the runtime's last words" in err


def test_first_process_probe_reports_without_a_gpu():
    """ harness/first_process.py as conftest / bench.py use it: a child process; without a GPU it says so and exits 0 (with one it works for two seconds). """
    from harness import first_process
    res = first_process.run(timeout=120)
    assert res['rc'] == 0 and ('no GPU' in res['output'] or 'host-pointer calls as the first GPU process' in res['output'])


def test_a_dying_probe_fails_the_run(tmp_path):
    """ A green run means no process died: with the probe switched on (HK_FIRST_PROCESS_PROBE=1) and made to die the way round 3's
    first GPU processes did, a pytest session whose own tests all pass ends non-zero, and bench.py's gate says `fatal`. """
    test = tmp_path / 'test_ok.py'
    test.write_text('import pytest\n\n@pytest.mark.gpu\ndef test_ok():\n    assert True\n')
    (tmp_path / 'conftest.py').write_text(open(os.path.join(REPO, 'tests', 'conftest.py')).read())
    os.makedirs(tmp_path / 'golden', exist_ok=True)
    env = dict(os.environ, HK_FIRST_PROCESS_PROBE='1', HK_FIRST_PROCESS_TEST_DIE='1', PYTHONPATH=REPO)
    # (the copied conftest resolves REPO from its own location: point it back at the repository's golden vectors)
    src = (tmp_path / 'conftest.py').read_text().replace(
        "REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", f"REPO = {REPO!r}")
    (tmp_path / 'conftest.py').write_text(src)
    run = subprocess.run([sys.executable, '-m', 'pytest', str(test), '-q', '-m', 'gpu', '-p', 'no:cacheprovider'], cwd=tmp_path, env=env,
                         capture_output=True, text=True, timeout=300)
    assert '1 passed' in run.stdout, run.stdout + run.stderr
    assert run.returncode != 0, run.stdout + run.stderr
    assert 'this run FAILS' in run.stderr
    # the same session with a probe that lives: green
    env.pop('HK_FIRST_PROCESS_TEST_DIE')
    run = subprocess.run([sys.executable, '-m', 'pytest', str(test), '-q', '-m', 'gpu', '-p', 'no:cacheprovider'], cwd=tmp_path, env=env,
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    # bench.py's side of it
    code = 'import sys; sys.path.insert(0, %r); from harness import first_process as f; g = f.gate(); sys.exit(3 if g and g["fatal"] else 0)' % REPO
    env['HK_FIRST_PROCESS_TEST_DIE'] = '1'
    assert subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, timeout=300).returncode == 3
    env.pop('HK_FIRST_PROCESS_PROBE')
    assert subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, timeout=300).returncode == 0   # off: nothing runs
