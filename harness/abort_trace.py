"""
Opt-in diagnostics of the test / bench harness: say who raised a fatal signal.

A GPU fault ends the process with an abort() on a native thread of the HSA runtime; the runtime's one-line message goes
to fd 2, which pytest may have captured, and faulthandler shows Python frames only.  install() loads
harness/_build/libhk_abort_trace.so (harness/hk_abort_trace.c) and puts its handler in front of faulthandler's: signal origin, thread,
native backtrace and the tail of a captured stderr go to a duplicate of the stderr that is current NOW (and to `path`).
Never called by the product path.
"""
import ctypes
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_DIR, 'hk_abort_trace.c')
_LIB = os.path.join(_DIR, '_build', 'libhk_abort_trace.so')
_state = {}


def build(force: bool = False) -> str:
    """ gcc the helper (plain C, no GPU code) into harness/_build/ when it is missing or older than its source. """
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        subprocess.run(['gcc', '-O1', '-g', '-std=gnu11', '-fPIC', '-shared', '-o', _LIB, _SRC], check=True)
    return _LIB


def install(path: str = None, fd: int = None) -> bool:
    """ Install the handler once per process; False when the helper library is not built. """
    if _state:
        return True
    if not os.path.exists(_LIB):
        try:
            build()
        except Exception:
            return False
    lib = ctypes.CDLL(_LIB)
    lib.hk_abort_trace_install.argtypes = [ctypes.c_int, ctypes.c_char_p]
    lib.hk_abort_trace_install.restype = ctypes.c_int
    out_fd = os.dup(2 if fd is None else fd)
    rc = lib.hk_abort_trace_install(out_fd, path.encode() if path else None)
    _state.update(lib=lib, fd=out_fd, rc=rc)
    return rc == 0
