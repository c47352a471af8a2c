"""
Opt-in diagnostics of the test / bench harness: say who raised a fatal signal.

A GPU fault ends the process with an abort() on a native thread of the HSA runtime; the runtime's one-line message goes
to fd 2, which pytest may have captured, and faulthandler shows Python frames only.  install() loads
lib/libhk_abort_trace.so (csrc/hk_abort_trace.c) and puts its handler in front of faulthandler's: signal origin, thread,
native backtrace and the tail of a captured stderr go to a duplicate of the stderr that is current NOW (and to `path`).
Never called by the product path.
"""
import ctypes
import os

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libhk_abort_trace.so')
_state = {}


def install(path: str = None, fd: int = None) -> bool:
    """ Install the handler once per process; False when the helper library is not built. """
    if _state:
        return True
    if not os.path.exists(_LIB):
        return False
    lib = ctypes.CDLL(_LIB)
    lib.hk_abort_trace_install.argtypes = [ctypes.c_int, ctypes.c_char_p]
    lib.hk_abort_trace_install.restype = ctypes.c_int
    out_fd = os.dup(2 if fd is None else fd)
    rc = lib.hk_abort_trace_install(out_fd, path.encode() if path else None)
    _state.update(lib=lib, fd=out_fd, rc=rc)
    return rc == 0
