"""
Diagnostics of the test / bench harness -- NOT part of the product package (`homonim_amd` never imports this).

    abort_trace     a fatal signal names its sender, thread and native frames (hk_abort_trace.c -> _build/libhk_abort_trace.so)
    first_process   a child process is the lease's first GPU process; its fate is reported

Used by tests/conftest.py, bench.py and __graft_entry__.smoke() only.
"""
