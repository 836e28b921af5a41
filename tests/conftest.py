import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_sessionstart(session):
    """GPU runs only: a C-level SIGABRT handler (tests/abort_trace.c) that records the C backtrace of the aborting thread
    and the tail of the captured stderr -- a process killed by abort() inside the runtime leaves pytest nothing to print.
    Test infrastructure; any failure to set it up is ignored."""
    if not _has_gpu():
        return
    try:
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))   # (a core file of a process holding tens of GB takes minutes to write)
    except Exception:
        pass
    try:
        import ctypes, subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so, src = os.path.join(here, "_abort_trace.so"), os.path.join(here, "abort_trace.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, src], check=True, capture_output=True, timeout=120)
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        lib = ctypes.CDLL(so)
        lib.gf3_install_abort_trace.argtypes = [ctypes.c_char_p, ctypes.c_int]
        term = -1
        try:                                             # the fd pytest's faulthandler plugin kept of the real stderr
            from _pytest.faulthandler import fault_handler_stderr_fd_key
            term = int(session.config.stash[fault_handler_stderr_fd_key])
        except Exception:
            pass
        lib.gf3_install_abort_trace(os.path.join(out_dir, "abort_trace.txt").encode(), term)
        session.config._gf3_abort_trace = lib            # (keeps the library loaded)
    except Exception:
        pass
