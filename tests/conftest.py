import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _install_native_backtrace():
    """GPU runs only: native frames of the faulting thread on SIGABRT / SIGSEGV / SIGBUS (tests/abort_trace.c), chained in front of
    pytest's faulthandler.  A GPU-side fault reaches the process as abort() from a ROCm runtime thread, for which Python's
    faulthandler prints no frame.  Best effort: any failure here leaves the run as it was."""
    import ctypes
    import subprocess
    import tempfile
    try:
        src = os.path.join(ROOT, "tests", "abort_trace.c")
        so = os.path.join(tempfile.gettempdir(), f"fa_abort_trace_{os.getuid()}.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O1", "-fPIC", "-shared", "-rdynamic", src, "-o", so], check=True, capture_output=True, timeout=60)
        lib = ctypes.CDLL(so)
        out_dir = os.path.join(ROOT, "gpurun_out")
        log = os.path.join(out_dir if os.path.isdir(out_dir) else tempfile.gettempdir(), "abort_trace.log")
        lib.fa_install_abort_trace.argtypes = [ctypes.c_char_p]
        lib.fa_install_abort_trace(log.encode())   # (pytest captures fd 2 per test and drops it when the process dies)
        return lib
    except Exception:   # pragma: no cover
        return None


_ABORT_TRACE = None


def pytest_sessionstart(session):
    global _ABORT_TRACE
    markexpr = session.config.getoption("markexpr", "") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        _ABORT_TRACE = _install_native_backtrace()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
