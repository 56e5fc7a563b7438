"""Helpers shared by the GPU parity tests (imported only under -m gpu)."""
import os
import numpy as np

import oracle


def rand_u(rng, shape):
    # U(-1, 1) like the reference harness, test_utils.py:104-105
    return ((rng.random(shape, dtype=np.float32) - np.float32(0.5)) * np.float32(2)).astype(np.float32)


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def to_np(t):
    return t.detach().float().cpu().numpy()


def oracle_heads(q, k, v, do, causal, heads):
    """fp64 dense oracle on a subset of the flattened (BH, N, d) heads.  Returns dict of stacked arrays."""
    out = {n: [] for n in ("o", "L", "m", "l", "dq", "dk", "dv")}
    for hh in heads:
        o, L, m, l = oracle.dense_attention_fw(q[hh], k[hh], v[hh], causal)
        out["o"].append(o); out["L"].append(L); out["m"].append(m); out["l"].append(l)
        if do is not None:
            dq, dk, dv = oracle.dense_attention_bw(q[hh], k[hh], v[hh], do[hh], causal)
            out["dq"].append(dq); out["dk"].append(dk); out["dv"].append(dv)
    return {n: np.stack(a) for n, a in out.items() if a}


def check_abort_hook(tmp_path):
    """tests/abort_trace.c (installed by conftest.py on GPU runs: a GPU-side fault reaches the process as abort() from a ROCm runtime
    thread, for which Python's faulthandler prints no frame) had never fired: a child process installs it, calls
    abort() from a NON-Python thread of a small C helper, and the native backtrace must land in the hook's log file with the helper's
    frame in it, before the process dies of SIGABRT."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    helper = tmp_path / "abort_from_thread.c"
    helper.write_text(textwrap.dedent("""
        #include <pthread.h>
        #include <stdlib.h>
        void* fa_test_runtime_thread(void* p) { (void)p; abort(); return 0; }
        int fa_test_abort_from_thread(void) {
          pthread_t t;
          pthread_create(&t, 0, fa_test_runtime_thread, 0);
          pthread_join(t, 0);
          return 0;
        }
    """))
    so_h, so_t, log = tmp_path / "helper.so", tmp_path / "trace.so", tmp_path / "abort_trace.log"
    subprocess.run(["gcc", "-O0", "-fPIC", "-shared", "-rdynamic", "-pthread", str(helper), "-o", str(so_h)], check=True)
    subprocess.run(["gcc", "-O1", "-fPIC", "-shared", "-rdynamic", os.path.join(root, "tests", "abort_trace.c"), "-o", str(so_t)], check=True)
    code = textwrap.dedent(f"""
        import ctypes, faulthandler
        faulthandler.enable()
        t = ctypes.CDLL({str(so_t)!r})
        t.fa_install_abort_trace.argtypes = [ctypes.c_char_p]
        assert t.fa_install_abort_trace({str(log)!r}.encode()) == 0
        ctypes.CDLL({str(so_h)!r}).fa_test_abort_from_thread()
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == -6, (r.returncode, r.stderr[-500:])           # died of SIGABRT, after the hook
    text = log.read_text()
    assert "native backtrace (tests/abort_trace.c)" in text and "helper.so(" in text, text[-1500:]   # (the frame of the thread that aborted)
    assert "native backtrace" in r.stderr                                  # ... and on stderr, in front of faulthandler's dump
