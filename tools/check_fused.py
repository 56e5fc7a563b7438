#!/usr/bin/env python3
"""GPU check of the one-pass backward (bwd_fused_kernel) against the fp64 oracle and the two-kernel backward, plus timing.
usage: python tools/check_fused.py [--time]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import os as _os
_os.environ["FA_MI355X_DIAG"] = "1"   # tools use the diagnostic build (set_tuning, stamps, ablations)
from flash_attention_minitorch_amd import _lib, device_ops  # noqa: E402


def set_split(on):
    _lib.core().fa_mi355x_set_tuning(4, 0 if on else 2)   # 2 = one-pass backward (opt-in), 0 = two kernels


def status(ws, bh, n, d):
    st = ctypes.c_int(0)
    rc = _lib.core().fa_mi355x_bwd_status(ctypes.c_void_p(ws.data_ptr()), bh, n, d, ctypes.byref(st))
    return rc, st.value


def run(B, H, N, d=64, seed=0, check_oracle=True, heads=None):
    rng = np.random.default_rng(seed)
    qf, kf, vf, dof = (oracle.bf16_round(rng.uniform(-1, 1, (B * H, N, d)).astype(np.float32)) for _ in range(4))
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in (qf, kf, vf, dof))
    o, L, _ = device_ops.flash_attn_fwd(tq, tk, tv, causal=False)
    ws = device_ops.bwd_workspace(tq)
    set_split(False)
    g1 = [t.clone() for t in device_ops.flash_attn_bwd(tq, tk, tv, o, tdo, L, workspace=ws)]
    torch.cuda.synchronize()
    rc, st = status(ws, B * H, N, d)
    g1b = [t.clone() for t in device_ops.flash_attn_bwd(tq, tk, tv, o, tdo, L, workspace=ws)]
    torch.cuda.synchronize()
    set_split(True)
    g2 = [t.clone() for t in device_ops.flash_attn_bwd(tq, tk, tv, o, tdo, L, workspace=ws)]
    torch.cuda.synchronize()
    set_split(False)
    names = ("dq", "dk", "dv")
    msg = [f"B{B} H{H} N{N}: status rc={rc} word={st}"]
    ok = rc == 0
    for n_, a, b, c in zip(names, g1, g2, g1b):
        dsplit = float((a - b).abs().max())
        rep = bool(torch.equal(a, c))
        msg.append(f"{n_}: |fused-split|={dsplit:.2e} repeat_bitwise={rep}")
        ok &= rep and dsplit < 2e-3 and bool(torch.isfinite(a).all())
    if check_oracle:
        hs = range(B * H) if heads is None else heads
        for hh in hs:
            rdq, rdk, rdv = oracle.dense_attention_bw(qf[hh:hh + 1], kf[hh:hh + 1], vf[hh:hh + 1], dof[hh:hh + 1])
            for n_, a, ref in zip(names, g1, (rdq, rdk, rdv)):
                e = float(np.max(np.abs(a[hh].cpu().numpy() - ref[0])))
                if e > 1e-3:
                    ok = False
                    msg.append(f"  head {hh} {n_} err {e:.2e} !!")
        msg.append("oracle checked")
    print(("OK   " if ok else "FAIL ") + " | ".join(msg), flush=True)
    return ok


def timeit(B, H, N, d=64, iters=30):
    gen = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: ((torch.rand((B * H, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal=False)
    ws = device_ops.bwd_workspace(q)
    grads = tuple(torch.empty((B * H, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
    res = {}
    variants = [("fused", False, 0), ("split", True, 0), ("fused2", False, 0), ]
    if "--abl" in sys.argv:
        variants += [("abl1_nohandoff", False, 1), ("abl2_nodq", False, 2), ("abl3", False, 3), ("abl7_dkdvonly", False, 7), ("abl8_nopre", False, 8)]
    if "--abl2" in sys.argv:
        variants = [("fused", False, 0), ("nohandoff", False, 1), ("nostores", False, 64), ("noloads", False, 128), ("noflags", False, 256),
                    ("flags_only", False, 192), ("loads_only", False, 320), ("stores_only", False, 384)]
    for name, split, abl in variants:
        set_split(split)
        _lib.core().fa_mi355x_set_tuning(5, abl)
        for _ in range(10):
            device_ops.flash_attn_bwd(q, k, v, o, do, L, workspace=ws, grads=grads)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            device_ops.flash_attn_bwd(q, k, v, o, do, L, workspace=ws, grads=grads)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters
    set_split(False)
    _lib.core().fa_mi355x_set_tuning(5, 0)
    _lib.core().fa_mi355x_set_tuning(3, 0)
    fl = 10.0 * B * H * N * N * d
    print(f"time B{B} H{H} N{N}: " + "  ".join(f"{k}={v:.4f} ms ({fl / v / 1e9:.0f} TF/s)" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    good = True
    good &= run(1, 1, 256)
    good &= run(1, 2, 512)
    good &= run(2, 3, 1024, seed=3)
    good &= run(1, 20, 768, seed=4, heads=[0, 7, 19])
    good &= run(8, 8, 4096, seed=5, heads=[0, 13, 63])
    good &= run(3, 7, 2048, seed=6, heads=[0, 20])
    print("ALL OK" if good else "SOME FAILED", flush=True)
    if "--time" in sys.argv:
        timeit(8, 8, 4096)
        if "--shapes" in sys.argv:
            timeit(4, 8, 8192)
            timeit(16, 8, 2048)
            timeit(32, 8, 1024)
            timeit(64, 8, 512)
            timeit(128, 8, 256)
            timeit(2, 8, 16384)
    sys.exit(0 if good else 1)
