#!/usr/bin/env python3
"""The causal tiled dK/dV build (bwd_dkdv_slot_kernel<.., CDIAG, TILED>: key blocks p and nkb-1-p of several heads per workgroup;
DIAGNOSTIC LIBRARY, option 5 = 2) against the one-block-per-workgroup build (the default) -- bitwise, both scalings -- and against
the fp64 oracle, plus timing.  Result (profiles/r04_causal_tiled_dkdv.txt): bitwise equal and 4-8 % slower: not kept.
usage: python tools/check_causal_tiled.py [--time]"""
import os
import sys

import numpy as np
import torch

os.environ["FA_MI355X_DIAG"] = "1"   # the tiled causal build lives in the diagnostic library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from gpu_util import oracle_heads, to_np  # noqa: E402
from flash_attention_minitorch_amd import device_ops as dev  # noqa: E402

STAGE = dev.STAGE_DKDV


def opts(mode, untiled):
    return (5, 3, 3, 0, 0, 0 if untiled else 2, 0, 0, mode)


def check(BH, N, seed, heads):
    rng = np.random.default_rng(seed)
    arrs = [oracle.bf16_round(((rng.random((BH, N, 64), dtype=np.float32) - 0.5) * 2).astype(np.float32)) for _ in range(4)]
    t = [torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs]
    o, L, _ = dev.flash_attn_fwd(*t[:3], True, opts=dev.OPTS_FOLDED_SCALE)
    ok = True
    msg = [f"BH {BH} N {N}"]
    for mode in (1, 2):
        a = [to_np(x) for x in dev.flash_attn_bwd(*t[:3], o, t[3], L, None, True, opts=opts(mode, False))]
        b = [to_np(x) for x in dev.flash_attn_bwd(*t[:3], o, t[3], L, None, True, opts=opts(mode, True))]
        a2 = [to_np(x) for x in dev.flash_attn_bwd(*t[:3], o, t[3], L, None, True, opts=opts(mode, False))]
        same = all(np.array_equal(x, y) for x, y in zip(a, b)) and all(np.array_equal(x, y) for x, y in zip(a, a2))
        ref = oracle_heads(*arrs, True, heads)
        err = max(float(np.max(np.abs(x[heads] - ref[n]))) for x, n in zip(a, ("dq", "dk", "dv")))
        ok &= same and err < 1e-3
        msg.append(f"mode {mode}: tiled == untiled == rerun bitwise {same}, oracle err {err:.2e}")
    print(("OK   " if ok else "FAIL ") + " | ".join(msg), flush=True)
    return ok


def timeit(BH, N, iters=40):
    gen = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: ((torch.rand((BH, N, 64), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, L, _ = dev.flash_attn_fwd(q, k, v, True, opts=dev.OPTS_FOLDED_SCALE)
    ws = dev.bwd_workspace(q)
    grads = tuple(torch.empty((BH, N, 64), dtype=torch.float32, device="cuda") for _ in range(3))
    dev.flash_attn_bwd(q, k, v, o, do, L, None, True, workspace=ws, grads=grads, opts=opts(1, False))   # (fills the row constants)
    res = {}
    for rnd in range(2):
        for name, op in (("tiled", opts(1, False)), ("untiled", opts(1, True)), ("tiled_fp32", opts(2, False)), ("untiled_fp32", opts(2, True))):
            for _ in range(5):
                dev.flash_attn_bwd(q, k, v, o, do, L, None, True, workspace=ws, grads=grads, stages=STAGE, opts=op)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                dev.flash_attn_bwd(q, k, v, o, do, L, None, True, workspace=ws, grads=grads, stages=STAGE, opts=op)
            e1.record()
            torch.cuda.synchronize()
            res[f"{name}{rnd}"] = e0.elapsed_time(e1) / iters
    print(f"time BH {BH} N {N} (dK/dV launch, ms): " + "  ".join(f"{k}={v:.4f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    good = True
    good &= check(8, 512, 1, [0, 7])          # one pair per head, one head per workgroup
    good &= check(16, 1024, 2, [0, 15])
    good &= check(64, 4096, 3, [0, 33])       # the metric shape: two heads x two blocks per workgroup
    good &= check(24, 2048, 4, [5, 23])
    good &= check(64, 1536, 5, [1, 63])       # an odd pair count is not tiled (nkb = 6: three pairs) -- nkb even, pairs odd
    good &= check(128, 768, 6, [2, 100])      # nkb = 3: odd, falls back to the ranked build
    print("ALL OK" if good else "SOME FAILED", flush=True)
    if "--time" in sys.argv:
        timeit(64, 4096)
        timeit(128, 2048)
        timeit(256, 4096)
    sys.exit(0 if good else 1)
