#!/usr/bin/env python3
"""GPU check + timing of the tiled dK/dV build (several consecutive key blocks per workgroup; option 5 = 1 turns it off).
usage: python tools/check_tiled.py [--time]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from flash_attention_minitorch_amd import device_ops  # noqa: E402

UNTILED = (0, 0, 0, 0, 0, 1)


def check(BH, N, d=64, seed=0, heads=(0,)):
    rng = np.random.default_rng(seed)
    arrs = [oracle.bf16_round(rng.uniform(-1, 1, (BH, N, d)).astype(np.float32)) for _ in range(4)]
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    o, L, _ = device_ops.flash_attn_fwd(tq, tk, tv)
    res = {}
    for name, opts in (("tiled", None), ("untiled", UNTILED)):
        g = device_ops.flash_attn_bwd(tq, tk, tv, o, tdo, L, opts=opts)
        torch.cuda.synchronize()
        res[name] = [t.cpu().numpy() for t in g]
    ok = True
    msg = [f"BH{BH} N{N}"]
    for i, nm in enumerate(("dq", "dk", "dv")):
        same = bool(np.array_equal(res["tiled"][i], res["untiled"][i]))
        msg.append(f"{nm} bitwise_equal {same}")
        ok &= same
    for hh in heads:
        ref = oracle.dense_attention_bw(*(a[hh:hh + 1] for a in arrs))
        for i, nm in enumerate(("dq", "dk", "dv")):
            e = float(np.max(np.abs(res["tiled"][i][hh] - ref[i][0])))
            ok &= e <= 1e-3
            msg.append(f"h{hh} {nm} {e:.1e}")
    print(("OK   " if ok else "FAIL ") + " | ".join(msg), flush=True)
    return ok


def timeit(B, H, N, d=64, iters=50):
    gen = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: ((torch.rand((B * H, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, L, _ = device_ops.flash_attn_fwd(q, k, v)
    ws = device_ops.bwd_workspace(q)
    grads = tuple(torch.empty((B * H, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
    device_ops.flash_attn_bwd(q, k, v, o, do, L, None, False, workspace=ws, grads=grads, stages=device_ops.STAGE_PREP)
    res = {}
    for rnd in range(2):
        for name, opts in (("tiled", None), ("untiled", UNTILED)):
            run = lambda: device_ops.flash_attn_bwd(q, k, v, o, do, L, None, False, workspace=ws, grads=grads,
                                                    stages=device_ops.STAGE_DKDV, opts=opts)
            for _ in range(10):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[f"{name}{rnd}"] = round(e0.elapsed_time(e1) / iters, 4)
    print(f"time dK/dV B{B} H{H} N{N}: {res}", flush=True)


if __name__ == "__main__":
    ok = True
    for shape in ((256, 512), (300, 1024), (64, 4096), (512, 256), (7, 2048)):
        ok &= check(*shape, heads=(0, shape[0] - 1))
    if "--time" in sys.argv:
        for shape in ((8, 8, 4096), (32, 8, 4096), (8, 8, 2048), (16, 8, 1024), (4, 8, 8192)):
            timeit(*shape)
    sys.exit(0 if ok else 1)
