#!/usr/bin/env python3
"""GPU check + timing of the causal forward builds at d = 64: default (phased, split-operand build) vs the causal slot build
(opts[1] = 3: unmasked sweep + diagonal block per wave), both against the fp64 oracle.
usage: python tools/check_causal_fwd.py [--time]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from flash_attention_minitorch_amd import device_ops  # noqa: E402

VARIANTS = {"default": (0, 2), "slot": (0, 3), "slot_paired": (0, 3, 0, 0, 0, 0, 0, 1), "slot_ranked": (0, 3, 0, 0, 0, 0, 0, 2)}   # phased (split-operand build) vs causal slot build


def check(B, H, N, d=64, seed=0, heads=(0,), scale=1.0):
    rng = np.random.default_rng(seed)
    qf, kf, vf = (oracle.bf16_round((sc * rng.uniform(-1, 1, (B * H, N, d))).astype(np.float32)) for sc in (scale, 1.0, 1.0))
    tq, tk, tv = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in (qf, kf, vf))
    ok = True
    msg = [f"B{B} H{H} N{N} d{d} scale {scale}"]
    outs = {}
    for name, opts in VARIANTS.items():
        o, L, _ = device_ops.flash_attn_fwd(tq, tk, tv, causal=True, opts=opts)
        torch.cuda.synchronize()
        outs[name] = (o.cpu().numpy(), L.cpu().numpy())
        eo = el = 0.0
        for hh in heads:
            ro, rl = oracle.dense_attention_fw(qf[hh:hh + 1], kf[hh:hh + 1], vf[hh:hh + 1], causal=True)[:2]
            eo = max(eo, float(np.max(np.abs(outs[name][0][hh] - ro[0]))))
            el = max(el, float(np.max(np.abs(outs[name][1][hh] - rl[0]))))
        fin = bool(np.isfinite(outs[name][0]).all())
        ok &= fin and eo <= 1e-3 and el <= 1e-3
        msg.append(f"{name}: errO {eo:.2e} errL {el:.2e} finite {fin}")
    msg.append(f"|slot-default| {float(np.max(np.abs(outs['slot'][0] - outs['default'][0]))):.2e}")
    print(("OK   " if ok else "FAIL ") + " | ".join(msg), flush=True)
    return ok


def timeit(B, H, N, d=64, iters=50):
    gen = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: ((torch.rand((B * H, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v = mk(), mk(), mk()
    out = torch.empty((B * H, N, d), dtype=torch.float32, device="cuda")
    L = torch.empty((B * H, N), dtype=torch.float32, device="cuda")
    res = {}
    for rnd in range(2):
        for name, opts in VARIANTS.items():
            for _ in range(10):
                device_ops.flash_attn_fwd(q, k, v, True, out=out, l=L, opts=opts)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                device_ops.flash_attn_fwd(q, k, v, True, out=out, l=L, opts=opts)
            e1.record()
            torch.cuda.synchronize()
            res[f"{name}{rnd}"] = round(e0.elapsed_time(e1) / iters, 4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        device_ops.flash_attn_fwd(q, k, v, False, out=out, l=L)
    e1.record()
    torch.cuda.synchronize()
    res["noncausal_half"] = round(e0.elapsed_time(e1) / iters / 2, 4)
    print(f"time B{B} H{H} N{N} d{d}: {res}", flush=True)


if __name__ == "__main__":
    ok = True
    for shape in ((1, 2, 256), (1, 2, 512), (1, 3, 768), (2, 2, 1024), (1, 2, 1280), (1, 1, 4096)):
        ok &= check(*shape)
    ok &= check(1, 2, 1024, seed=3, scale=3.0)   # larger scores: the reference moves in the diagonal block
    for shape in ((1, 2, 256), (1, 3, 768), (2, 2, 1024), (1, 1, 4096)):
        ok &= check(*shape, d=128)
    if "--time" in sys.argv:
        for shape in ((8, 8, 4096), (32, 8, 4096), (8, 8, 2048), (8, 8, 1024), (2, 8, 4096), (1, 8, 8192), (16, 8, 512)):
            timeit(*shape)
        timeit(16, 16, 4096, d=128, iters=10)
        timeit(8, 8, 4096, d=128, iters=20)
    sys.exit(0 if ok else 1)
