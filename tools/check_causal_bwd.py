#!/usr/bin/env python3
"""GPU check + timing of the causal backward builds at d = 64 against the fp64 oracle: phased dQ kernel (opts[2] = 2) vs the causal
slot build (opts[2] = 3: unmasked sweep + diagonal block per wave, paired query blocks).
usage: python tools/check_causal_bwd.py [--time]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from flash_attention_minitorch_amd import device_ops  # noqa: E402

VARIANTS = {"phased": (3, 0, 2), "slot": (5, 0, 3), "slot_paired": (5, 0, 3, 0, 0, 0, 0, 1), "slot_ranked": (5, 0, 3, 0, 0, 0, 0, 2)}   # key 0 = 3: phased dK/dV
NAMES = ("dq", "dk", "dv")


def check(B, H, N, d=64, seed=0, heads=(0,), scale=1.0):
    rng = np.random.default_rng(seed)
    qf, kf, vf, dof = (oracle.bf16_round((sc * rng.uniform(-1, 1, (B * H, N, d))).astype(np.float32)) for sc in (scale, 1.0, 1.0, 1.0))
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in (qf, kf, vf, dof))
    o, L, _ = device_ops.flash_attn_fwd(tq, tk, tv, causal=True)
    ok = True
    msg = [f"B{B} H{H} N{N} scale {scale}"]
    res = {}
    for name, opts in VARIANTS.items():
        g = device_ops.flash_attn_bwd(tq, tk, tv, o, tdo, L, causal=True, opts=opts)
        torch.cuda.synchronize()
        res[name] = [t.cpu().numpy() for t in g]
        errs = [0.0, 0.0, 0.0]
        for hh in heads:
            ref = oracle.dense_attention_bw(qf[hh:hh + 1], kf[hh:hh + 1], vf[hh:hh + 1], dof[hh:hh + 1], causal=True)
            for i in range(3):
                errs[i] = max(errs[i], float(np.max(np.abs(res[name][i][hh] - ref[i][0]))))
        fin = all(bool(np.isfinite(a).all()) for a in res[name])
        ok &= fin and max(errs) <= 1e-3
        msg.append(f"{name}: " + " ".join(f"{n} {e:.2e}" for n, e in zip(NAMES, errs)) + f" finite {fin}")
    msg.append("|slot-phased| " + " ".join(f"{n} {float(np.max(np.abs(a - b))):.2e}" for n, a, b in zip(NAMES, res["slot"], res["phased"])))
    print(("OK   " if ok else "FAIL ") + " | ".join(msg), flush=True)
    return ok


def timeit(B, H, N, d=64, iters=50):
    gen = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: ((torch.rand((B * H, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal=True)
    ws = device_ops.bwd_workspace(q)
    grads = tuple(torch.empty((B * H, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
    res = {}

    def run(causal, stages, opts):
        device_ops.flash_attn_bwd(q, k, v, o, do, L, None, causal, workspace=ws, grads=grads, stages=stages, opts=opts)

    def t(causal, stages, opts):
        for _ in range(10):
            run(causal, stages, opts)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run(causal, stages, opts)
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / iters, 4)

    run(True, device_ops.STAGE_PREP, None)
    for rnd in range(2):
        for name, opts in VARIANTS.items():
            res[f"dq_{name}{rnd}"] = t(True, device_ops.STAGE_DQ, opts)
    res["dq_noncausal_half"] = round(t(False, device_ops.STAGE_DQ, None) / 2, 4)
    res["dkdv_causal_phased"] = t(True, device_ops.STAGE_DKDV, (3,))
    res["dkdv_causal_slot"] = t(True, device_ops.STAGE_DKDV, None)
    res["dkdv_causal_phased1"] = t(True, device_ops.STAGE_DKDV, (3,))
    res["dkdv_causal_slot1"] = t(True, device_ops.STAGE_DKDV, None)
    res["dkdv_noncausal_half"] = round(t(False, device_ops.STAGE_DKDV, None) / 2, 4)
    print(f"time B{B} H{H} N{N}: {res}", flush=True)


if __name__ == "__main__":
    ok = True
    for shape in ((1, 2, 256), (1, 2, 512), (1, 3, 768), (2, 2, 1024), (1, 2, 1280), (1, 1, 4096)):
        ok &= check(*shape)
    ok &= check(1, 2, 1024, seed=3, scale=3.0)
    if "--time" in sys.argv:
        for shape in ((8, 8, 4096), (32, 8, 4096), (8, 8, 2048), (8, 8, 1024), (2, 8, 4096)):
            timeit(*shape)
    sys.exit(0 if ok else 1)
