#!/usr/bin/env python3
"""A/B timing of kernel variants in ONE process (interleaved rounds): prints per-kernel ms for each tuning value."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_attention_minitorch_amd import device_ops, _lib

B, H, N, d = 8, 8, 4096, 64
if len(sys.argv) > 1:
    B, H, N, d = map(int, sys.argv[1:5])
BH = B * H
mk = lambda: ((torch.rand((BH, N, d), device="cuda") - 0.5) * 2).to(torch.bfloat16)
q, k, v, do = mk(), mk(), mk(), mk()
o, L, _ = device_ops.flash_attn_fwd(q, k, v)
ws = device_ops.bwd_workspace(q)
grads = tuple(torch.empty((BH, N, d), dtype=torch.float32, device="cuda") for _ in range(3))


def t_ms(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bw(stage):
    return lambda: device_ops.flash_attn_bwd(q, k, v, o, do, L, None, False, workspace=ws, grads=grads, stages=stage)


res = {}
for rnd in range(3):
    for cfg in (1, 0):
        _lib.core().fa_mi355x_set_tuning(0, cfg)
        res.setdefault(f"dkdv_cfg{cfg}", []).append(round(t_ms(bw(device_ops.STAGE_DKDV)), 4))
    res.setdefault("fwd", []).append(round(t_ms(lambda: device_ops.flash_attn_fwd(q, k, v, out=o, l=L)), 4))
    res.setdefault("dq", []).append(round(t_ms(bw(device_ops.STAGE_DQ)), 4))
fl = BH * N * N * d
print(json.dumps(res))
for kname, mult in (("fwd", 4), ("dkdv_cfg1", 8), ("dkdv_cfg0", 8), ("dq", 2)):
    print(kname, "best ms", min(res[kname]), "algorithmic TFLOP/s", round(mult * fl / (min(res[kname]) * 1e-3) / 1e12, 1))
