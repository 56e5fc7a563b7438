#!/usr/bin/env python3
"""Per-kernel timings of the device path for every BASELINE.json config shape (HIP events, best of 3 rounds)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_attention_minitorch_amd import device_ops

SHAPES = [("c1 fp32 fw", 8, 8, 1024, 64, torch.float32, False), ("c2 fp32 fw+bw", 8, 8, 2048, 64, torch.float32, False),
          ("M bf16 fw+bw", 8, 8, 4096, 64, torch.bfloat16, False), ("M bf16 fw+bw causal", 8, 8, 4096, 64, torch.bfloat16, True),
          ("c3 bf16 fw+bw", 16, 16, 4096, 128, torch.bfloat16, False), ("c4/8 bf16 fw (one GPU's shard)", 16, 16, 4096, 128, torch.bfloat16, False)]


def t_ms(fn, iters=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


res = {}
for name, B, H, N, d, dt, causal in SHAPES:
    BH = B * H
    mk = lambda: ((torch.rand((BH, N, d), device="cuda") - 0.5) * 2).to(dt)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal)
    ws = device_ops.bwd_workspace(q)
    grads = tuple(torch.empty((BH, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
    cf = 0.5 if causal else 1.0
    fl = BH * N * N * d * cf
    fw = t_ms(lambda: device_ops.flash_attn_fwd(q, k, v, causal, out=o, l=L))
    bw = t_ms(lambda: device_ops.flash_attn_bwd(q, k, v, o, do, L, None, causal, workspace=ws, grads=grads))
    res[name] = {"fw_ms": round(fw, 4), "bw_ms": round(bw, 4), "fw_TFLOPs": round(4 * fl / fw / 1e9, 1),
                 "bw_TFLOPs": round(10 * fl / bw / 1e9, 1), "fwbw_TFLOPs": round(14 * fl / (fw + bw) / 1e9, 1)}
    del q, k, v, do, o, L, ws, grads
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
