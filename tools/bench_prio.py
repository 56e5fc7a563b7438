#!/usr/bin/env python3
"""A/B of the static young-half priority (tuning key 3: waves 4-7 of the slot kernels at s_setprio 1) in one process,
interleaved rounds, per kernel.  usage: python tools/bench_prio.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ["FA_MI355X_DIAG"] = "1"   # tools use the diagnostic build (set_tuning, stamps, ablations)
from flash_attention_minitorch_amd import device_ops, _lib
B, H, N, d = 8, 8, 4096, 64
BH = B * H
mk = lambda: ((torch.rand((BH, N, d), device="cuda") - 0.5) * 2).to(torch.bfloat16)
q, k, v, do = mk(), mk(), mk(), mk()
o, L, _ = device_ops.flash_attn_fwd(q, k, v)
ws = device_ops.bwd_workspace(q)
grads = tuple(torch.empty((BH, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
run = {"fwd": lambda: device_ops.flash_attn_fwd(q, k, v, out=o, l=L),
       "dkdv": lambda: device_ops.flash_attn_bwd(q, k, v, o, do, L, None, False, workspace=ws, grads=grads, stages=device_ops.STAGE_DKDV),
       "dq": lambda: device_ops.flash_attn_bwd(q, k, v, o, do, L, None, False, workspace=ws, grads=grads, stages=device_ops.STAGE_DQ)}
def t_ms(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
device_ops.flash_attn_bwd(q, k, v, o, do, L, None, False, workspace=ws, grads=grads)
res = {}
for rnd in range(4):
    for prio in (0, 1):
        _lib.core().fa_mi355x_set_tuning(3, prio)
        for name, fn in run.items():
            res.setdefault((name, prio), []).append(round(t_ms(fn), 4))
_lib.core().fa_mi355x_set_tuning(3, 0)
for (name, prio), ts in sorted(res.items()):
    print(f"{name:5s} prio {prio}: {ts}  best {min(ts):.4f}")
