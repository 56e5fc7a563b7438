#!/usr/bin/env python3
"""A/B timing of kernel variants in ONE process (interleaved rounds): per-kernel ms for each tuning value.
usage: bench_variants.py [B H N d] [--knobs 0:0,1,2 1:0,1 2:0,1]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ["FA_MI355X_DIAG"] = "1"   # tools use the diagnostic build (set_tuning, stamps, ablations)
from flash_attention_minitorch_amd import device_ops, _lib

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B, H, N, d = (8, 8, 4096, 64) if len(args) < 4 else map(int, args[:4])
knobs = {0: [0, 1, 2], 1: [0, 1], 2: [0, 1]}
for a in sys.argv[1:]:
    if a.startswith("--knobs="):
        knobs = {int(kv.split(":")[0]): [int(x) for x in kv.split(":")[1].split(",")] for kv in a[8:].split()}
causal = "--causal" in sys.argv
DT = torch.float32 if "--f32" in sys.argv else torch.bfloat16
BH = B * H
mk = lambda: ((torch.rand((BH, N, d), device="cuda") - 0.5) * 2).to(DT)
q, k, v, do = mk(), mk(), mk(), mk()
o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal)
ws = device_ops.bwd_workspace(q)
grads = tuple(torch.empty((BH, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
KERNEL_OF_KNOB = {0: "dkdv", 1: "fwd", 2: "dq"}


def t_ms(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


run = {"fwd": lambda: device_ops.flash_attn_fwd(q, k, v, causal, out=o, l=L),
       "dkdv": lambda: device_ops.flash_attn_bwd(q, k, v, o, do, L, None, causal, workspace=ws, grads=grads, stages=device_ops.STAGE_DKDV),
       "dq": lambda: device_ops.flash_attn_bwd(q, k, v, o, do, L, None, causal, workspace=ws, grads=grads, stages=device_ops.STAGE_DQ)}
# correctness of every variant against variant 0 of the same kernel (max-abs difference of its outputs)
def outputs(kname):
    run[kname]()
    torch.cuda.synchronize()
    return [o.clone(), L.clone()] if kname == "fwd" else [g.clone() for g in grads]

for knob, vals in knobs.items():
    kname = KERNEL_OF_KNOB[knob]
    _lib.core().fa_mi355x_set_tuning(knob, 0)
    ref = outputs(kname)
    for val in vals:
        if val >= 10:
            continue
        _lib.core().fa_mi355x_set_tuning(knob, val)
        got = outputs(kname)
        sel = {"fwd": (0, 1), "dkdv": (1, 2), "dq": (0,)}[kname]
        print(f"{kname}[{val}] max|diff| vs [0]:", [float((got[i] - ref[i]).abs().max()) for i in sel])
    _lib.core().fa_mi355x_set_tuning(knob, 0)

res = {}
for rnd in range(3):
    for knob, vals in knobs.items():
        for val in vals:
            _lib.core().fa_mi355x_set_tuning(knob, val)
            res.setdefault(f"{KERNEL_OF_KNOB[knob]}[{val}]", []).append(round(t_ms(run[KERNEL_OF_KNOB[knob]]), 4))
        _lib.core().fa_mi355x_set_tuning(knob, 0)
fl = BH * N * N * d * (0.5 if causal else 1.0)
mult = {"fwd": 4, "dkdv": 8, "dq": 2}
for name, ts in res.items():
    print(f"{name:10s} ms {ts}  best {min(ts):.4f}  algorithmic TFLOP/s {mult[name.split('[')[0]] * fl / (min(ts) * 1e-3) / 1e12:.1f}")
