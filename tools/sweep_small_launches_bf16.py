#!/usr/bin/env python3
"""bf16 d = 64 forward / backward at launches below the size of the chip: the MFMA-slot kernels (default; 512-thread workgroups, one
per 256 rows) against the phased ones (256-thread workgroups, one per 128 rows): which side a small launch should take.
usage: python tools/sweep_small_launches_bf16.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flash_attention_minitorch_amd import device_ops  # noqa: E402

FOLD = device_ops.OPTS_FOLDED_SCALE
PH = device_ops.OPTS_PHASED


def t_ms(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(B, H, N, causal, d=64):
    gen = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: ((torch.rand((B * H, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal)
    ws = device_ops.bwd_workspace(q)
    g = tuple(torch.empty((B * H, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
    res = {}
    for name, opts in (("slot", FOLD), ("phased", PH)):
        res["fw_" + name] = t_ms(lambda: device_ops.flash_attn_fwd(q, k, v, causal, out=o, l=L, opts=opts, guard=None))
        res["bw_" + name] = t_ms(lambda: device_ops.flash_attn_bwd(q, k, v, o, do, L, None, causal, workspace=ws, grads=g, opts=opts, guard=None))
    cf = 0.5 if causal else 1.0
    fl = B * H * N * N * d * cf
    print(f"B{B} H{H} N{N}{' causal' if causal else ''} ({B * H * N // 256} blocks): " +
          "  ".join(f"{kk}={vv:.4f} ms ({(4 if kk.startswith('fw') else 10) * fl / vv / 1e9:.0f} TF/s)" for kk, vv in res.items()), flush=True)


if __name__ == "__main__":
    for causal in (False, True):
        for shp in ((1, 8, 1024), (1, 8, 2048), (2, 8, 1024), (2, 8, 2048), (4, 4, 2048), (4, 8, 1024), (1, 8, 4096), (4, 8, 2048), (8, 8, 1024), (8, 8, 4096)):
            run(*shp, causal)
