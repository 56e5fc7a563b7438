#!/usr/bin/env python3
"""Small launches: do the 256-row MFMA-slot kernels or the 128-row phased kernels win when the grid does not fill the chip?
Per kernel (forward, dQ incl. preprocess, dK/dV), bf16, non-causal, default dispatch vs the phased kernels.
usage: python tools/bench_small.py [d]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
import vanilla_gpu as vg
from flash_attention_minitorch_amd import device_ops as dev

d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for BH, N in ((8, 1024), (16, 1024), (32, 1024), (64, 1024), (8, 2048), (16, 2048), (32, 2048), (48, 2048), (64, 2048), (8, 4096), (16, 4096), (32, 4096),
              (16, 512), (64, 512), (128, 512), (256, 256), (64, 256)):
    gen = torch.Generator(device="cuda").manual_seed(3)
    mk = lambda: ((torch.rand((BH, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, L, _ = dev.flash_attn_fwd(q, k, v)
    ws = dev.bwd_workspace(q)
    g = tuple(torch.empty(q.shape, dtype=torch.float32, device="cuda") for _ in range(3))
    row = []
    for tag, opts in (("slot", None), ("phased", dev.OPTS_PHASED)):
        fw = lambda: dev.flash_attn_fwd(q, k, v, out=o, l=L, opts=opts)
        dq = lambda: dev.flash_attn_bwd(q, k, v, o, do, L, workspace=ws, grads=g, stages=dev.STAGE_PREP | dev.STAGE_DQ, opts=opts)
        dkdv = lambda: dev.flash_attn_bwd(q, k, v, o, do, L, workspace=ws, grads=g, stages=dev.STAGE_DKDV, opts=opts)
        row.append((tag, vg.time_ms(fw, 20, 5), vg.time_ms(dq, 20, 5), vg.time_ms(dkdv, 20, 5)))
    s, p = row
    print(f"BH {BH:4d} N {N:5d} blocks256 {BH * ((N + 255) // 256):5d}  fwd {s[1]:.4f} / {p[1]:.4f}  dq {s[2]:.4f} / {p[2]:.4f}  dkdv {s[3]:.4f} / {p[3]:.4f}   (slot / phased ms)", flush=True)
