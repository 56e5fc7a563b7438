#!/usr/bin/env python3
"""Runs the metric-shape backward a few times with the kernel options given on the command line (for rocprofv3 passes).
usage: python tools/prof_chain.py 0,0,0,0,3 [iters]"""
import os
import sys

os.environ["FA_MI355X_DIAG"] = "1"

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flash_attention_minitorch_amd import device_ops  # noqa: E402

opts = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] != "-" else None
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B, H, N, d = (int(x) for x in os.environ.get("FA_SHAPE", "8,8,4096,64").split(","))
gen = torch.Generator(device="cuda").manual_seed(1)
mk = lambda: ((torch.rand((B * H, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
q, k, v, do = mk(), mk(), mk(), mk()
o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal=False)
ws = device_ops.bwd_workspace(q, (0, 0, 0, 0, 3))
grads = tuple(torch.empty((B * H, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
for _ in range(iters):
    device_ops.flash_attn_bwd(q, k, v, o, do, L, workspace=ws, grads=grads, opts=opts)
torch.cuda.synchronize()
print("done", opts, iters)
