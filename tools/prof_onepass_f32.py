#!/usr/bin/env python3
"""Runs the fp32 fw+bw of BASELINE configs[2] (B=8 H=8 N=2048 d=64, FA-1 side outputs) a few times: the workload of the rocprofv3
passes in tools/profile_onepass_f32.sh.  usage: python tools/prof_onepass_f32.py [iters] [opts like 0,0,0,0,4] [causal]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flash_attention_minitorch_amd import _lib, device_ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
opts = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 and sys.argv[2] != "-" else None
causal = len(sys.argv) > 3 and sys.argv[3] == "causal"
B, H, N, d = (int(x) for x in os.environ.get("FA_SHAPE", "8,8,2048,64").split(","))
gen = torch.Generator(device="cuda").manual_seed(1)
mk = lambda: (torch.rand((B * H, N, d), device="cuda", generator=gen) - 0.5) * 2
q, k, v, do = mk(), mk(), mk(), mk()
ws = device_ops.bwd_workspace(q)
grads = tuple(torch.empty((B * H, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
for _ in range(iters):
    o, l, m = device_ops.flash_attn_fwd(q, k, v, causal, _lib.FA_VARIANT_FA1)
    device_ops.flash_attn_bwd(q, k, v, o, do, l, m, causal, _lib.FA_VARIANT_FA1, workspace=ws, grads=grads, opts=opts)
torch.cuda.synchronize()
print("done", iters, opts, causal)
