#!/usr/bin/env python3
"""A/B of the tiled dQ build (several consecutive query blocks per workgroup) against one block per workgroup (option 5 = 1), same
process, interleaved rounds; the dQ launch with its folded preprocess (stages PREP | DQ).  Also checks that both give the same dQ
bit for bit.  usage: python tools/ab_dq_tiles.py [B H N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_minitorch_amd import device_ops as dev  # noqa: E402

B, H, N = (8, 8, 4096) if len(sys.argv) < 4 else map(int, sys.argv[1:4])
d, BH = 64, B * H
gen = torch.Generator(device="cuda").manual_seed(3)
q, k, v, do = (((torch.rand((BH, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16) for _ in range(4))
o, L, _ = dev.flash_attn_fwd(q, k, v)
ws = dev.bwd_workspace(q)
grads = tuple(torch.empty((BH, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
ST = dev.STAGE_PREP | dev.STAGE_DQ
OPTS = {"tiled": None, "one_block": (0, 0, 0, 0, 0, 1)}


def run(opts):
    dev.flash_attn_bwd(q, k, v, o, do, L, None, False, workspace=ws, grads=grads, stages=ST, opts=opts)


def t_ms(opts, iters=40):
    for _ in range(5):
        run(opts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run(opts)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


outs = {}
for name, opts in OPTS.items():
    run(opts)
    torch.cuda.synchronize()
    outs[name] = grads[0].clone()
print("bitwise equal dQ:", bool(torch.equal(outs["tiled"], outs["one_block"])), " max|diff|", float((outs["tiled"] - outs["one_block"]).abs().max()))
for r in range(3):
    print({name: round(t_ms(opts), 4) for name, opts in OPTS.items()})
