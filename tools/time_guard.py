#!/usr/bin/env python3
"""Times the separate scale-guard pass (fa_mi355x_scale_guard) at the metric shape and at configs[3]'s.  usage: python tools/time_guard.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_attention_minitorch_amd import device_ops as dev

for BH, N, d in ((64, 4096, 64), (256, 4096, 128), (2048, 4096, 128)):
    q = torch.randn((BH, N, d), device="cuda").to(torch.bfloat16)
    k = torch.randn((BH, N, d), device="cuda").to(torch.bfloat16)
    g = dev.scale_guard(q, k)
    for _ in range(5):
        dev.scale_guard(q, k, out=g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        dev.scale_guard(q, k, out=g)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    gb = 2 * q.numel() * 2 / 1e9
    print(f"BH {BH} N {N} d {d}: {ms * 1e3:.1f} us  {gb / ms:.0f} GB/s", flush=True)
