#!/usr/bin/env python3
"""Times the fp32 d = 64 backward (two kernels / one-pass forced / the library's choice) at launches below and at the size of the chip:
the measurement behind the one-pass kernel's dispatch rule (profiles/r04_onepass_f32_launch_sizes.txt).  usage: python tools/sweep_onepass_f32.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import check_onepass_f32 as c
for causal in (False, True):
    for (B, H, N) in ((8, 8, 256), (8, 8, 512), (4, 8, 512), (2, 8, 1024), (4, 8, 1024), (2, 8, 2048), (1, 8, 2048), (1, 8, 4096), (16, 8, 512), (16, 8, 256), (32, 8, 256), (1, 4, 8192), (1, 8, 1024), (5, 8, 2048)):
        c.timeit(B, H, N, causal=causal, iters=10)
