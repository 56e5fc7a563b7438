#!/usr/bin/env python3
"""Where do the slot kernels with the scale folded into their operand (round 3) stand against the oracle and against the phased
kernels (fp32 scaling), and which of the two does the DEFAULT call run (round 4: under the scale guard)?  Prints, per case, max-abs
error of o / L / dq / dk / dv vs the fp64 oracle for the default call, the forced slot kernels with the folded scale (option 8 = 1)
and the phased kernels, and the row of the worst error; `default = slot|phased`: which of them the default equals bitwise.
usage: python tools/check_prescale.py [scale ...]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import oracle
from gpu_util import oracle_heads, rand_u, to_np
from flash_attention_minitorch_amd import device_ops as dev

scales = [float(x) for x in sys.argv[1:]] or [1.0]
names = ("o", "L", "dq", "dk", "dv")
for scale in scales:
    for causal in (False, True):
        for BH, N in ((2, 256), (2, 512), (2, 1280), (2, 2048), (4, 4096)):
            rng = np.random.default_rng(7000 + N)
            arrs = [oracle.bf16_round(scale * rand_u(rng, (BH, N, 64))) for _ in range(2)] + \
                   [oracle.bf16_round(rand_u(rng, (BH, N, 64))) for _ in range(2)]
            t = [torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs]
            ref = oracle_heads(*arrs, causal, range(BH))
            k0 = 5 if causal else 0
            res = {}
            for tag, opts in (("default", None), ("slot", (k0, 3, 3, 0, 0, 0, 0, 2, 1)), ("phased", dev.OPTS_PHASED)):
                o, l, m = dev.flash_attn_fwd(*t[:3], causal=causal, opts=opts)
                g = dev.flash_attn_bwd(*t[:3], o, t[3], l, m, causal=causal, opts=opts)
                outs = [to_np(x) for x in (o, l) + tuple(g)]
                res[tag] = outs
                msg = []
                for nm, got in zip(names, outs):
                    e = np.abs(got.astype(np.float64) - ref[nm])
                    idx = np.unravel_index(np.argmax(e), e.shape)
                    msg.append(f"{nm} {e.max():.2e}@{idx[1]}")
                print(f"scale {scale} causal {int(causal)} BH {BH} N {N:5d} {tag:7s} " + "  ".join(msg), flush=True)
            eq = [tag for tag in ("slot", "phased") if all(np.array_equal(a, b) for a, b in zip(res["default"], res[tag]))]
            print(f"scale {scale} causal {int(causal)} BH {BH} N {N:5d} default = {'|'.join(eq) or 'neither (a mix of kernels)'}", flush=True)
