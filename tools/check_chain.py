#!/usr/bin/env python3
"""GPU check of the chained one-pass backward (bwd_chain_kernel, option 4 = 3) against the fp64 oracle and the two-kernel
backward, plus timing.
usage: python tools/check_chain.py [--time] [--abl] [--shapes]"""
import os
import sys

os.environ["FA_MI355X_DIAG"] = "1"   # the kernel lives in the diagnostic build

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from flash_attention_minitorch_amd import _lib, device_ops  # noqa: E402

CHAIN = (0, 0, 0, 0, 3)


def run(B, H, N, d=64, seed=0, heads=None, scale=1.0):
    rng = np.random.default_rng(seed)
    qf, kf, vf, dof = (oracle.bf16_round((scale * rng.uniform(-1, 1, (B * H, N, d))).astype(np.float32)) for _ in range(4))
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in (qf, kf, vf, dof))
    o, L, _ = device_ops.flash_attn_fwd(tq, tk, tv, causal=False)
    ws = device_ops.bwd_workspace(tq, CHAIN)
    g2 = [t.clone() for t in device_ops.flash_attn_bwd(tq, tk, tv, o, tdo, L, workspace=ws)]
    g1 = [t.clone() for t in device_ops.flash_attn_bwd(tq, tk, tv, o, tdo, L, workspace=ws, opts=CHAIN)]
    g1b = [t.clone() for t in device_ops.flash_attn_bwd(tq, tk, tv, o, tdo, L, workspace=ws, opts=CHAIN)]
    torch.cuda.synchronize()
    names = ("dq", "dk", "dv")
    msg = [f"B{B} H{H} N{N}"]
    ok = True
    for n_, a, b, c in zip(names, g1, g2, g1b):
        dsplit = float((a - b).abs().max())
        drep = float((a - c).abs().max())
        msg.append(f"{n_}: |chain-split|={dsplit:.2e} |rerun|={drep:.1e}")
        ok &= dsplit < 2e-3 * scale * scale and bool(torch.isfinite(a).all()) and drep < 1e-5 * scale * scale
    hs = range(B * H) if heads is None else heads
    worst = 0.0
    for hh in hs:
        refs = oracle.dense_attention_bw(qf[hh:hh + 1], kf[hh:hh + 1], vf[hh:hh + 1], dof[hh:hh + 1])
        for n_, a, ref in zip(names, g1, refs):
            e = float(np.max(np.abs(a[hh].cpu().numpy() - ref[0])))
            worst = max(worst, e)
            if e > 1e-3 * scale * scale:
                ok = False
                msg.append(f"  head {hh} {n_} err {e:.2e} !!")
    msg.append(f"oracle max err {worst:.2e}")
    print(("OK   " if ok else "FAIL ") + " | ".join(msg), flush=True)
    return ok


def timeit(B, H, N, d=64, iters=30, abl=False):
    gen = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: ((torch.rand((B * H, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal=False)
    ws = device_ops.bwd_workspace(q, CHAIN)
    grads = tuple(torch.empty((B * H, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
    variants = [("split", None), ("chain", CHAIN), ("split2", None), ("chain2", CHAIN)]
    if abl:
        variants += [("chain_notiles", (0, 0, 0, 0, 3, 1)), ("chain_nodq", (0, 0, 0, 0, 3, 2)), ("chain_dkdvonly", (0, 0, 0, 0, 3, 3)),
                     ("chain_nostores", (0, 0, 0, 0, 3, 64)), ("chain_noloads", (0, 0, 0, 0, 3, 128)),
                     ("noatomics", (0, 0, 0, 0, 3, 256)), ("nt_stores", (0, 0, 0, 0, 3, 512)), ("nt_loads", (0, 0, 0, 0, 3, 1024)),
                     ("nt_both", (0, 0, 0, 0, 3, 1536)), ("one_spot", (0, 0, 0, 0, 3, 2048)), ("one_spot_noatomics", (0, 0, 0, 0, 3, 2304)),
                     ("loads_one_spot_noatomics", (0, 0, 0, 0, 3, 4352)), ("stores_one_spot_noatomics", (0, 0, 0, 0, 3, 8448))]
    res = {}
    for name, opts in variants:
        for _ in range(10):
            device_ops.flash_attn_bwd(q, k, v, o, do, L, workspace=ws, grads=grads, opts=opts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            device_ops.flash_attn_bwd(q, k, v, o, do, L, workspace=ws, grads=grads, opts=opts)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters
    fl = 10.0 * B * H * N * N * d
    print(f"time B{B} H{H} N{N}: " + "  ".join(f"{k}={v:.4f} ms ({fl / v / 1e9:.0f} TF/s)" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    good = True
    good &= run(1, 1, 256)                                 # one key block, one chain: direct store
    good &= run(1, 2, 512)                                 # two chains of one block: atomics only
    good &= run(2, 3, 1024, seed=3)
    good &= run(1, 20, 768, seed=4, heads=[0, 7, 19])
    good &= run(8, 8, 4096, seed=5, heads=[0, 13, 63])     # the metric shape: 4 chains of 4 blocks
    good &= run(3, 7, 2048, seed=6, heads=[0, 20])
    good &= run(32, 8, 1024, seed=7, heads=[0, 100, 255])  # a whole head per workgroup: sums in the slab, plain final store, no atomics
    good &= run(16, 8, 2048, seed=8, heads=[3, 127])       # two chains of four
    print("ALL OK" if good else "SOME FAILED", flush=True)
    if "--time" in sys.argv:
        timeit(8, 8, 4096, abl="--abl" in sys.argv)
        if "--shapes" in sys.argv:
            timeit(4, 8, 8192)
            timeit(16, 8, 2048)
            timeit(32, 8, 1024)
            timeit(64, 8, 512)
            timeit(32, 8, 4096)
            timeit(2, 8, 16384)
    sys.exit(0 if good else 1)
