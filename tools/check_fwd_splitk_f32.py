#!/usr/bin/env python3
"""GPU check of the fp32 split-key forward (fwd_splitk_f32_kernel, option 1 = 4; the default of small fp32 d = 64 launches) against the
fp64 oracle and the phased forward (option 1 = 2), plus timing of both.  usage: python tools/check_fwd_splitk_f32.py [--time]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from flash_attention_minitorch_amd import _lib, device_ops  # noqa: E402

SPLIT, PHASED = (0, 4), (0, 2)


def run(BH, N, causal, variant, seed=0, scale=1.0):
    rng = np.random.default_rng(seed)
    qf, kf, vf = ((scale * rng.uniform(-1, 1, (BH, N, 64))).astype(np.float32) for _ in range(3))
    tq, tk, tv = (torch.from_numpy(a).to("cuda") for a in (qf, kf, vf))
    o1, l1, m1 = device_ops.flash_attn_fwd(tq, tk, tv, causal, variant, opts=SPLIT)
    o2, l2, m2 = device_ops.flash_attn_fwd(tq, tk, tv, causal, variant, opts=PHASED)
    torch.cuda.synchronize()
    ro, rL, rm, rl = oracle.dense_attention_fw(qf, kf, vf, causal)
    eo = float(np.max(np.abs(o1.cpu().numpy() - ro)))
    if variant == _lib.FA_VARIANT_FA1:
        L1 = m1.cpu().numpy() + np.log(l1.cpu().numpy())
        em = float(np.max(np.abs(m1.cpu().numpy() - rm)))
    else:
        L1, em = l1.cpu().numpy(), 0.0
    eL = float(np.max(np.abs(L1 - rL)))
    d12 = float((o1 - o2).abs().max())
    ok = eo < 1e-4 * scale and eL < 1e-4 * max(1.0, scale * scale) and em < 1e-5 * max(1.0, scale * scale) and d12 < 1e-5 * scale
    print(("OK   " if ok else "FAIL ") + f"BH{BH} N{N}{' causal' if causal else ''} v{variant} x{scale}: |o-oracle|={eo:.2e} |L-oracle|={eL:.2e} |m-oracle|={em:.1e} "
          f"|split-phased|={d12:.1e}", flush=True)
    return ok


def timeit(B, H, N, causal, iters=30):
    gen = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: (torch.rand((B * H, N, 64), device="cuda", generator=gen) - 0.5) * 2
    q, k, v = mk(), mk(), mk()
    o, l, m = device_ops.flash_attn_fwd(q, k, v, causal, _lib.FA_VARIANT_FA1)
    res = {}
    for name, opts in (("phased", PHASED), ("split", SPLIT), ("default", None)):
        fn = lambda: device_ops.flash_attn_fwd(q, k, v, causal, _lib.FA_VARIANT_FA1, out=o, l=l, m=m, opts=opts)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters
    fl = 4.0 * B * H * N * N * 64 * (0.5 if causal else 1.0)
    print(f"time B{B} H{H} N{N}{' causal' if causal else ''} ({B * H * ((N + 127) // 128)} phased workgroups): " +
          "  ".join(f"{kk}={vv:.4f} ms ({fl / vv / 1e9:.0f} TF/s)" for kk, vv in res.items()), flush=True)


if __name__ == "__main__":
    good = True
    for causal in (False, True):
        for variant in (_lib.FA_VARIANT_FA1, _lib.FA_VARIANT_FA2):
            for BH, N in ((1, 128), (3, 129), (2, 200), (8, 1024), (5, 1000), (3, 33 * 32), (2, 2048), (7, 160)):
                good &= run(BH, N, causal, variant, seed=BH * 100 + N)
    good &= run(4, 512, True, _lib.FA_VARIANT_FA1, seed=9, scale=6.0)
    good &= run(4, 512, False, _lib.FA_VARIANT_FA2, seed=10, scale=6.0)
    print("ALL OK" if good else "SOME FAILED", flush=True)
    if "--time" in sys.argv:
        for causal in (False, True):
            for shp in ((1, 8, 1024), (1, 8, 2048), (2, 8, 1024), (2, 8, 2048), (4, 8, 1024), (4, 8, 2048), (1, 8, 4096), (8, 8, 1024), (8, 8, 2048), (16, 8, 1024)):
                timeit(*shp, causal)
    sys.exit(0 if good else 1)
