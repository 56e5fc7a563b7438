#!/usr/bin/env python3
"""One-off randomized sweep of the device path against the CPU oracle (bug hunting, not part of the test suite):
random dtype / head dim / N / batch*head / causal / variant / layout / key mask / dropout.  Prints every failure.
usage: python tools/fuzz_gpu.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle
from flash_attention_minitorch_amd import device_ops as dev, _lib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mx = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))
bad = 0
t0 = time.time()
for ci in range(cases):
    dtype = rng.choice(["bf16", "bf16", "f32"])
    d = int(rng.choice([32, 64, 64, 128]))
    kind = rng.integers(0, 6)
    N = int([rng.integers(1, 40), rng.integers(40, 200), 64 * rng.integers(1, 7), 128 * rng.integers(1, 9), rng.integers(200, 1300),
             256 * rng.integers(1, 9)][kind])
    B, H = int(rng.integers(1, 3)), int(rng.choice([1, 2, 3, 4, 8]))
    causal = bool(rng.integers(0, 2))
    variant = int(rng.choice([_lib.FA_VARIANT_FA1, _lib.FA_VARIANT_FA2]))
    mode = rng.choice(["plain", "plain", "bnhd", "mask", "dropout"])
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    arrs = [((rng.random((B, H, N, d), dtype=np.float32) - 0.5) * 2).astype(np.float32) for _ in range(4)]
    if dtype == "bf16":
        arrs = [oracle.bf16_round(a) for a in arrs]
    # bf16: P and dS enter the second MFMA of each product as bf16 (2^-9 relative).  Rows with many keys average that out
    # (<= 1e-3); rows with few keys do not -- early causal rows, N < 64, heavily masked rows (tests/test_gpu_parity.py header)
    few_keys = causal or N < 64 or mode == "mask"
    # (the causal figure is a ~3-sigma tail of zero-mean rounding noise on key 0's dV: 4.0-4.1e-3 appears about once in 300 cases)
    tol = 1e-3 if dtype == "bf16" else 1e-4   # round 2: no few-key exemption (split P / dS fragments on such rows)
    t = [torch.from_numpy(a).to("cuda", tdt) for a in arrs]
    # plain mode also draws per-call kernel options (fa_mi355x_*_ex): every admissible value must give the same results within tol
    # (0 = launch-size default; key 0 = 5 / keys 1, 2 = 3 force the slot kernels, under the causal mask their causal builds when
    # N % 256 == 0; key 7: paired / ranked block order of those builds)
    opts = None
    if mode == "plain" and rng.random() < 0.6:
        opts = (int(rng.choice([0, 3, 4, 5])), int(rng.choice([0, 2, 3])), int(rng.choice([0, 2, 3])), 0, int(rng.choice([0, 0, 1, 4, 5])),
                int(rng.integers(0, 2)), 0, int(rng.integers(0, 3)))   # (every value the product library accepts, round 3)
    desc = (ci, dtype, d, N, B, H, causal, variant, mode, opts)
    try:
        if mode == "plain":
            o, l, m = dev.flash_attn_fwd(*t[:3], causal, variant, opts=opts)
            g = dev.flash_attn_bwd(*t[:3], o, t[3], l, m, causal, variant, opts=opts)
            ro, rL, _, _ = oracle.dense_attention_fw(*arrs[:3], causal)
            rg = oracle.dense_attention_bw(*arrs, causal)
        elif mode == "bnhd":
            tb = [x.permute(0, 2, 1, 3).contiguous() for x in t]
            o, l, m = dev.flash_attn_fwd_bnhd(*tb[:3], causal, variant)
            g = dev.flash_attn_bwd_bnhd(*tb[:3], o, tb[3], l, m, causal, variant)
            o = o.permute(0, 2, 1, 3)
            g = [x.permute(0, 2, 1, 3) for x in g]
            ro, rL, _, _ = oracle.dense_attention_fw(*arrs[:3], causal)
            rg = oracle.dense_attention_bw(*arrs, causal)
        elif mode == "mask":
            mask = np.where(rng.random((B, N)) < 0.25, -np.inf, 0.0).astype(np.float32)
            mask[:, 0] = 0.0 if rng.random() < 0.7 else -np.inf
            tm = torch.from_numpy(mask).cuda()
            o, l, m = dev.flash_attn_fwd_masked(*t[:3], tm, causal, variant)
            g = dev.flash_attn_bwd_masked(*t[:3], o, t[3], l, m, tm, causal, variant)
            ro, rL = oracle.masked_attention_fw(*arrs[:3], mask[:, None, :], causal)
            rg = oracle.masked_attention_bw(*arrs, mask[:, None, :], causal)
        else:
            rate, seed = float(rng.choice([0.1, 0.3, 0.5])), int(rng.integers(0, 2**31))
            scale = 1.0 / (1.0 - rate)
            keep = oracle.dropout_keep_mask(B * H, N, rate, seed)
            o, l, m = dev.flash_attn_fwd_dropout(*t[:3], rate, seed, scale, None, causal, variant)
            g = dev.flash_attn_bwd_dropout(*t[:3], o, t[3], l, m, rate, seed, scale, None, causal, variant)
            ro, rL = oracle.dropout_attention_fw(*arrs[:3], keep, scale, None, causal)
            rg = oracle.dropout_attention_bw(*arrs, keep, scale, None, causal)
            tol *= scale
        with np.errstate(divide="ignore", invalid="ignore"):
            L = (m.float().cpu().numpy() + np.log(l.float().cpu().numpy())) if variant == _lib.FA_VARIANT_FA1 else l.float().cpu().numpy()
        dead = np.isneginf(rL)
        errs = {"o": mx(o.float().cpu().numpy(), ro), "L": mx(np.where(dead, 0, L), np.where(dead, 0, rL))}
        for nm, a, b in zip(("dq", "dk", "dv"), g, rg):
            errs[nm] = mx(a.float().cpu().numpy(), b)
        fin = all(np.all(np.isfinite(x.float().cpu().numpy())) for x in (o, *g))
        if not fin or max(errs.values()) >= tol or not np.array_equal(np.isneginf(L), dead):
            bad += 1
            print("FAIL", desc, "finite" if fin else "NON-FINITE", {k: f"{v:.2e}" for k, v in errs.items()}, "tol", tol, flush=True)
    except Exception as e:
        bad += 1
        print("ERROR", desc, repr(e)[:300], flush=True)
    if ci % 25 == 24:
        print(f"... {ci + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done:", cases, "cases,", bad, "bad")
