#!/usr/bin/env python3
"""Phase shares of the one-pass backward (diagnostic build with s_memtime stamps; read the SHARES, not the run time).
usage: fused_stamps.py [32 | 33]   (33: the same without hand-off memory traffic)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import os as _os
_os.environ["FA_MI355X_DIAG"] = "1"   # tools use the diagnostic build (set_tuning, stamps, ablations)
from flash_attention_minitorch_amd import device_ops, _lib
B, H, N, d = 8, 8, 4096, 64
BH = B * H
mk = lambda: ((torch.rand((BH, N, d), device="cuda") - 0.5) * 2).to(torch.bfloat16)
q, k, v, do = mk(), mk(), mk(), mk()
o, L, _ = device_ops.flash_attn_fwd(q, k, v)
core = _lib.core()
core.fa_mi355x_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 32
core.fa_mi355x_set_tuning(5, mode)
ws = device_ops.bwd_workspace(q)
for _ in range(3):
    device_ops.flash_attn_bwd(q, k, v, o, do, L, workspace=ws)
torch.cuda.synchronize()
buf = np.zeros(8 * 8192, dtype=np.uint64)
_lib.check(core.fa_mi355x_debug_phase_cycles(buf.ctypes.data, buf.size))
core.fa_mi355x_set_tuning(5, 0)
a = buf.reshape(8192, 8)[:2048].astype(np.float64)
ph, life, real = a[:, :5], a[:, 6], a[:, 7]
ok = real > 0
tot = ph.sum(axis=1)
npairs_total = 4 * (N // 64)
print("mode", mode, "clock GHz:", round(float(np.median(life[ok] / real[ok])) * 0.1, 3), "lifetime cycles (median):", np.median(life[ok]),
      "stamped:", np.median(tot[ok]))
names = ["top of pair (finish/signal/fetch/DMA issue)", "period A", "period B", "vmcnt(0) wait", "barrier"]
for j, nm in enumerate(names):
    print(f"{nm:46s} {100 * np.median(ph[ok, j] / tot[ok]):5.1f} %  ({np.median(ph[ok, j]) / npairs_total:8.1f} cycles per pair)   max-wave {np.max(ph[ok, j]) / npairs_total:8.1f}")
sp = a[ok, 5]
print("spin iterations: total", sp.sum(), "mean per wave", sp.mean());
print("spin iterations per wave: median", np.median(sp), "max", sp.max(), "waves that ever spun:", int((sp > 0).sum()), "of", int(ok.sum()))
for wv in range(8):
    sel = ok & (np.arange(2048) % 8 == wv)
    print(f"wave {wv}: " + "  ".join(f"{np.median(ph[sel, j]) / npairs_total:7.1f}" for j in range(5)))
