#!/usr/bin/env python3
"""Runs a DIAGNOSTIC dK/dV build (in-kernel s_memtime stamps) at the metric shape and prints the share of each
loop phase in a wave's life.  Read the SHARES, not the run time (the stamps forbid overlaps the real kernel has).
usage: phase_cycles.py [9 | 93]     9 = phased kernel, 93 = slot-interleaved kernel"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import os as _os
_os.environ["FA_MI355X_DIAG"] = "1"   # tools use the diagnostic build (set_tuning, stamps, ablations)
from flash_attention_minitorch_amd import device_ops, _lib
B, H, N, d = map(int, os.environ.get("FA_SHAPE", "8,8,4096,64").split(","))   # FA_SHAPE=16,16,4096,128: configs[3]'s shape (forward stamps only)
BH = B * H
mk = lambda: ((torch.rand((BH, N, d), device="cuda") - 0.5) * 2).to(torch.bfloat16)
q, k, v, do = mk(), mk(), mk(), mk()
o, L, _ = device_ops.flash_attn_fwd(q, k, v)
core = _lib.core()
core.fa_mi355x_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 9
KEY = 2 if MODE == 293 else (1 if MODE == 193 else 0)   # 293 / 193: the slot-interleaved dQ / forward kernel's stamps
core.fa_mi355x_set_tuning(KEY, 93 if MODE in (293, 193) else (193 if MODE == 393 else MODE))   # 393: continuous dK/dV kernel
for _ in range(3):
    if KEY == 1:
        device_ops.flash_attn_fwd(q, k, v)
        continue
    device_ops.flash_attn_bwd(q, k, v, o, do, L, stages=device_ops.STAGE_PREP | (device_ops.STAGE_DQ if KEY == 2 else device_ops.STAGE_DKDV))
torch.cuda.synchronize()
buf = np.zeros(8 * 8192, dtype=np.uint64)
_lib.check(core.fa_mi355x_debug_phase_cycles(buf.ctypes.data, buf.size))
core.fa_mi355x_set_tuning(KEY, 0)
allc = buf.reshape(8192, 8).astype(np.float64)
ph = allc[:, :6]
tot = ph.sum(axis=1)
life, real = allc[:, 6], allc[:, 7]
ok = real > 0
print("in-kernel clock GHz (median over waves):", round(float(np.median(life[ok] / real[ok])) * 0.1, 3),
      " wave lifetime cycles:", np.median(life[ok]), " stamped cycles:", np.median(tot))
if MODE == 193:
    names = ["prologue (Q load, first stage DMA + barrier)", "periods (MFMA slots)", "vmcnt(0) wait for own DMA", "barrier",
             "epilogue (normalise, store O and L)", "-"]
elif MODE == 393:
    # [4] and [5] lie inside / in front of [0]: [4] = first instruction -> fragment loads issued (before the prologue stamp),
    # [5] = prologue stamp -> fragments and stage 0 landed (the rest of [0] is the barrier); lifetime runs from the first instruction
    # to the dK / dV stores drained, so lifetime - sum([0..3]) - [4] = the epilogue
    names = ["prologue (K/V fragments, first stage DMA + barrier)", "periods (MFMA slots)", "vmcnt(0) wait for own DMA", "barrier",
             "(set-up: first instruction -> K/V loads issued)", "(of the prologue: wait for fragments + stage 0)"]
elif MODE == 293:
    names = ["prologue (first stage DMA + barrier)", "periods (MFMA slots)", "vmcnt(0) wait for own DMA", "barrier", "-", "-"]
elif MODE == 93:
    names = ["stage_load issue", "prologue period (8 MFMA: S,dP of sub 0)", "periods 1-3 (48 MFMA + VALU)",
             "last period (8 MFMA + VALU of sub 3)", "stage_store (incl. vmcnt wait)", "barrier"]
else:
    names = ["stage_load issue", "row reads + MFMA S,dP issue", "exp/fma/mul/pack (incl. MFMA drain)",
             "tr reads + MFMA dV,dK issue", "stage_store (incl. vmcnt wait)", "barrier"]
if MODE == 393:
    tot = ph[:, :4].sum(axis=1)
    epi = life - tot - ph[:, 4]
    print("set-up cycles (median):", np.median(ph[:, 4]), " prologue load wait:", np.median(ph[:, 5]), " prologue total:", np.median(ph[:, 0]),
          " epilogue (stores drained):", np.median(epi), " lifetime:", np.median(life))
print("waves:", len(tot), "per 128-row stage:", np.median(tot) / 32, "per 32-query sub-slice:", np.median(tot) / 128)
for j, nm in enumerate(names):
    print(f"{nm:42s} {100 * np.median(ph[:, j] / tot):5.1f} %   ({np.median(ph[:, j]) / 32:7.1f} cycles per stage)")
