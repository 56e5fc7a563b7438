#!/usr/bin/env python3
"""Forward only: slot-interleaved kernel vs the phased one (tuning key 1) at d = 128 and d = 64 shapes, with the max-abs
difference of their outputs.  usage: python tools/bench_fwd_d128.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ["FA_MI355X_DIAG"] = "1"   # tools use the diagnostic build (set_tuning, stamps, ablations)
from flash_attention_minitorch_amd import device_ops, _lib
core = _lib.core()
def t_ms(fn, iters=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best
for (B,H,N,d) in ((16,16,4096,128),(8,8,4096,64),(4,8,1000,128)):
    BH=B*H
    mk = lambda: ((torch.rand((BH, N, d), device="cuda") - 0.5) * 2).to(torch.bfloat16)
    q,k,v = mk(),mk(),mk()
    res={}
    for knob in (0,2):
        core.fa_mi355x_set_tuning(1, knob)
        o,L,_ = device_ops.flash_attn_fwd(q,k,v)
        res[knob]=(o.clone(),L.clone(), t_ms(lambda: device_ops.flash_attn_fwd(q,k,v,out=o,l=L)))
    core.fa_mi355x_set_tuning(1, 0)
    fl=4*BH*N*N*d
    print((B,H,N,d), "slot ms %.4f (%.0f TF)  phased ms %.4f (%.0f TF)  max|dO| %.2e max|dL| %.2e" % (res[0][2], fl/res[0][2]/1e9, res[2][2], fl/res[2][2]/1e9, (res[0][0]-res[2][0]).abs().max().item(), (res[0][1]-res[2][1]).abs().max().item()))
