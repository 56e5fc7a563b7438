import os, sys, torch
sys.path.insert(0, os.getcwd())
from flash_attention_minitorch_amd import device_ops, _lib
def t_ms(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for causal in (False, True):
    for (B,H,N) in ((1,8,1024),(1,8,2048),(2,8,1024),(2,8,2048),(4,8,1024),(4,8,2048),(1,8,4096),(8,8,1024),(8,8,2048)):
        gen = torch.Generator(device="cuda").manual_seed(1)
        mk = lambda: (torch.rand((B*H, N, 64), device="cuda", generator=gen) - 0.5) * 2
        q,k,v,do = mk(),mk(),mk(),mk()
        o,l,m = device_ops.flash_attn_fwd(q,k,v,causal,_lib.FA_VARIANT_FA1)
        ws = device_ops.bwd_workspace(q); g = tuple(torch.empty_like(o) for _ in range(3))
        tf = t_ms(lambda: device_ops.flash_attn_fwd(q,k,v,causal,_lib.FA_VARIANT_FA1,out=o,l=l,m=m))
        tb = t_ms(lambda: device_ops.flash_attn_bwd(q,k,v,o,do,l,m,causal,_lib.FA_VARIANT_FA1,workspace=ws,grads=g))
        cf = 0.5 if causal else 1.0
        fl = B*H*N*N*64*cf
        print(f"B{B} H{H} N{N}{' causal' if causal else ''}: fw {tf:.4f} ms ({4*fl/tf/1e9:.0f} TF/s)  bw {tb:.4f} ms ({10*fl/tb/1e9:.0f} TF/s)", flush=True)
