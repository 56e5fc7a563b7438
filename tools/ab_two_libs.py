#!/usr/bin/env python3
"""A/B of two builds of the core library in ONE process on one device (box-to-box spread is larger than most kernel changes):
library A = the in-tree product build, library B = another build of csrc/fa_api.hip (e.g. with a -D switch), both loaded with ctypes
and timed alternately on the same tensors through fa_mi355x_fwd_ex / fa_mi355x_bwd_stages_ex-free entry points.
usage: python tools/ab_two_libs.py path/to/libB.so [fwd|bwd|fwdg|bwdg|fwd32|bwd32] [causal]   (fwdg / bwdg: the guarded default calls; *32: fp32 at configs[2])"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flash_attention_minitorch_amd import _lib, device_ops  # noqa: E402


def load(path):
    lib = ctypes.CDLL(path)
    src = _lib.core()
    for name in ("fa_mi355x_fwd_ex", "fa_mi355x_bwd_ex", "fa_mi355x_fwd_guarded", "fa_mi355x_bwd_guarded"):
        f, g = getattr(lib, name), getattr(src, name)
        f.argtypes, f.restype = g.argtypes, g.restype
    return lib


def main():
    libs = {"A": _lib.core(), "B": load(sys.argv[1])}
    what = sys.argv[2] if len(sys.argv) > 2 else "fwd"
    causal = int(len(sys.argv) > 3 and sys.argv[3] == "causal")
    B, H, N, d = 8, 8, 4096, 64
    gen = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: ((torch.rand((B * H, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o = torch.empty((B * H, N, d), dtype=torch.float32, device="cuda")
    L = torch.empty((B * H, N), dtype=torch.float32, device="cuda")
    grads = [torch.empty_like(o) for _ in range(3)]
    ws = device_ops.bwd_workspace(q)
    arr, cnt = _lib.opts_array(device_ops.OPTS_FOLDED_SCALE)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fwd(lib):
        rc = lib.fa_mi355x_fwd_ex(p(q), p(k), p(v), p(o), p(L), None, B * H, N, d, causal, _lib.FA_VARIANT_FA2, _lib.FA_DTYPE_BF16, arr, cnt, st)
        assert rc == 0, rc

    def bwd(lib):
        rc = lib.fa_mi355x_bwd_ex(p(q), p(k), p(v), p(o), p(do), p(grads[0]), p(grads[1]), p(grads[2]), p(L), None, p(ws), B * H, N, d,
                                  causal, _lib.FA_VARIANT_FA2, _lib.FA_DTYPE_BF16, device_ops.STAGE_ALL, arr, cnt, st)
        assert rc == 0, rc

    guard = device_ops.new_guard(q)
    arr0, cnt0 = _lib.opts_array(None)

    def fwdg(lib):   # the default call of device_ops / bench.py: the forward fills the scale guard inside its own launch
        rc = lib.fa_mi355x_fwd_guarded(p(q), p(k), p(v), p(o), p(L), None, B * H, 1, N, d, 0, 0.0, causal, _lib.FA_VARIANT_FA2, _lib.FA_DTYPE_BF16,
                                       arr0, cnt0, p(guard), 1, st)
        assert rc == 0, rc

    def bwdg(lib):
        rc = lib.fa_mi355x_bwd_guarded(p(q), p(k), p(v), p(o), p(do), p(grads[0]), p(grads[1]), p(grads[2]), p(L), None, p(ws), B * H, 1, N, d, 0,
                                       0.0, causal, _lib.FA_VARIANT_FA2, _lib.FA_DTYPE_BF16, device_ops.STAGE_ALL, arr0, cnt0, p(guard), st)
        assert rc == 0, rc

    fwd(libs["A"])
    fwdg(libs["A"])
    # fp32 (the reference's dtype) at BASELINE configs[2]: B=8 H=8 N=2048 d=64, FA-1 side outputs
    N32 = 2048
    q32, k32, v32, do32 = ((torch.rand((B * H, N32, d), device="cuda", generator=gen) - 0.5) * 2 for _ in range(4))
    o32 = torch.empty((B * H, N32, d), dtype=torch.float32, device="cuda")
    l32, m32 = (torch.empty((B * H, N32), dtype=torch.float32, device="cuda") for _ in range(2))
    g32 = [torch.empty_like(o32) for _ in range(3)]
    ws32 = device_ops.bwd_workspace(q32)

    def fwd32(lib):
        rc = lib.fa_mi355x_fwd_ex(p(q32), p(k32), p(v32), p(o32), p(l32), p(m32), B * H, N32, d, causal, _lib.FA_VARIANT_FA1, _lib.FA_DTYPE_F32,
                                  arr0, cnt0, st)
        assert rc == 0, rc

    def bwd32(lib):
        rc = lib.fa_mi355x_bwd_ex(p(q32), p(k32), p(v32), p(o32), p(do32), p(g32[0]), p(g32[1]), p(g32[2]), p(l32), p(m32), p(ws32), B * H, N32, d,
                                  causal, _lib.FA_VARIANT_FA1, _lib.FA_DTYPE_F32, device_ops.STAGE_ALL, arr0, cnt0, st)
        assert rc == 0, rc

    fwd32(libs["A"])
    fn = {"fwd": fwd, "bwd": bwd, "fwdg": fwdg, "bwdg": bwdg, "fwd32": fwd32, "bwd32": bwd32}[what]
    res = {"A": [], "B": []}
    for rnd in range(6):
        for name in ("A", "B"):
            for _ in range(10):
                fn(libs[name])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn(libs[name])
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 50)
    for name in ("A", "B"):
        r = res[name]
        print(f"{what} {'causal ' if causal else ''}{name}: " + " ".join(f"{x:.4f}" for x in r) + f"  | median of last 4 = {sorted(r[2:])[1]:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
