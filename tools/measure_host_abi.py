#!/usr/bin/env python3
"""Times the drop-in host-pointer ABI (launch_flash_attn_fw / _bw through the reference-named shims), i.e. the
PCIe-inclusive rate of the reference's own calling convention, at the fp32 configs of BASELINE.json.  Never the
headline number (bench.py times device-resident tensors).  Three figures per call:
  *_ms        through the Python operator surface (CudaKernelOps.flash_attn2_fw / _bw): what a minitorch user sees, including
              NumPy's allocation of the result arrays (np.zeros pages are first touched inside the call: pinning faults them in)
  *_abi_ms    the C call alone (launch_flash_attn_fw / _bw on arrays that already exist and have been touched)
  and a pinned-copy rate of the box for scale (64 MiB H2D / D2H through torch)."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from flash_attention_minitorch_amd import CudaKernelOps as ops, _lib
from flash_attention_minitorch_amd.cuda_kernel_ops import _FW_ARGTYPES, _BW_ARGTYPES, _stream


def best(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


out = {}
p = torch.empty(16 * 2**20, dtype=torch.float32).pin_memory()
dd = torch.empty(16 * 2**20, dtype=torch.float32, device="cuda")
def h2d(): dd.copy_(p, non_blocking=True); torch.cuda.synchronize()
def d2h(): p.copy_(dd, non_blocking=True); torch.cuda.synchronize()
out["box_pinned_copy_GBps"] = {"h2d": round(64 * 2**20 / best(h2d, 5) / 1e6, 1), "d2h": round(64 * 2**20 / best(d2h, 5) / 1e6, 1)}
fw = _lib.load("flash_attn2_fw.so").launch_flash_attn_fw
bw = _lib.load("flash_attn2_bw.so").launch_flash_attn_bw
fw.argtypes, fw.restype, bw.argtypes, bw.restype = _FW_ARGTYPES, None, _BW_ARGTYPES, None
for name, (B, H, N, d) in {"c1_B8H8N1024d64": (8, 8, 1024, 64), "c2_B8H8N2048d64": (8, 8, 2048, 64),
                           "M_B8H8N4096d64": (8, 8, 4096, 64)}.items():
    rng = np.random.default_rng(0)
    q, k, v, do = (rng.uniform(-1, 1, (B, H, N, d)).astype(np.float32) for _ in range(4))
    o, l, m = ops.flash_attn2_fw(q, k, v, False)  # warm-up (device arena, code object load)
    ops.flash_attn2_bw(q, k, v, o, do, l, m, False)
    fw_ms = best(lambda: ops.flash_attn2_fw(q, k, v, False))
    bw_ms = best(lambda: ops.flash_attn2_bw(q, k, v, o, do, l, m, False))
    f = lambda a: a.reshape(-1)
    ob, lb, mb = np.ones(B * H * N * d, np.float32), np.ones(B * H * N, np.float32), np.ones(B * H * N, np.float32)
    g = [np.ones(B * H * N * d, np.float32) for _ in range(3)]
    st = _stream()
    fw_abi = best(lambda: fw(f(q), f(k), f(v), ob, lb, mb, B * H, N, d, False, st))
    bw_abi = best(lambda: bw(f(q), f(k), f(v), f(o), f(do), g[0], g[1], g[2], f(l), f(m), B * H, N, d, False, st))
    fl = B * H * N * N * d
    mib_fw = (4 * B * H * N * d * 4 + 2 * B * H * N * 4) / 2**20
    mib_bw = (8 * B * H * N * d * 4 + 2 * B * H * N * 4) / 2**20
    out[name] = {"fw_ms": round(fw_ms, 2), "bw_ms": round(bw_ms, 2), "fw_abi_ms": round(fw_abi, 2), "bw_abi_ms": round(bw_abi, 2),
                 "fw_abi_GBps": round(mib_fw * 2**20 / fw_abi / 1e6, 1), "bw_abi_GBps": round(mib_bw * 2**20 / bw_abi / 1e6, 1),
                 "fw_abi_TFLOPs": round(4 * fl / fw_abi / 1e9, 2), "bw_abi_TFLOPs": round(10 * fl / bw_abi / 1e9, 2),
                 "bytes_moved_fw_MiB": round(mib_fw, 1), "bytes_moved_bw_MiB": round(mib_bw, 1)}
print(json.dumps(out, indent=1))
