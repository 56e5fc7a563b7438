#!/usr/bin/env python3
"""Times the drop-in host-pointer ABI (launch_flash_attn_fw / _bw through the reference-named shims), i.e. the
PCIe-inclusive rate of the reference's own calling convention, at the fp32 configs of BASELINE.json.  Never the
headline number (bench.py times device-resident tensors)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flash_attention_minitorch_amd import CudaKernelOps as ops

out = {}
for name, (B, H, N, d) in {"c1_B8H8N1024d64": (8, 8, 1024, 64), "c2_B8H8N2048d64": (8, 8, 2048, 64),
                           "M_B8H8N4096d64": (8, 8, 4096, 64)}.items():
    rng = np.random.default_rng(0)
    q, k, v, do = (rng.uniform(-1, 1, (B, H, N, d)).astype(np.float32) for _ in range(4))
    ops.flash_attn2_fw(q, k, v, False)  # warm-up (device arena, code object load)
    t0 = time.perf_counter(); o, l, m = ops.flash_attn2_fw(q, k, v, False); t1 = time.perf_counter()
    ops.flash_attn2_bw(q, k, v, o, do, l, m, False)
    t2 = time.perf_counter(); ops.flash_attn2_bw(q, k, v, o, do, l, m, False); t3 = time.perf_counter()
    fl = B * H * N * N * d
    out[name] = {"fw_ms": round((t1 - t0) * 1e3, 2), "bw_ms": round((t3 - t2) * 1e3, 2),
                 "fw_TFLOPs": round(4 * fl / (t1 - t0) / 1e12, 2), "bw_TFLOPs": round(10 * fl / (t3 - t2) / 1e12, 2),
                 "bytes_moved_fw_MiB": round((4 * B * H * N * d * 4 + 2 * B * H * N * 4) / 2**20, 1)}
print(json.dumps(out, indent=1))
