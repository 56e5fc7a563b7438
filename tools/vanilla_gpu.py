"""GPU "vanilla" (materialised-S) attention in plain torch-ROCm: the comparator of the reference's own flash-vs-vanilla tests and
timing harness (softmax((q @ kT) / sqrt(d) + mask) @ v with mask = -FLT_MAX * triu(ones, 1):
kernel_tests/test_flashattn_fw.py:18-20,64-72, test_flashattn_time.py:38-62, minitorch/modules_transfomer.py:123-127).
Measurement / test infrastructure only: nothing in the product path imports it."""
import numpy as np
import torch


def causal_mask(n, device, dtype=torch.float32):
    # kernel_tests/test_flashattn_fw.py:18-20: -finfo(float32).max on the strict upper triangle
    return torch.triu(torch.full((n, n), -float(np.finfo(np.float32).max), device=device, dtype=dtype), 1)


def vanilla_attention(q, k, v, causal, softmax_dtype=torch.float32):
    """q, k, v: (..., N, d).  Scores and softmax in ``softmax_dtype``; the two matmuls in the inputs' dtype."""
    n, d = q.shape[-2], q.shape[-1]
    s = torch.matmul(q, k.transpose(-1, -2)).to(softmax_dtype) * (d ** -0.5)
    if causal:
        s = s + causal_mask(n, q.device, softmax_dtype)
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p.to(v.dtype), v)


def vanilla_fw_bw(q, k, v, do, causal):
    """Returns (o, dq, dk, dv) by autograd through the vanilla forward."""
    qq, kk, vv = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    o = vanilla_attention(qq, kk, vv, causal)
    o.backward(do.to(o.dtype))
    return o.detach(), qq.grad, kk.grad, vv.grad


_FUSED_KIND = [None]


def _fused_mask_softmax(s, pad_mask_bool):
    """softmax(s + padding mask) over the last axis as ONE kernel where torch-ROCm has one (aten::_masked_softmax, the fused
    mask + softmax of nn.MultiheadAttention's fast path); otherwise mask fill + softmax.  Forward-only calls pass the (B, N)
    key-padding mask (mask_type 1); under autograd the op's backward wants a mask of the scores' own shape (mask_type 2)."""
    if _FUSED_KIND[0] in (None, "aten::_masked_softmax"):
        try:
            if s.requires_grad:
                full = pad_mask_bool[:, None, None, :].expand(s.shape).contiguous()
                p = torch._masked_softmax(s, full, -1, 2)
            else:
                p = torch._masked_softmax(s, pad_mask_bool, -1, 1)
            _FUSED_KIND[0] = "aten::_masked_softmax"
            return p
        except Exception:
            _FUSED_KIND[0] = "torch.softmax(masked_fill(s)): no usable fused masked softmax in this torch build"
    return torch.softmax(s.masked_fill(pad_mask_bool[:, None, None, :], float("-inf")), dim=-1)


def fused_softmax_kind():
    return _FUSED_KIND[0] or "not run"


def fused_softmax_attention(q, k, v, causal):
    """The reference's use_fused_kernel path (minitorch/modules_transfomer.py:131-136): ((q @ kT) / sqrt(d) + M).attn_softmax(mask) @ v
    with the scores materialised, M the causal mask (when causal) and `mask` a [B, 1, 1, N] padding mask that is all zeros there.
    q, k, v: (B, H, N, d)."""
    B, H, n, d = q.shape
    s = torch.matmul(q, k.transpose(-1, -2)).float() * (d ** -0.5)
    if causal:
        s = s + causal_mask(n, q.device)
    pad = torch.zeros((B, n), dtype=torch.bool, device=q.device)
    p = _fused_mask_softmax(s, pad)
    return torch.matmul(p.to(v.dtype), v)


def fused_softmax_fw_bw(q, k, v, do, causal):
    qq, kk, vv = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    o = fused_softmax_attention(qq, kk, vv, causal)
    o.backward(do.to(o.dtype))
    return o.detach(), qq.grad, kk.grad, vv.grad


def vanilla_breakdown_ms(q, k, v, causal=True, iters=5):
    """Per-phase time of the vanilla forward, the phases of the reference's breakdown harness
    (kernel_tests/test_flashattn_breakdown.py:44-66): qk = (q @ kT) / sqrt(d); mask = build the causal mask and add it;
    softmax; dropout = multiply by a ones "drop" matrix (what the harness does); pv = P @ v.  HIP events between the phases,
    device resident, mean of ``iters`` runs after one warm-up.  Matmuls in the inputs' dtype, the rest in fp32."""
    n, d = q.shape[-2], q.shape[-1]
    names = ("qk", "mask", "softmax", "dropout", "pv")
    acc = dict.fromkeys(names, 0.0)
    drop = torch.ones((n, n), dtype=torch.float32, device=q.device)
    for it in range(iters + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()
        s = torch.matmul(q, k.transpose(-1, -2)).float() * (d ** -0.5)
        ev[1].record()
        if causal:
            s = s + causal_mask(n, q.device)
        ev[2].record()
        p = torch.softmax(s, dim=-1)
        ev[3].record()
        p = drop * p
        ev[4].record()
        o = torch.matmul(p.to(v.dtype), v)
        ev[5].record()
        torch.cuda.synchronize()
        if it:   # the first run is the warm-up
            for i, nm in enumerate(names):
                acc[nm] += ev[i].elapsed_time(ev[i + 1]) / iters
        del s, p, o
    return acc


def time_ms(fn, iters=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
