"""The in-model caller of the attention path: minitorch's ``MultiHeadAttention`` data flow
(``minitorch/modules_transfomer.py:67-157``: project -> split heads -> flash attention -> merge heads -> out projection)
over device-resident tensors, with the head split / merge FUSED into the kernels.

The reference materialises four full-tensor copies per layer around its flash operator:
``projection(x).view(B, N, H, d).permute(0, 2, 1, 3)`` followed by ``.contiguous()`` for q, k and v (:80-88, :113-115) and
``output.permute(0, 2, 1, 3).contiguous()`` for the result (:152).  Here the projection's ``(B, N, H*d)`` output is handed to the
kernels as ``[B][N][H][d]`` (``fa_mi355x_fwd_layout / _bwd_layout``, element (b, n, h, :) at ((b*N + n)*H + h)*d) and the
attention output comes back in that layout, i.e. already merged: no permute, no copy, forward or backward.

Only the attention operator is this repository's product; the projections are the caller's GEMMs (the reference runs them on
its own matmul kernels, ``src/combine.cu:150-252``, out of scope per SURVEY.md section 2) and are plain ``torch.matmul`` here.
LayerNorm, the feed-forward block and the embedding of ``DecoderLM`` (:255-351) are out of scope for the same reason:
``attention_stack`` chains residual attention layers only, which is what exercises the operator the way the 4-layer model does
(causal, forward and backward through several layers).
"""
from __future__ import annotations

import torch

from . import _lib, device_ops


class _FlashAttnBNHD(torch.autograd.Function):
    """flash_attn2 (``q.flash_attn2(kT, v, self.causal)``, modules_transfomer.py:119-120; autograd contract
    minitorch/tensor_functions.py:462-497) on (B, N, H, d) tensors."""

    @staticmethod
    def forward(ctx, q, k, v, causal, softmax_scale=None):
        # the forward fills the scale guard inside its own launch, the backward reads it (none needed when the caller folded the
        # scale: the kernels' factor is then exactly 1)
        guard = None if softmax_scale is not None else device_ops.new_guard(q)
        o, l, _ = device_ops.flash_attn_fwd_bnhd(q, k, v, causal, _lib.FA_VARIANT_FA2, softmax_scale, guard=guard, produce_guard=True)
        none = torch.empty(0, device=q.device)
        ctx.save_for_backward(q, k, v, o, l, guard if guard is not None else none)
        ctx.causal, ctx.softmax_scale = causal, softmax_scale
        return o

    @staticmethod
    def backward(ctx, out_grad):
        q, k, v, o, l, guard = ctx.saved_tensors
        dq, dk, dv = device_ops.flash_attn_bwd_bnhd(q, k, v, o, out_grad.to(q.dtype).contiguous(), l, None, ctx.causal,
                                                    _lib.FA_VARIANT_FA2, ctx.softmax_scale, guard=guard if guard.numel() else None)
        return dq.to(q.dtype), dk.to(q.dtype), dv.to(q.dtype), None, None


class _FlashAttnBHND(torch.autograd.Function):
    """The same operator on (B, H, N, d): what the reference's module calls after its permute + contiguous copies."""

    @staticmethod
    def forward(ctx, q, k, v, causal):
        guard = device_ops.new_guard(q)
        o, l, _ = device_ops.flash_attn_fwd(q, k, v, causal, _lib.FA_VARIANT_FA2, guard=guard, produce_guard=True)
        none = torch.empty(0, device=q.device)
        ctx.save_for_backward(q, k, v, o, l, guard if guard is not None else none)
        ctx.causal = causal
        return o

    @staticmethod
    def backward(ctx, out_grad):
        q, k, v, o, l, guard = ctx.saved_tensors
        dq, dk, dv = device_ops.flash_attn_bwd(q, k, v, o, out_grad.to(q.dtype).contiguous(), l, None, ctx.causal,
                                               _lib.FA_VARIANT_FA2, guard=guard if guard.numel() else None)
        return dq.to(q.dtype), dk.to(q.dtype), dv.to(q.dtype), None


LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453


def multi_head_attention(x, wq, wk, wv, wo, n_head: int, causal: bool = True, fused_layout: bool = True, fold_scale: bool = False):
    """MultiHeadAttention.forward (modules_transfomer.py:141-157).  x: (B, N, E); wq, wk, wv, wo: (E, E) (bias-free, as the
    reference's ``Linear(..., bias=False)`` projections, :40-52).  ``fused_layout=False`` reproduces the reference's four
    permute + contiguous copies (for comparison); both give the same values.
    ``fold_scale`` (with ``fused_layout``): log2(e)/sqrt(d) is folded into the query projection's weights and the operator is called
    with softmax_scale = ln 2 -- the same function of x, but the bf16 MFMA-slot kernels' folded scale is then exactly 1: no extra
    operand rounding whatever the magnitude of the activations (DESIGN.md section 3 "Scaling")."""
    B, N, E = x.shape
    d = E // n_head
    x2 = x.reshape(B * N, E)
    if fused_layout and fold_scale:
        q = (x2 @ (wq * (LOG2E / d ** 0.5))).view(B, N, n_head, d)
        k = (x2 @ wk).view(B, N, n_head, d)
        v = (x2 @ wv).view(B, N, n_head, d)
        o = _FlashAttnBNHD.apply(q, k, v, causal, LN2)
        merged = o.reshape(B * N, E)
    elif fused_layout:
        q = (x2 @ wq).view(B, N, n_head, d)
        k = (x2 @ wk).view(B, N, n_head, d)
        v = (x2 @ wv).view(B, N, n_head, d)
        o = _FlashAttnBNHD.apply(q, k, v, causal)                     # (B, N, H, d) fp32: already merged
        merged = o.reshape(B * N, E)
    else:
        q = (x2 @ wq).view(B, N, n_head, d).permute(0, 2, 1, 3).contiguous()
        k = (x2 @ wk).view(B, N, n_head, d).permute(0, 2, 1, 3).contiguous()
        v = (x2 @ wv).view(B, N, n_head, d).permute(0, 2, 1, 3).contiguous()
        o = _FlashAttnBHND.apply(q, k, v, causal)                     # (B, H, N, d)
        merged = o.permute(0, 2, 1, 3).contiguous().view(B * N, E)
    return (merged.to(x.dtype) @ wo).view(B, N, E)


def attention_stack(x, layers, n_head: int, causal: bool = True, fused_layout: bool = True, fold_scale: bool = False):
    """x <- x + MultiHeadAttention_l(x) for every (wq, wk, wv, wo) in ``layers``: the attention data flow of the reference's
    4-layer causal DecoderLM (modules_transfomer.py:255-351) without its out-of-scope LayerNorm / FFN blocks."""
    for (wq, wk, wv, wo) in layers:
        x = x + multi_head_attention(x, wq, wk, wv, wo, n_head, causal, fused_layout, fold_scale)
    return x
