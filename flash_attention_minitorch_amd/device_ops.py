"""Device-resident entry points: torch-ROCm tensors in, torch-ROCm tensors out, no host round trip.

This is SURVEY.md row f2 (replace the reference's per-call malloc / H2D / D2H,
``src/flash_attn_fw.cu:314-357``, with persistent device tensors) and what ``bench.py`` times.
torch is plumbing only: device memory, streams.  All arithmetic runs in the HIP kernels.

Tensors are (B, H, N, d) or (BH, N, d), contiguous, float32 or bfloat16; outputs (O, dQ, dK, dV) are
float32 (a bf16 store alone would exceed the 1e-3 max-abs bound).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.FA_DTYPE_F32, torch.bfloat16: _lib.FA_DTYPE_BF16}


def _check_inputs(*ts):
    t0 = ts[0]
    if not t0.is_cuda:
        raise _lib.FlashAttnLibraryError("device_ops needs GPU tensors; there is no CPU fallback")
    if t0.dtype not in _DTYPES:
        raise TypeError(f"unsupported dtype {t0.dtype}: use float32 or bfloat16")
    for t in ts:
        if t.shape != t0.shape or t.dtype != t0.dtype or t.device != t0.device:
            raise ValueError("q, k, v (and out_grad) must share shape, dtype and device")
        if not t.is_contiguous():
            raise ValueError("tensors must be contiguous [.., N, d]")
    if t0.dim() not in (3, 4):
        raise ValueError("expected (B, H, N, d) or (BH, N, d)")
    n, d = t0.shape[-2], t0.shape[-1]
    bh = t0.numel() // (n * d)
    return bh, n, d


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# per-call kernel options of fa_mi355x_fwd_ex / _bwd_ex / *_guarded (include/flash_attn_mi355x.h); all give the same results
OPTS_PHASED = (4, 2, 2)          # the round-1 phased kernels instead of the MFMA-slot ones
# Where the softmax scale is applied (option 8; include/flash_attn_mi355x.h "Softmax scale and the scale guard").  The MFMA-slot
# kernels (bf16, d = 64 / 128) fold tau*log2(e) into one bf16 operand (one more 2^-9 relative rounding of q or k, worth 8-10 % of the
# step); every other kernel scales each score in fp32, as the reference does.  The default of this module is the GUARDED call: one
# device-side pass over q and k (scale_guard, no host synchronisation), then every kernel and its fp32-scaling twin are launched and
# the one on the wrong side of the budget returns at once.
OPTS_EXACT_SCALE = (0, 0, 0, 0, 0, 0, 0, 0, 2)    # fp32 scaling whatever the operands look like (no guard pass)
OPTS_FOLDED_SCALE = (0, 0, 0, 0, 0, 0, 0, 0, 1)   # the caller vouches for U(-1, 1)-sized operands: folded scale, no guard pass


def pick_opts(q, k, budget=1e-2):
    """The guard's decision on the HOST (two reductions and ONE synchronisation): OPTS_FOLDED_SCALE when one more 2^-9 rounding of q / k
    is estimated to move a score by less than ``budget`` (log2 units; 2^-9 / sqrt(3) * tau*log2(e) * max_row |q| * max_row |k|: U(-1, 1)
    gives about 6e-3 at d = 64 and 7e-3 at d = 128), else OPTS_EXACT_SCALE.  For callers that decide once per tensor family (at model
    set-up, or every few hundred steps) and then skip the per-call guard pass; the default calls need none of this."""
    if q.dtype != torch.bfloat16 or q.shape[-1] not in (64, 128):
        return None   # fp32 and d = 32 run kernels with fp32 scaling anyway
    d = q.shape[-1]
    qn = q.float().norm(dim=-1).amax()
    kn = k.float().norm(dim=-1).amax()
    est = float(qn * kn) * (1.4426950408889634 / d ** 0.5) * (2.0 ** -9) / 3.0 ** 0.5
    return OPTS_EXACT_SCALE if est > budget else OPTS_FOLDED_SCALE


def _scale_mode(opts):
    return int(opts[8]) if opts is not None and len(opts) > 8 else 0


def scale_guard(q, k, out=None):
    """The device-side evidence the folded-scale kernels run on (fa_mi355x_scale_guard): the largest squared row norms of q and k as
    partial maxima, fa_mi355x_guard_bytes() bytes.  One pass over both tensors, asynchronous, no host synchronisation.  The forward
    and the backward of one (q, k) pair take the same guard.  Any layout: a row is the last dimension."""
    if q.dtype != k.dtype or q.shape[-1] != k.shape[-1] or not q.is_cuda or not q.is_contiguous() or not k.is_contiguous():
        raise ValueError("q and k must be contiguous GPU tensors of one dtype and row length")
    if out is None:
        out = torch.empty(_lib.core().fa_mi355x_guard_bytes() // 4, dtype=torch.float32, device=q.device)
    d = q.shape[-1]
    if q.numel() != k.numel():
        raise ValueError("q and k must have the same number of rows")
    _lib.check(_lib.core().fa_mi355x_scale_guard(_ptr(q), _ptr(k), q.numel() // d, d, _DTYPES[q.dtype], _ptr(out), _stream_ptr()))
    return out


def _wants_guard(q, opts):
    """Could a folded-scale kernel run for these operands (bf16, d = 64 / 128, option 8 left at 0)?"""
    if _lib.DIAG:
        return False   # (tools/ on the diagnostic library: its process-wide default is the folded scale, option 8 = 1)
    return q.dtype == torch.bfloat16 and q.shape[-1] in (64, 128) and _scale_mode(opts) == 0


def new_guard(q, opts=None):
    """An empty guard for a forward call to FILL (flash_attn_fwd(..., guard=g, produce_guard=True)) and the backward of the same
    (q, k) to read; None where no kernel of the call could fold the scale."""
    if not _wants_guard(q, opts):
        return None
    return torch.empty(_lib.core().fa_mi355x_guard_bytes() // 4, dtype=torch.float32, device=q.device)


def _auto_guard(q, k, opts, guard):
    """guard = "auto" on a call that only READS a guard (the backward on its own): the separate pass over q and k."""
    if not isinstance(guard, str):
        return guard
    return scale_guard(q, k) if _wants_guard(q, opts) else None


OPTS_ONE_PASS_BWD = (0, 0, 0, 0, 2)   # DIAGNOSTIC LIBRARY ONLY (tools/check_fused.py): dQ inside the key-stationary kernel, ordered hand-off
# DIAGNOSTIC LIBRARY ONLY (tools/check_chain.py): the chained one-pass backward (bf16, d = 64, non-causal, N % 256 == 0): five products
# instead of seven, the running dQ tiles carried through memory along a workgroup's key blocks, fp32 atomics only from each chain's last
# block (none when B*H >= CUs).  Measured 2-5 % slower than the two-kernel default from N = 2048 up (profiles/r04_chain_backward.txt).
OPTS_CHAINED_BWD = (0, 0, 0, 0, 3)


_NATIVE_D = (32, 64, 128)


def _padded_d(d):
    if d > 128:
        raise ValueError("head dimension d > 128 is not supported (the reference kernels assert d <= 128, src/flash_attn_fw.cu:43)")
    return 32 if d <= 32 else (64 if d <= 64 else 128)


def _pad_cols(t, dp):
    """(.., N, d) -> contiguous (.., N, dp) with zero columns d .. dp-1 (what fa_mi355x_*_padded expects)."""
    return torch.nn.functional.pad(t, (0, dp - t.shape[-1]))


def _with_opt(opts, index, value):
    o = list(opts or ()) + [0] * (index + 1)
    o[index] = value
    return tuple(o[:max(index + 1, len(opts or ()))])


def flash_attn_fwd(q, k, v, causal=False, variant=_lib.FA_VARIANT_FA2, out=None, l=None, m=None, opts=None, guard="auto",
                   out_dtype=torch.float32, produce_guard=False):
    """Forward.  Returns (out fp32, l, m): FA-1 -> l = sum exp(s - rowmax), m = rowmax;
    FA-2 -> l = logsumexp, m = None.  ``opts``: per-call kernel options (see OPTS_*).  ``guard``: "auto" = the call guards itself (a
    forward with the folded scale forms the row norms of q and k inside its own launch and its fp32-scaling twin redoes the call if
    they are beyond the budget); a tensor with ``produce_guard`` = the same, and the tensor (new_guard) is FILLED for the backward of
    this (q, k); a tensor without = a guard computed before (scale_guard); None = none (fp32 scaling).
    ``out_dtype`` = torch.bfloat16: the kernels store O as bf16 (one rounding of the fp32 result; option 9; native d only) -- for
    consumers that take a bf16 activation (sharded.py's gather at half the bytes); the backward needs the fp32 O.
    Any head dim d <= 128: d outside {32, 64, 128} is zero-padded to the next of them on the device (tau keeps the caller's d; the
    reference operator takes any d up to its assert, minitorch/cuda_kernel_ops.py:527-581 / src/flash_attn_fw.cu:43)."""
    bh, n, d = _check_inputs(q, k, v)
    lead = q.shape[:-2]
    if d not in _NATIVE_D:
        if out_dtype != torch.float32 or opts is not None:
            raise ValueError("per-call options and a bf16 output need a native head dim (32, 64, 128): other d run zero-padded through "
                             "fa_mi355x_fwd_padded, which takes neither")
        dp = _padded_d(d)
        qp, kp, vp = (_pad_cols(t, dp) for t in (q, k, v))
        outp = torch.empty(lead + (n, dp), dtype=torch.float32, device=q.device)
        if l is None:
            l = torch.empty(lead + (n,), dtype=torch.float32, device=q.device)
        if variant == _lib.FA_VARIANT_FA1 and m is None:
            m = torch.empty(lead + (n,), dtype=torch.float32, device=q.device)
        _lib.check(_lib.core().fa_mi355x_fwd_padded(_ptr(qp), _ptr(kp), _ptr(vp), _ptr(outp), _ptr(l), _ptr(m), bh, n, d, dp,
                                                    int(bool(causal)), variant, _DTYPES[q.dtype], _stream_ptr()))
        if out is None:
            out = outp[..., :d].contiguous()
        else:
            out.copy_(outp[..., :d])
        return out, l, m
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("out_dtype must be float32 or bfloat16")
    if out is None:
        out = torch.empty(q.shape, dtype=out_dtype, device=q.device)
    elif out.dtype != out_dtype or out.shape != q.shape or not out.is_contiguous():
        raise ValueError("out must be a contiguous tensor of q's shape and of out_dtype")
    if out_dtype == torch.bfloat16:
        opts = _with_opt(opts, 9, 1)
    if l is None:
        l = torch.empty(lead + (n,), dtype=torch.float32, device=q.device)
    if variant == _lib.FA_VARIANT_FA1 and m is None:
        m = torch.empty(lead + (n,), dtype=torch.float32, device=q.device)
    arr, cnt = _lib.opts_array(opts)
    if isinstance(guard, str):
        guard, produce_guard = new_guard(q, opts), True
    _lib.check(_lib.core().fa_mi355x_fwd_guarded(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(l), _ptr(m), bh, 1, n, d,
                                                 _lib.FA_LAYOUT_BHND, 0.0, int(bool(causal)), variant, _DTYPES[q.dtype], arr, cnt,
                                                 _ptr(guard), int(bool(produce_guard and guard is not None)), _stream_ptr()))
    return out, l, m


def _workspace_bytes(bh, n, d, opts=None):
    arr, cnt = _lib.opts_array(opts)
    return _lib.core().fa_mi355x_bwd_workspace_bytes_ex(bh, n, d, arr, cnt)


def _workspace(bh, n, d, device, opts=None):
    return torch.empty((_workspace_bytes(bh, n, d, opts) + 3) // 4, dtype=torch.float32, device=device)


def bwd_workspace(q, opts=None):
    """Scratch for the backward of (.., N, d) tensors, sized by the library (fa_mi355x_bwd_workspace_bytes_ex: the chained
    one-pass backward, OPTS_CHAINED_BWD, needs slabs for its running dQ tiles on top of the row constants)."""
    n, d = q.shape[-2], q.shape[-1]
    return _workspace(q.numel() // (n * d), n, _padded_d(d), q.device, opts)


def bwd_status(workspace, q):
    """Synchronous check of the one-pass backward's error word after a backward call that used ``workspace``; raises if a
    hand-off wait timed out (fa_mi355x_bwd_status)."""
    n, d = q.shape[-2], q.shape[-1]
    st = ctypes.c_int(0)
    _lib.check(_lib.core().fa_mi355x_bwd_status(_ptr(workspace), q.numel() // (n * d), n, d, ctypes.byref(st)))
    return st.value


STAGE_PREP, STAGE_DKDV, STAGE_DQ, STAGE_ALL = 1, 2, 4, 7


def flash_attn_bwd(q, k, v, out, out_grad, l, m=None, causal=False, variant=_lib.FA_VARIANT_FA2,
                   workspace=None, grads=None, stages=STAGE_ALL, opts=None, guard="auto"):
    """Backward.  out: the forward's fp32 output.  Returns (dq, dk, dv) fp32.
    ``stages`` restricts the call to some of its kernels (profiling only); ``opts``: per-call kernel options (see OPTS_*);
    ``guard`` as flash_attn_fwd (pass the forward's guard tensor to save the second pass over q and k)."""
    bh, n, d = _check_inputs(q, k, v, out_grad)
    if out.dtype != torch.float32 or out.shape != q.shape or not out.is_contiguous():
        raise ValueError("out must be the forward's contiguous float32 output")
    if d not in _NATIVE_D:   # any d <= 128: zero-padded columns, see flash_attn_fwd
        if opts is not None or stages != STAGE_ALL or workspace is not None:
            raise ValueError("per-call options, a stage mask and a caller's workspace need a native head dim (32, 64, 128): other d run "
                             "zero-padded through fa_mi355x_bwd_padded, which takes none of them")
        dp = _padded_d(d)
        qp, kp, vp, op, dop = (_pad_cols(t, dp) for t in (q, k, v, out, out_grad))
        ws = _workspace(bh, n, dp, q.device)
        gp = tuple(torch.empty(qp.shape, dtype=torch.float32, device=q.device) for _ in range(3))
        _lib.check(_lib.core().fa_mi355x_bwd_padded(_ptr(qp), _ptr(kp), _ptr(vp), _ptr(op), _ptr(dop), _ptr(gp[0]), _ptr(gp[1]),
                                                    _ptr(gp[2]), _ptr(l), _ptr(m), _ptr(ws), bh, n, d, dp, int(bool(causal)),
                                                    variant, _DTYPES[q.dtype], _stream_ptr()))
        if grads is None:
            return tuple(g[..., :d].contiguous() for g in gp)
        for dst, g in zip(grads, gp):
            dst.copy_(g[..., :d])
        return tuple(grads)
    if workspace is None:
        workspace = bwd_workspace(q, opts)
    elif workspace.numel() * workspace.element_size() < _workspace_bytes(bh, n, d, opts):
        raise ValueError("workspace too small for these options: size it with bwd_workspace(q, opts)")
    if grads is None:
        grads = tuple(torch.empty(q.shape, dtype=torch.float32, device=q.device) for _ in range(3))
    dq, dk, dv = grads
    arr, cnt = _lib.opts_array(opts)
    guard = _auto_guard(q, k, opts, guard)
    _lib.check(_lib.core().fa_mi355x_bwd_guarded(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(out_grad), _ptr(dq),
                                                 _ptr(dk), _ptr(dv), _ptr(l), _ptr(m), _ptr(workspace), bh, 1, n, d,
                                                 _lib.FA_LAYOUT_BHND, 0.0, int(bool(causal)), variant, _DTYPES[q.dtype], int(stages),
                                                 arr, cnt, _ptr(guard), _stream_ptr()))
    return dq, dk, dv


def flash_attn_fwd_bnhd(q, k, v, causal=False, variant=_lib.FA_VARIANT_FA2, softmax_scale=None, guard="auto", opts=None,
                        produce_guard=False):
    """Forward on (B, N, H, d) tensors -- the layout minitorch's projection writes before its
    permute(0,2,1,3).contiguous() (minitorch/modules_transfomer.py:67-89): no head-split copies.
    Returns (out (B, N, H, d) fp32, l (B, H, N), m (B, H, N) or None).
    ``softmax_scale``: P = softmax(softmax_scale * q.k) instead of the reference's sqrt(1/d) (fa_mi355x_fwd_scaled): for callers that
    fold the scale into their query projection (modules_transformer.multi_head_attention(fold_scale=True))."""
    if q.dim() != 4:
        raise ValueError("expected (B, N, H, d)")
    for t in (q, k, v):
        if not t.is_cuda or t.shape != q.shape or t.dtype != q.dtype or not t.is_contiguous():
            raise ValueError("q, k, v must be contiguous GPU tensors of one shape and dtype")
    B, N, H, d = q.shape
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    l = torch.empty((B, H, N), dtype=torch.float32, device=q.device)
    m = torch.empty((B, H, N), dtype=torch.float32, device=q.device) if variant == _lib.FA_VARIANT_FA1 else None
    arr, cnt = _lib.opts_array(opts)
    if isinstance(guard, str):   # (with softmax_scale = ln 2 the library ignores it: the folded factor is 1)
        guard, produce_guard = new_guard(q, opts), True
    _lib.check(_lib.core().fa_mi355x_fwd_guarded(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(l), _ptr(m), B, H, N, d,
                                                 _lib.FA_LAYOUT_BNHD, float(softmax_scale or 0.0), int(bool(causal)), variant,
                                                 _DTYPES[q.dtype], arr, cnt, _ptr(guard), int(bool(produce_guard and guard is not None)),
                                                 _stream_ptr()))
    return out, l, m


def flash_attn_bwd_bnhd(q, k, v, out, out_grad, l, m=None, causal=False, variant=_lib.FA_VARIANT_FA2, softmax_scale=None,
                        guard="auto", opts=None):
    """Backward on (B, N, H, d) tensors; returns (dq, dk, dv) in the same layout, fp32 (``softmax_scale`` as the forward's)."""
    B, N, H, d = q.shape
    ws = _workspace(B * H, N, d, q.device)
    dq, dk, dv = (torch.empty(q.shape, dtype=torch.float32, device=q.device) for _ in range(3))
    arr, cnt = _lib.opts_array(opts)
    guard = _auto_guard(q, k, opts, guard)
    _lib.check(_lib.core().fa_mi355x_bwd_guarded(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(out_grad), _ptr(dq), _ptr(dk),
                                                 _ptr(dv), _ptr(l), _ptr(m), _ptr(ws), B, H, N, d, _lib.FA_LAYOUT_BNHD,
                                                 float(softmax_scale or 0.0), int(bool(causal)), variant, _DTYPES[q.dtype],
                                                 STAGE_ALL, arr, cnt, _ptr(guard), _stream_ptr()))
    return dq, dk, dv


def _check_mask(key_mask, q):
    if q.dim() != 4:
        raise ValueError("a key mask needs (B, H, N, d) tensors: it is shared by the heads of a batch element")
    B, H, N, d = q.shape
    if (not key_mask.is_cuda or key_mask.dtype != torch.float32 or tuple(key_mask.shape) != (B, N)
            or not key_mask.is_contiguous()):
        raise ValueError("key_mask must be a contiguous float32 GPU tensor of shape (B, N)")
    return B, H, N, d


def flash_attn_fwd_masked(q, k, v, key_mask, causal=False, variant=_lib.FA_VARIANT_FA2):
    """Forward with an additive key mask: P = softmax_k(tau * q.k + key_mask[b, k]), the
    [batch, to_len] mask of the reference's fused softmax (src/softmax_kernel.cu:27-34; 0 keeps a key, -inf drops it).
    q, k, v: (B, H, N, d); key_mask: (B, N) float32.  Returns (out, l, m) as flash_attn_fwd."""
    _check_inputs(q, k, v)
    B, H, N, d = _check_mask(key_mask, q)
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    l = torch.empty((B, H, N), dtype=torch.float32, device=q.device)
    m = torch.empty((B, H, N), dtype=torch.float32, device=q.device) if variant == _lib.FA_VARIANT_FA1 else None
    _lib.check(_lib.core().fa_mi355x_fwd_masked(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(l), _ptr(m), _ptr(key_mask),
                                                B, H, N, d, _lib.FA_LAYOUT_BHND, int(bool(causal)), variant,
                                                _DTYPES[q.dtype], _stream_ptr()))
    return out, l, m


def flash_attn_bwd_masked(q, k, v, out, out_grad, l, m, key_mask, causal=False, variant=_lib.FA_VARIANT_FA2):
    """Backward of flash_attn_fwd_masked; returns (dq, dk, dv) fp32 (no gradient flows into the mask)."""
    _check_inputs(q, k, v, out_grad)
    B, H, N, d = _check_mask(key_mask, q)
    ws = _workspace(B * H, N, d, q.device)
    dq, dk, dv = (torch.empty(q.shape, dtype=torch.float32, device=q.device) for _ in range(3))
    _lib.check(_lib.core().fa_mi355x_bwd_masked(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(out_grad), _ptr(dq), _ptr(dk),
                                                _ptr(dv), _ptr(l), _ptr(m), _ptr(key_mask), _ptr(ws), B, H, N, d,
                                                _lib.FA_LAYOUT_BHND, int(bool(causal)), variant, _DTYPES[q.dtype],
                                                _stream_ptr()))
    return dq, dk, dv


def flash_attn_fwd_dropout(q, k, v, rate, seed, scale=1.0, key_mask=None, causal=False, variant=_lib.FA_VARIANT_FA2):
    """Forward with dropout on the attention probabilities (and an optional key mask): out = scale * (M o P) v with the
    stateless mask of include/flash_attn_mi355x.h (kept iff rate < r, minitorch/nn.py:168-186; scale = 1 is minitorch's
    convention).  q, k, v: (B, H, N, d).  Returns (out, l, m); l / m are the statistics before dropout."""
    _check_inputs(q, k, v)
    if q.dim() != 4:
        raise ValueError("expected (B, H, N, d)")
    B, H, N, d = _check_mask(key_mask, q) if key_mask is not None else q.shape
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    l = torch.empty((B, H, N), dtype=torch.float32, device=q.device)
    m = torch.empty((B, H, N), dtype=torch.float32, device=q.device) if variant == _lib.FA_VARIANT_FA1 else None
    _lib.check(_lib.core().fa_mi355x_fwd_dropout(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(l), _ptr(m), _ptr(key_mask),
                                                 float(rate), float(scale), int(seed) & 0xFFFFFFFF, B, H, N, d,
                                                 _lib.FA_LAYOUT_BHND, int(bool(causal)), variant, _DTYPES[q.dtype],
                                                 _stream_ptr()))
    return out, l, m


def flash_attn_bwd_dropout(q, k, v, out, out_grad, l, m, rate, seed, scale=1.0, key_mask=None, causal=False,
                           variant=_lib.FA_VARIANT_FA2):
    """Backward of flash_attn_fwd_dropout (same rate, seed, scale, mask); returns (dq, dk, dv) fp32."""
    _check_inputs(q, k, v, out_grad)
    B, H, N, d = _check_mask(key_mask, q) if key_mask is not None else q.shape
    ws = _workspace(B * H, N, d, q.device)
    dq, dk, dv = (torch.empty(q.shape, dtype=torch.float32, device=q.device) for _ in range(3))
    _lib.check(_lib.core().fa_mi355x_bwd_dropout(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(out_grad), _ptr(dq), _ptr(dk),
                                                 _ptr(dv), _ptr(l), _ptr(m), _ptr(key_mask), float(rate), float(scale),
                                                 int(seed) & 0xFFFFFFFF, _ptr(ws), B, H, N, d, _lib.FA_LAYOUT_BHND,
                                                 int(bool(causal)), variant, _DTYPES[q.dtype], _stream_ptr()))
    return dq, dk, dv


class _FlashAttnFn(torch.autograd.Function):
    """Autograd contract of the reference's Flash_Attn / Flash_Attn2 / Flash_Attn_Causal
    (minitorch/tensor_functions.py:462-497): forward returns o and saves (q, k, v, o, l, m, causal);
    backward hands them to the SAME variant's backward.  Gradients are cast to the input dtype."""

    @staticmethod
    def forward(ctx, q, k, v, causal, variant):
        guard = new_guard(q) if q.shape[-1] in _NATIVE_D else None   # filled by the forward's own launch, read by the backward
        o, l, m = flash_attn_fwd(q, k, v, causal, variant, guard=guard, produce_guard=True)
        none = torch.empty(0, device=q.device)
        ctx.save_for_backward(q, k, v, o, l, m if m is not None else none, guard if guard is not None else none)
        ctx.causal, ctx.variant = causal, variant
        return o

    @staticmethod
    def backward(ctx, out_grad):
        q, k, v, o, l, m, guard = ctx.saved_tensors
        dq, dk, dv = flash_attn_bwd(q, k, v, o, out_grad.to(q.dtype).contiguous(), l,
                                    m if m.numel() else None, ctx.causal, ctx.variant, guard=guard if guard.numel() else None)
        return dq.to(q.dtype), dk.to(q.dtype), dv.to(q.dtype), None, None


def flash_attn(q, k, v, causal=False):        # Tensor.flash_attn, minitorch/tensor.py:422-423
    return _FlashAttnFn.apply(q, k, v, bool(causal), _lib.FA_VARIANT_FA1)


def flash_attn_causal(q, k, v, causal=True):  # Tensor.flash_attn_causal, minitorch/tensor.py:425-426
    return _FlashAttnFn.apply(q, k, v, bool(causal), _lib.FA_VARIANT_FA1)


def flash_attn2(q, k, v, causal=False):       # Tensor.flash_attn2, minitorch/tensor.py:428-429
    return _FlashAttnFn.apply(q, k, v, bool(causal), _lib.FA_VARIANT_FA2)
