"""Row f1: the in-model caller.  A MultiHeadAttention-shaped chain (projection -> flash attention on [B][N][H][d] ->
merge -> out projection; minitorch/modules_transfomer.py:67-157) stacked four layers deep and causal, as the reference's
DecoderLM uses it (:255-351), forward and backward, against an fp64 oracle chain built from oracle.dense_attention_fw / _bw."""
import numpy as np
import pytest

import oracle
from gpu_util import maxabs, rand_u, to_np

pytestmark = pytest.mark.gpu


def _oracle_stack(x, layers, H, causal, dout):
    """fp64 forward + backward of attention_stack.  Returns (y, dx, [(dwq, dwk, dwv, dwo)] per layer)."""
    B, N, E = x.shape
    d = E // H
    split = lambda t: t.reshape(B, N, H, d).transpose(0, 2, 1, 3)
    merge = lambda t: t.transpose(0, 2, 1, 3).reshape(B, N, E)
    saved = []
    x = x.astype(np.float64)
    for (wq, wk, wv, wo) in layers:
        wq, wk, wv, wo = (w.astype(np.float64) for w in (wq, wk, wv, wo))
        q, k, v = split(x @ wq), split(x @ wk), split(x @ wv)
        o, _, _, _ = oracle.dense_attention_fw(q, k, v, causal)
        om = merge(o)
        saved.append((x, q, k, v, o, om, (wq, wk, wv, wo)))
        x = x + om @ wo
    y = x
    dx = dout.astype(np.float64)
    grads = []
    for (xin, q, k, v, o, om, (wq, wk, wv, wo)) in reversed(saved):
        dy = dx
        dwo = np.einsum("bne,bnf->ef", om, dy)
        do = split(dy @ wo.T)
        dq, dk, dv = oracle.dense_attention_bw(q, k, v, do, causal, o=o)
        dqm, dkm, dvm = merge(dq), merge(dk), merge(dv)
        dwq = np.einsum("bne,bnf->ef", xin, dqm)
        dwk = np.einsum("bne,bnf->ef", xin, dkm)
        dwv = np.einsum("bne,bnf->ef", xin, dvm)
        dx = dy + dqm @ wq.T + dkm @ wk.T + dvm @ wv.T
        grads.append((dwq, dwk, dwv, dwo))
    return y, dx, list(reversed(grads))


@pytest.mark.parametrize("N,H,E", [(192, 4, 256), (77, 2, 64), (130, 1, 128)])
def test_four_layer_causal_attention_stack_fp32_matches_fp64_oracle_chain(N, H, E):
    """fp32 (the reference's dtype) end to end: output and every input / weight gradient of a 4-layer causal stack.  The head
    split and merge never exist as copies: the kernels read and write [B][N][H][d] in place (d = 64, 32, 128; N ragged)."""
    import torch
    from flash_attention_minitorch_amd import modules_transformer as mt
    rng = np.random.default_rng(4100 + N)
    B, LAYERS = 2, 4
    x = rand_u(rng, (B, N, E))
    layers = [tuple((rand_u(rng, (E, E)) * np.float32(1.5 / np.sqrt(E))).astype(np.float32) for _ in range(4)) for _ in range(LAYERS)]
    dout = rand_u(rng, (B, N, E))
    tx = torch.from_numpy(x).cuda().requires_grad_(True)
    tl = [tuple(torch.from_numpy(w).cuda().requires_grad_(True) for w in lw) for lw in layers]
    y = mt.attention_stack(tx, tl, H, causal=True)
    y.backward(torch.from_numpy(dout).cuda())
    ry, rdx, rg = _oracle_stack(x, layers, H, True, dout)
    tol = lambda ref: 2e-4 * max(1.0, float(np.max(np.abs(ref))))
    assert maxabs(to_np(y), ry) < tol(ry)
    assert maxabs(to_np(tx.grad), rdx) < tol(rdx)
    for li in range(LAYERS):
        for nm, got, ref in zip(("wq", "wk", "wv", "wo"), tl[li], rg[li]):
            assert maxabs(to_np(got.grad), ref) < tol(ref), (li, nm, maxabs(to_np(got.grad), ref), tol(ref))
    # the reference's data flow (permute + contiguous copies around a [B][H][N][d] operator) gives the same values
    tx2 = torch.from_numpy(x).cuda().requires_grad_(True)
    tl2 = [tuple(torch.from_numpy(w).cuda().requires_grad_(True) for w in lw) for lw in layers]
    y2 = mt.attention_stack(tx2, tl2, H, causal=True, fused_layout=False)
    y2.backward(torch.from_numpy(dout).cuda())
    assert maxabs(to_np(y2), to_np(y)) < tol(ry) and maxabs(to_np(tx2.grad), to_np(tx.grad)) < tol(rdx)
    # log2(e)/sqrt(d) folded into the query projection's weights, the operator called with softmax_scale = ln 2
    # (fa_mi355x_*_scaled): the same function of x and of every weight
    tx3 = torch.from_numpy(x).cuda().requires_grad_(True)
    tl3 = [tuple(torch.from_numpy(w).cuda().requires_grad_(True) for w in lw) for lw in layers]
    y3 = mt.attention_stack(tx3, tl3, H, causal=True, fold_scale=True)
    y3.backward(torch.from_numpy(dout).cuda())
    assert maxabs(to_np(y3), ry) < tol(ry) and maxabs(to_np(tx3.grad), rdx) < tol(rdx)
    for li in range(LAYERS):
        for nm, got, ref in zip(("wq", "wk", "wv", "wo"), tl3[li], rg[li]):
            assert maxabs(to_np(got.grad), ref) < tol(ref), ("fold_scale", li, nm, maxabs(to_np(got.grad), ref), tol(ref))


def test_bf16_attention_layer_in_place_layout_matches_oracle():
    """One bf16 layer (the dtype of the metric): the BNHD operator inside the chain against the oracle evaluated on the SAME
    bf16 q, k, v the projections produced (the projections themselves are the caller's GEMMs)."""
    import torch
    from flash_attention_minitorch_amd import device_ops as dev
    rng = np.random.default_rng(4200)
    B, N, H, d = 2, 320, 4, 64
    E = H * d
    x = torch.from_numpy(oracle.bf16_round(rand_u(rng, (B, N, E)))).to("cuda", torch.bfloat16)
    wq, wk, wv = (torch.from_numpy(oracle.bf16_round(rand_u(rng, (E, E)) * np.float32(1.0 / np.sqrt(E)))).to("cuda", torch.bfloat16)
                  for _ in range(3))
    q, k, v = ((x.reshape(B * N, E) @ w).view(B, N, H, d) for w in (wq, wk, wv))       # [B][N][H][d], no copies
    do = torch.from_numpy(oracle.bf16_round(rand_u(rng, (B, N, H, d)))).to("cuda", torch.bfloat16)
    o, l, _ = dev.flash_attn_fwd_bnhd(q, k, v, causal=True)
    dq, dk, dv = dev.flash_attn_bwd_bnhd(q, k, v, o, do, l, None, causal=True)
    f = lambda t: to_np(t.float()).transpose(0, 2, 1, 3)                              # oracle works on [B][H][N][d]
    ro, rL, _, _ = oracle.dense_attention_fw(f(q), f(k), f(v), True)
    rdq, rdk, rdv = oracle.dense_attention_bw(f(q), f(k), f(v), f(do), True)
    assert maxabs(f(o), ro) < 1e-3 and maxabs(to_np(l), rL) < 1e-3
    for got, ref in ((dq, rdq), (dk, rdk), (dv, rdv)):
        assert maxabs(f(got), ref) < 1e-3


@pytest.mark.parametrize("causal", [False, True])
def test_folded_softmax_scale_keeps_the_slot_kernels_exact_on_large_activations(causal):
    """The MFMA-slot kernels fold tau*log2(e) into a bf16 operand (one more rounding of q / k; tests/test_gpu_parity.py
    operand_rounding_envelope).  A caller that folds log2(e)/sqrt(d) into its query projection (q' formed in fp32, rounded to bf16
    ONCE) and passes softmax_scale = ln 2 makes that factor exactly 1: with activations x6 (scores of order 100) the default kernels
    are as accurate as the kernels with fp32 scaling: 5e-3 x the tensor's scale on O and L, 1e-2 on the gradients (the softmax is
    nearly one-hot at these scores, so the bf16 rounding of P / dS is not averaged out: tools/check_prescale.py measures 4e-3 relative
    for the fp32-scaling kernels on such inputs) -- an order of magnitude inside operand_rounding_envelope (0.1 here)."""
    import torch
    from flash_attention_minitorch_amd import device_ops as dev
    rng = np.random.default_rng(4300)
    B, N, H, d = 2, 512, 4, 64
    c = 1.4426950408889634 / np.sqrt(d)
    q0 = 6.0 * rand_u(rng, (B, N, H, d))
    qf = oracle.bf16_round((q0 * np.float32(c)).astype(np.float32))          # the folded projection output, rounded once
    kf, vf = oracle.bf16_round(6.0 * rand_u(rng, (B, N, H, d))), oracle.bf16_round(rand_u(rng, (B, N, H, d)))
    dof = oracle.bf16_round(rand_u(rng, (B, N, H, d)))
    t = [torch.from_numpy(a).to("cuda", torch.bfloat16) for a in (qf, kf, vf, dof)]
    ln2 = 0.6931471805599453
    o, l, _ = dev.flash_attn_fwd_bnhd(*t[:3], causal, softmax_scale=ln2)
    dq, dk, dv = dev.flash_attn_bwd_bnhd(*t[:3], o, t[3], l, None, causal, softmax_scale=ln2)
    f = lambda a: np.asarray(a, np.float64).transpose(0, 2, 1, 3)
    g = ln2 * np.sqrt(d)                                                      # the oracle applies sqrt(1/d) itself
    ro, rL, _, _ = oracle.dense_attention_fw(f(qf) * g, f(kf), f(vf), causal)
    rdq, rdk, rdv = oracle.dense_attention_bw(f(qf) * g, f(kf), f(vf), f(dof), causal)
    rdq = rdq * g                                                             # gradient with respect to the q that was passed in
    for nm, got, ref in (("o", o, ro), ("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv)):
        got = f(to_np(got))
        assert np.all(np.isfinite(got)), nm
        scale = max(1.0, float(np.max(np.abs(ref))))
        assert maxabs(got, ref) < (5e-3 if nm == "o" else 1e-2) * scale, (nm, maxabs(got, ref), scale)
    assert maxabs(to_np(l), rL) < 1e-3 * max(1.0, float(np.max(np.abs(rL))))
