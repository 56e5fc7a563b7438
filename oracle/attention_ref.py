"""NumPy restatement of the reference's attention math (CPU oracle; test infrastructure only).

Every function cites the reference file:line it follows (paths relative to the
reference repository root).  The reference's CPU model works per head on (N, d)
torch tensors; here everything is batched over arbitrary leading dims
``(..., N, d)`` and computed in a caller-chosen dtype (float64 by default so it
can act as the "exact" side of a tolerance check).

Two families:

* dense_*  -- the materialised-S formulation
  (``kernel_tests/flash_attn_python.py:4-14`` forward,
  ``minitorch/modules_transfomer.py:123-127`` in-model form) with the analytic
  backward the tiled reference kernels implement
  (``kernel_tests/flash_attn_python.py:127-141``).  Causal masking follows the
  kernels (``src/flash_attn_fw.cu:152-159``: key index <= query index is kept)
  and the vanilla mask ``-FLT_MAX * triu(ones, 1)``
  (``kernel_tests/test_flashattn_fw.py:18-20``).
* fa1_* / fa2_* -- the tiled FlashAttention-1 / -2 recurrences with
  configurable tile sizes, restating ``kernel_tests/flash_attn_python.py:16-192``.
  Python loops over tiles: small cases only.
"""
from __future__ import annotations

import math
import numpy as np

__all__ = [
    "bf16_round",
    "dense_attention_fw",
    "dense_attention_bw",
    "fa1_forward_tiled",
    "fa2_forward_tiled",
    "fa1_backward_tiled",
    "fa2_backward_tiled",
    "vanilla_attention_fw_bw_f32",
    "masked_attention_fw",
    "masked_attention_bw",
    "dropout_keep_mask",
    "dropout_attention_fw",
    "dropout_attention_bw",
]


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 values to the nearest bfloat16 (ties to even), returned as fp32.

    The bf16 configurations (BASELINE.json configs[3], configs[4] and the metric
    shape) feed the SAME bf16-rounded Q/K/V/dO to the GPU path and to this oracle
    (SURVEY.md section 8d "Synthetic inputs").
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounding = np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))
    r = ((u + rounding) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return r.view(np.float32).reshape(x.shape)


def _tau(d: int) -> float:
    # tau = sqrt(1/d): kernel_tests/flash_attn_python.py:10, src/flash_attn_fw.cu:37
    return math.sqrt(1.0 / d)


def _scores(q, k, causal: bool, dtype):
    """tau * Q K^T with the strict upper triangle excluded when causal.

    kernel_tests/flash_attn_python.py:11-12 (scores * tau);
    src/flash_attn_fw.cu:152-159 (keep iff key <= query).
    """
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    n, d = q.shape[-2], q.shape[-1]
    s = np.matmul(q, np.swapaxes(k, -1, -2)) * dtype(_tau(d))
    if causal:
        keep = np.tril(np.ones((n, k.shape[-2]), dtype=bool))
        s = np.where(keep, s, -np.inf)
    return s


def dense_attention_fw(q, k, v, causal: bool = False, dtype=np.float64):
    """softmax(tau Q K^T [+ causal mask]) V.

    Follows ``compute_attention`` / ``softmax`` in
    kernel_tests/flash_attn_python.py:4-14 (max-subtracted softmax).

    Returns ``(O, L, m, l)``: ``m`` = row max of the (masked) scaled scores,
    ``l`` = sum exp(s - m), ``L`` = m + log l -- the side outputs of the FA-1
    kernel (src/flash_attn_fw.cu:228-229,259-276) and of the FA-2 kernel
    (src/flash_attn2_fw.cu:279-294) respectively.
    """
    dtype = np.dtype(dtype).type
    s = _scores(q, k, causal, dtype)
    m = s.max(axis=-1)
    p = np.exp(s - m[..., None])
    l = p.sum(axis=-1)
    o = np.matmul(p, np.asarray(v, dtype=dtype)) / l[..., None]
    L = m + np.log(l)
    return o, L, m, l


def dense_attention_bw(q, k, v, do, causal: bool = False, dtype=np.float64, o=None):
    """Analytic backward of dense attention.

    dV = P^T dO; dP = dO V^T; D = rowsum(dO * O); dS = P * (dP - D);
    dQ = tau dS K; dK = tau dS^T Q  -- the per-tile algebra of
    kernel_tests/flash_attn_python.py:130-141 / :177-189 summed over all tiles.
    """
    dtype = np.dtype(dtype).type
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    do = np.asarray(do, dtype=dtype)
    d = q.shape[-1]
    tau = dtype(_tau(d))
    s = _scores(q, k, causal, dtype)
    m = s.max(axis=-1, keepdims=True)
    p = np.exp(s - m)
    p /= p.sum(axis=-1, keepdims=True)
    if o is None:
        o = np.matmul(p, v)
    else:
        o = np.asarray(o, dtype=dtype)
    dv = np.matmul(np.swapaxes(p, -1, -2), do)
    dp = np.matmul(do, np.swapaxes(v, -1, -2))
    delta = (do * o).sum(axis=-1, keepdims=True)
    ds = p * (dp - delta)
    dq = tau * np.matmul(ds, k)
    dk = tau * np.matmul(np.swapaxes(ds, -1, -2), q)
    return dq, dk, dv


def _masked_scores(q, k, key_mask, causal, dtype):
    """tau * Q K^T + key_mask[..., None, :] with the causal rule applied first, exactly the order of the reference's
    fused softmax (src/softmax_kernel.cu:77-90: `mask_future` entries become -inf, every other entry gets
    `+ attn_mask[to]`); attn_mask is [batch, to_len], 0 for tokens and -inf for padding (:27-34), broadcast over heads
    and queries.  Parity unpinned for the mask itself: the reference implements it only in CUDA (no CPU model, no
    fixture); tests anchor it on two identities the pinned unmasked oracle provides -- a zero mask changes nothing, and
    a -inf mask equals attention over the kept keys alone."""
    s = _scores(q, k, causal, dtype)
    km = np.asarray(key_mask, dtype=dtype)
    return s + km[..., None, :]


def masked_attention_fw(q, k, v, key_mask, causal: bool = False, dtype=np.float64):
    """softmax(tau Q K^T + key_mask) V for (..., N, d) arrays and a (..., N) additive key mask broadcast over the
    leading dims it lacks (give it shape (B, 1, N) for (B, H, N, d) inputs).  Returns (O, L).  A row whose every key
    is dropped returns O = 0 and L = -inf (the HIP path's documented convention, include/flash_attn_mi355x.h)."""
    dtype = np.dtype(dtype).type
    s = _masked_scores(q, k, key_mask, causal, dtype)
    m = s.max(axis=-1, keepdims=True)
    dead = ~np.isfinite(m)
    with np.errstate(invalid="ignore", divide="ignore"):
        p = np.exp(s - np.where(dead, 0.0, m))
        l = p.sum(axis=-1, keepdims=True)
        o = np.where(dead, 0.0, np.matmul(p, np.asarray(v, dtype=dtype)) / np.where(dead, 1.0, l))
        L = np.where(dead[..., 0], -np.inf, m[..., 0] + np.log(np.where(dead, 1.0, l))[..., 0])
    return o, L


def masked_attention_bw(q, k, v, do, key_mask, causal: bool = False, dtype=np.float64):
    """Analytic backward of masked_attention_fw (the algebra of dense_attention_bw; dropped keys have P = 0, fully
    dropped rows contribute nothing).  Returns (dQ, dK, dV)."""
    dtype = np.dtype(dtype).type
    q, k, v, do = (np.asarray(x, dtype=dtype) for x in (q, k, v, do))
    tau = dtype(_tau(q.shape[-1]))
    s = _masked_scores(q, k, key_mask, causal, dtype)
    m = s.max(axis=-1, keepdims=True)
    dead = ~np.isfinite(m)
    with np.errstate(invalid="ignore", divide="ignore"):
        p = np.exp(s - np.where(dead, 0.0, m))
        p = np.where(dead, 0.0, p / np.where(dead, 1.0, p.sum(axis=-1, keepdims=True)))
    o = np.matmul(p, v)
    dv = np.matmul(np.swapaxes(p, -1, -2), do)
    dp = np.matmul(do, np.swapaxes(v, -1, -2))
    ds = p * (dp - (do * o).sum(axis=-1, keepdims=True))
    return tau * np.matmul(ds, k), tau * np.matmul(np.swapaxes(ds, -1, -2), q), dv


def dropout_keep_mask(BH: int, N: int, rate: float, seed: int) -> np.ndarray:
    """The HIP path's stateless dropout mask (csrc/fa_atoms.h drop_base / drop_keep) restated in NumPy uint32
    arithmetic: keep[bh, q, k] = (hash32(seed + bh*0xC2B2AE3D + q*0x9E3779B1 + k*0x85EBCA77) >> 8) >= floor(rate * 2^24),
    i.e. "rate < r" with r a 24-bit uniform, minitorch's keep rule (minitorch/nn.py:168-186: `drop = rate < r`).
    The reference applies no dropout on its flash path and draws its masks from NumPy's global RNG elsewhere, so
    there is nothing bit-level to pin: what IS the reference's is the rule and that the mask multiplies the
    probabilities (kernel_tests/test_flashattn_fw.py:66,71)."""
    thr = np.uint32(int(float(np.float32(rate)) * 16777216.0))
    with np.errstate(over="ignore"):
        bh = np.arange(BH, dtype=np.uint32)[:, None, None] * np.uint32(0xC2B2AE3D)
        qq = np.arange(N, dtype=np.uint32)[None, :, None] * np.uint32(0x9E3779B1)
        kk = np.arange(N, dtype=np.uint32)[None, None, :] * np.uint32(0x85EBCA77)
        a = (np.uint32(seed & 0xFFFFFFFF) + bh + qq + kk).astype(np.uint32)
        a ^= a >> np.uint32(16)
        a *= np.uint32(0x7FEB352D)
        a ^= a >> np.uint32(15)
        a *= np.uint32(0x846CA68B)
        a ^= a >> np.uint32(16)
    return (a >> np.uint32(8)) >= thr


def dropout_attention_fw(q, k, v, keep, scale=1.0, key_mask=None, causal: bool = False, dtype=np.float64):
    """out = scale * (keep o softmax(tau Q K^T + mask)) V; returns (O, L) with L the log-sum-exp BEFORE dropout.
    q, k, v: (B, H, N, d); keep: boolean (B*H, N, N) or broadcastable to (B, H, N, N)."""
    dtype = np.dtype(dtype).type
    B, H, N, d = q.shape
    km = np.zeros((B, 1, N)) if key_mask is None else key_mask
    s = _masked_scores(q, k, km, causal, dtype)
    m = s.max(axis=-1, keepdims=True)
    dead = ~np.isfinite(m)
    with np.errstate(invalid="ignore", divide="ignore"):
        p = np.exp(s - np.where(dead, 0.0, m))
        l = p.sum(axis=-1, keepdims=True)
        p = np.where(dead, 0.0, p / np.where(dead, 1.0, l))
        L = np.where(dead[..., 0], -np.inf, m[..., 0] + np.log(np.where(dead, 1.0, l))[..., 0])
    pd = p * np.asarray(keep).reshape(B, H, N, N) * dtype(scale)
    return np.matmul(pd, np.asarray(v, dtype=dtype)), L


def dropout_attention_bw(q, k, v, do, keep, scale=1.0, key_mask=None, causal: bool = False, dtype=np.float64):
    """Backward of dropout_attention_fw: dV = (scale M o P)^T dO; dS = P o (scale M o (dO V^T) - rowsum(dO o O));
    dQ = tau dS K; dK = tau dS^T Q (the tile algebra of kernel_tests/flash_attn_python.py:130-141 with the mask on P)."""
    dtype = np.dtype(dtype).type
    q, k, v, do = (np.asarray(x, dtype=dtype) for x in (q, k, v, do))
    B, H, N, d = q.shape
    tau = dtype(_tau(d))
    km = np.zeros((B, 1, N)) if key_mask is None else key_mask
    s = _masked_scores(q, k, km, causal, dtype)
    m = s.max(axis=-1, keepdims=True)
    dead = ~np.isfinite(m)
    with np.errstate(invalid="ignore", divide="ignore"):
        p = np.exp(s - np.where(dead, 0.0, m))
        p = np.where(dead, 0.0, p / np.where(dead, 1.0, p.sum(axis=-1, keepdims=True)))
    mk = np.asarray(keep).reshape(B, H, N, N) * dtype(scale)
    pd = p * mk
    o = np.matmul(pd, v)
    dv = np.matmul(np.swapaxes(pd, -1, -2), do)
    dp = mk * np.matmul(do, np.swapaxes(v, -1, -2))
    ds = p * (dp - (do * o).sum(axis=-1, keepdims=True))
    return tau * np.matmul(ds, k), tau * np.matmul(np.swapaxes(ds, -1, -2), q), dv


def vanilla_attention_fw_bw_f32(q, k, v, do, causal: bool = False):
    """fp32 "vanilla attention" forward + backward: the CPU baseline that is timed.

    The reference's comparison point is materialised-S attention
    ``softmax((q @ kT)/sqrt(d) + M) @ v`` (minitorch/modules_transfomer.py:123-127,
    kernel_tests/test_flashattn_fw.py:60-76).  Restated in NumPy fp32 with the
    analytic backward; (BH, N, d) inputs.  Returns (o, dq, dk, dv).
    """
    o, _, _, _ = dense_attention_fw(q, k, v, causal, dtype=np.float32)
    dq, dk, dv = dense_attention_bw(q, k, v, do, causal, dtype=np.float32, o=o)
    return o, dq, dk, dv


# --------------------------------------------------------------------------
# Tiled restatements (per head, (N, d) arrays).  Small shapes only.
# --------------------------------------------------------------------------

def _mask_tile(s, i0, j0, causal):
    """src/flash_attn_fw.cu:152-159: entry kept iff (j0 + c) <= (i0 + r)."""
    if not causal:
        return s
    r = np.arange(s.shape[0])[:, None] + i0
    c = np.arange(s.shape[1])[None, :] + j0
    return np.where(c <= r, s, -np.inf)


def fa1_forward_tiled(Q, K, V, B_r=None, B_c=16, causal=False, dtype=np.float64):
    """FlashAttention-1 forward, K/V outer loop, O/l/m running state.

    Restates kernel_tests/flash_attn_python.py:16-56 (and the kernel
    src/flash_attn_fw.cu:67-276).  Default tiles B_c=16, B_r=min(16, d)
    (flash_attn_python.py:26-27).  Returns (O, l, m) with
    l = sum exp(s - m), m = running row max.
    """
    dtype = np.dtype(dtype).type
    Q = np.asarray(Q, dtype=dtype); K = np.asarray(K, dtype=dtype); V = np.asarray(V, dtype=dtype)
    N, d = Q.shape
    tau = dtype(_tau(d))
    if B_r is None:
        B_r = min(B_c, d)
    O = np.zeros_like(Q)
    l = np.zeros(N, dtype=np.float64)            # flash_attn_python.py:30
    m = np.full(N, -np.inf, dtype=np.float64)    # flash_attn_python.py:31
    for j0 in range(0, N, B_c):
        Kj, Vj = K[j0:j0 + B_c], V[j0:j0 + B_c]
        for i0 in range(0, N, B_r):
            if causal and j0 > i0 + B_r - 1:
                continue                          # src/flash_attn_fw.cu:88-92 (block skip)
            sl = slice(i0, i0 + B_r)
            S = _mask_tile(tau * (Q[sl] @ Kj.T), i0, j0, causal)   # :41
            m_ij = S.max(axis=1)                                   # :42
            with np.errstate(invalid="ignore"):
                P = np.exp(S - m_ij[:, None])                      # :43
            P = np.where(np.isfinite(m_ij)[:, None], P, 0.0)
            l_ij = P.sum(axis=1)                                   # :44
            m_new = np.maximum(m[sl], m_ij)                        # :45
            with np.errstate(invalid="ignore"):
                a = np.where(np.isfinite(m[sl]), np.exp(m[sl] - m_new), 0.0)
                b = np.where(np.isfinite(m_ij), np.exp(m_ij - m_new), 0.0)
            l_new = a * l[sl] + b * l_ij                           # :47
            O[sl] = ((a * l[sl])[:, None] * O[sl] + b[:, None] * (P @ Vj)) / l_new[:, None]  # :48
            m[sl] = m_new                                          # :50
            l[sl] = l_new                                          # :52
    return O, l, m


def fa2_forward_tiled(Q, K, V, B_r=None, B_c=4, causal=False, dtype=np.float64):
    """FlashAttention-2 forward, Q outer loop, on-chip O/l/m, logsumexp output.

    Restates kernel_tests/flash_attn_python.py:59-98 (kernel:
    src/flash_attn2_fw.cu:67-294).  Default tiles B_c = B_r = 4
    (flash_attn_python.py:63-64).  Returns (O, L), L = m + log l (:94).
    """
    dtype = np.dtype(dtype).type
    Q = np.asarray(Q, dtype=dtype); K = np.asarray(K, dtype=dtype); V = np.asarray(V, dtype=dtype)
    N, d = Q.shape
    tau = dtype(_tau(d))
    if B_r is None:
        B_r = min(B_c, d)
    O = np.zeros_like(Q)
    L = np.zeros(N, dtype=np.float64)
    for i0 in range(0, N, B_r):
        Qi = Q[i0:i0 + B_r]
        Oi = np.zeros_like(Qi)
        li = np.zeros(len(Qi), dtype=dtype)
        mi = np.full(len(Qi), -np.inf, dtype=dtype)
        for j0 in range(0, N, B_c):
            if causal and j0 > i0 + B_r - 1:
                break                                     # src/flash_attn2_fw.cu:95-99
            S = _mask_tile(tau * (Qi @ K[j0:j0 + B_c].T), i0, j0, causal)   # :82
            m_prev = mi.copy()                                              # :83
            mi = np.maximum(mi, S.max(axis=1))                              # :84
            P = np.exp(S - mi[:, None])                                     # :85
            with np.errstate(invalid="ignore"):
                alpha = np.where(np.isfinite(m_prev), np.exp(m_prev - mi), 0.0)
            li = alpha * li + P.sum(axis=1)                                 # :86
            Oi = alpha[:, None] * Oi + P @ V[j0:j0 + B_c]                   # :88-91
        O[i0:i0 + B_r] = Oi / li[:, None]                                   # :93
        L[i0:i0 + B_r] = mi + np.log(li)                                    # :94
    return O, L


def _bw_common(Q, K, V, O, dO, prob_fn, B_r, B_c, causal, dtype):
    dtype = np.dtype(dtype).type
    Q = np.asarray(Q, dtype=dtype); K = np.asarray(K, dtype=dtype); V = np.asarray(V, dtype=dtype)
    O = np.asarray(O, dtype=dtype); dO = np.asarray(dO, dtype=dtype)
    N, d = Q.shape
    tau = dtype(_tau(d))
    dQ = np.zeros_like(Q); dK = np.zeros_like(K); dV = np.zeros_like(V)
    for j0 in range(0, N, B_c):
        Kj, Vj = K[j0:j0 + B_c], V[j0:j0 + B_c]
        dKj = np.zeros_like(Kj); dVj = np.zeros_like(Vj)
        for i0 in range(0, N, B_r):
            if causal and j0 > i0 + B_r - 1:
                continue                                  # src/flash_attn_bw.cu:94-98
            sl = slice(i0, i0 + B_r)
            S = _mask_tile(tau * (Q[sl] @ Kj.T), i0, j0, causal)
            P = prob_fn(S, sl)
            dVj = dVj + P.T @ dO[sl]                      # flash_attn_python.py:133 / :180
            dP = dO[sl] @ Vj.T                            # :134 / :181
            D = (dO[sl] * O[sl]).sum(axis=1)              # :135 / :183 (per tile)
            dS = P * (dP - D[:, None])                    # :138 / :184
            dQ[sl] = dQ[sl] + tau * dS @ Kj               # :139 / :186
            dKj = dKj + tau * dS.T @ Q[sl]                # :140 / :187
        dK[j0:j0 + B_c] = dKj
        dV[j0:j0 + B_c] = dVj
    return dQ, dK, dV


def fa1_backward_tiled(Q, K, V, O, dO, l, m, B_r=None, B_c=4, causal=False, dtype=np.float64):
    """FlashAttention-1 backward: P = (1/l) exp(S - m).

    Restates kernel_tests/flash_attn_python.py:100-145 (kernel
    src/flash_attn_bw.cu:94-257); tiles B_c = B_r = 4 (:107-108).
    """
    dt = np.dtype(dtype).type
    l = np.asarray(l, dtype=dt); m = np.asarray(m, dtype=dt)
    if B_r is None:
        B_r = min(B_c, np.asarray(Q).shape[1])

    def prob(S, sl):
        return (1.0 / l[sl])[:, None] * np.exp(S - m[sl][:, None])   # :131

    return _bw_common(Q, K, V, O, dO, prob, B_r, B_c, causal, dtype)


def fa2_backward_tiled(Q, K, V, O, dO, L, B_r=None, B_c=4, causal=False, dtype=np.float64):
    """FlashAttention-2 backward: P = exp(S - L).

    Restates kernel_tests/flash_attn_python.py:147-192 (kernel
    src/flash_attn2_bw.cu:94-259).
    """
    dt = np.dtype(dtype).type
    L = np.asarray(L, dtype=dt)
    if B_r is None:
        B_r = min(B_c, np.asarray(Q).shape[1])

    def prob(S, sl):
        return np.exp(S - L[sl][:, None])                             # :178

    return _bw_common(Q, K, V, O, dO, prob, B_r, B_c, causal, dtype)
