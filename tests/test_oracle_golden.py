"""Pin the CPU oracle (oracle/attention_ref.py) to the golden vectors that were produced by
the reference's own CPU model (kernel_tests/flash_attn_python.py, see tests/golden/make_golden.py).

CPU only.  fp64 golden vs fp64 oracle: 1e-10.  fp32 golden (reference run in fp32) vs fp64
oracle: 2e-5 (reference's own GPU-test tolerances are 1e-3 fw / 1e-2 bw:
kernel_tests/test_flashattn_fw.py:23, kernel_tests/test_flashattn_bw.py:19).
"""
import os

import numpy as np
import pytest

import oracle

CASES_TILED = ["c0_b1h2n128d64", "ragged_n40d32", "ragged_n327d34"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


@pytest.mark.parametrize("name", CASES_TILED)
def test_dense_forward_matches_reference_f64(golden_dir, name):
    g = _load(golden_dir, name)
    o, L, m, l = oracle.dense_attention_fw(g["q"], g["k"], g["v"], causal=False)
    assert _maxabs(o, g["o_dense_f64"]) < 1e-12
    # FA-1 side outputs: l = sum exp(s - m), m = row max (flash_attn_python.py:45-52)
    assert _maxabs(m, g["fa1_m_f64"]) < 1e-12
    assert _maxabs(l, g["fa1_l_f64"]) < 1e-9
    # FA-2 side output: logsumexp (flash_attn_python.py:94); reference keeps l_i, m_i in fp32 there
    assert _maxabs(L, g["fa2_L_f64"]) < 5e-6
    assert _maxabs(o, g["fa1_o_f64"]) < 1e-12
    assert _maxabs(o, g["fa2_o_f64"]) < 1e-6


@pytest.mark.parametrize("name", CASES_TILED)
@pytest.mark.parametrize("kind", ["rand", "ones"])
def test_dense_backward_matches_reference_autograd_f64(golden_dir, name, kind):
    g = _load(golden_dir, name)
    do = g["do_rand"] if kind == "rand" else np.ones_like(g["q"])
    dq, dk, dv = oracle.dense_attention_bw(g["q"], g["k"], g["v"], do, causal=False)
    assert _maxabs(dq, g[f"dq_dense_{kind}_f64"]) < 1e-12
    assert _maxabs(dk, g[f"dk_dense_{kind}_f64"]) < 1e-12
    assert _maxabs(dv, g[f"dv_dense_{kind}_f64"]) < 1e-12


@pytest.mark.parametrize("name", CASES_TILED)
def test_dense_backward_matches_reference_tiled_backward_f64(golden_dir, name):
    g = _load(golden_dir, name)
    dq, dk, dv = oracle.dense_attention_bw(g["q"], g["k"], g["v"], g["do_rand"], causal=False)
    for fam, tol in (("fa1", 1e-10), ("fa2", 2e-6)):   # fa2: L went through the fp32 l_i, m_i of :74-75
        assert _maxabs(dq, g[f"{fam}_dq_rand_f64"]) < tol
        assert _maxabs(dk, g[f"{fam}_dk_rand_f64"]) < tol
        assert _maxabs(dv, g[f"{fam}_dv_rand_f64"]) < tol


@pytest.mark.parametrize("name", ["c0_b1h2n128d64", "ragged_n40d32"])
def test_reference_fp32_run_is_within_tolerance_of_oracle(golden_dir, name):
    g = _load(golden_dir, name)
    o, L, m, l = oracle.dense_attention_fw(g["q"], g["k"], g["v"])
    dq, dk, dv = oracle.dense_attention_bw(g["q"], g["k"], g["v"], g["do_rand"])
    tol = 2e-5
    assert _maxabs(o, g["o_dense_f32"]) < tol
    assert _maxabs(o, g["fa1_o_f32"]) < tol
    assert _maxabs(o, g["fa2_o_f32"]) < tol
    for fam in ("fa1", "fa2"):
        assert _maxabs(dq, g[f"{fam}_dq_rand_f32"]) < tol
        assert _maxabs(dk, g[f"{fam}_dk_rand_f32"]) < tol
        assert _maxabs(dv, g[f"{fam}_dv_rand_f32"]) < tol


def test_notebook_shape_dense(golden_dir):
    # notebooks/flash_attention_backward.ipynb cell 1: N=334, d=233, rtol=atol=1e-4
    g = _load(golden_dir, "nb_n334d233")
    o, *_ = oracle.dense_attention_fw(g["q"], g["k"], g["v"])
    dq, dk, dv = oracle.dense_attention_bw(g["q"], g["k"], g["v"], g["do_rand"])
    assert _maxabs(o, g["o_dense_f64"]) < 1e-6      # stored rounded to fp32
    assert _maxabs(dq, g["dq_dense_rand_f64"]) < 1e-6
    assert _maxabs(dk, g["dk_dense_rand_f64"]) < 1e-6
    assert _maxabs(dv, g["dv_dense_rand_f64"]) < 1e-6


@pytest.mark.parametrize("name", ["c0_b1h2n128d64", "ragged_n40d32"])
def test_tiled_restatements_match_reference(golden_dir, name):
    """Our tiled FA-1 / FA-2 restatements, run with the reference's tile sizes, reproduce its outputs."""
    g = _load(golden_dir, name)
    q, k, v, do = g["q"], g["k"], g["v"], g["do_rand"]
    for h in range(q.shape[0]):
        o1, l1, m1 = oracle.fa1_forward_tiled(q[h], k[h], v[h])                 # B_c=16, B_r=min(16,d)
        assert _maxabs(o1, g["fa1_o_f64"][h]) < 1e-12
        assert _maxabs(l1, g["fa1_l_f64"][h]) < 1e-10
        assert _maxabs(m1, g["fa1_m_f64"][h]) < 1e-12
        o2, L2 = oracle.fa2_forward_tiled(q[h], k[h], v[h])                     # B_c=B_r=4
        assert _maxabs(o2, g["fa2_o_f64"][h]) < 1e-6
        assert _maxabs(L2, g["fa2_L_f64"][h]) < 5e-6
        g1 = oracle.fa1_backward_tiled(q[h], k[h], v[h], g["fa1_o_f64"][h], do[h], g["fa1_l_f64"][h], g["fa1_m_f64"][h])
        g2 = oracle.fa2_backward_tiled(q[h], k[h], v[h], g["fa2_o_f64"][h], do[h], g["fa2_L_f64"][h])
        for a, nm in zip(g1, ("dq", "dk", "dv")):
            assert _maxabs(a, g[f"fa1_{nm}_rand_f64"][h]) < 1e-10
        for a, nm in zip(g2, ("dq", "dk", "dv")):
            assert _maxabs(a, g[f"fa2_{nm}_rand_f64"][h]) < 1e-10


@pytest.mark.parametrize("causal", [False, True])
def test_tiled_equals_dense_including_causal(causal):
    """Causal semantics (src/flash_attn_fw.cu:152-159) are not in the reference's CPU model; the tiled
    and dense restatements must agree with each other, with ragged N and unequal tiles."""
    rng = np.random.default_rng(7)
    N, d = 45, 16
    q, k, v, do = (rng.uniform(-1, 1, (N, d)) for _ in range(4))
    o, L, m, l = oracle.dense_attention_fw(q, k, v, causal)
    o1, l1, m1 = oracle.fa1_forward_tiled(q, k, v, B_r=8, B_c=16, causal=causal)
    o2, L2 = oracle.fa2_forward_tiled(q, k, v, B_r=16, B_c=8, causal=causal)
    assert _maxabs(o, o1) < 1e-12 and _maxabs(o, o2) < 1e-12
    assert _maxabs(L, m1 + np.log(l1)) < 1e-12 and _maxabs(L, L2) < 1e-12
    assert _maxabs(m, m1) < 1e-12
    dq, dk, dv = oracle.dense_attention_bw(q, k, v, do, causal)
    for got in (oracle.fa1_backward_tiled(q, k, v, o, do, l, m, B_r=8, B_c=4, causal=causal),
                oracle.fa2_backward_tiled(q, k, v, o, do, L, B_r=4, B_c=8, causal=causal)):
        for a, b in zip(got, (dq, dk, dv)):
            assert _maxabs(a, b) < 1e-12


def test_dense_backward_matches_finite_differences():
    rng = np.random.default_rng(3)
    N, d = 9, 4
    q, k, v, do = (rng.uniform(-1, 1, (N, d)) for _ in range(4))
    for causal in (False, True):
        dq, dk, dv = oracle.dense_attention_bw(q, k, v, do, causal)
        f = lambda q_, k_, v_: float((oracle.dense_attention_fw(q_, k_, v_, causal)[0] * do).sum())
        eps = 1e-6
        for arr, grad, idx in ((q, dq, 0), (k, dk, 1), (v, dv, 2)):
            for (i, j) in ((0, 0), (3, 2), (8, 3)):
                args_p = [q.copy(), k.copy(), v.copy()]; args_m = [q.copy(), k.copy(), v.copy()]
                args_p[idx][i, j] += eps; args_m[idx][i, j] -= eps
                fd = (f(*args_p) - f(*args_m)) / (2 * eps)
                assert abs(fd - grad[i, j]) < 1e-7


def test_bf16_round_ties_to_even():
    x = np.array([1.0, 1.00390625, 1.005859375, 1.01171875, -2.0078125, 3.14159], dtype=np.float32)
    r = oracle.bf16_round(x)
    assert r[0] == 1.0
    assert r[1] == 1.0            # 1 + 2^-8 is a tie -> even mantissa (1.0)
    assert r[2] == 1.0078125      # above the tie -> up
    assert r[3] == 1.015625       # 1 + 3*2^-8 tie -> even (1 + 2^-6)
    assert r[4] == -2.0           # -(2 + 2^-7) tie -> even
    import torch
    t = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(r, t)
    big = np.random.default_rng(0).standard_normal(10000).astype(np.float32)
    assert np.array_equal(oracle.bf16_round(big), torch.from_numpy(big).to(torch.bfloat16).float().numpy())


# ---------------------------------------------------------------- additive key mask
def test_masked_oracle_reduces_to_pinned_unmasked_oracle():
    """The mask has no reference fixture (CUDA only, src/softmax_kernel.cu:27-34,77-90): it is tied to the pinned dense
    functions by identities -- a zero mask changes nothing; a -inf mask equals attention over the kept keys alone; a
    finite mask equals adding it to the scores (checked through the log-sum-exp shift of one key)."""
    rng = np.random.default_rng(11)
    B, H, N, d = 2, 3, 37, 16
    q, k, v, do = (rng.uniform(-1, 1, (B, H, N, d)) for _ in range(4))
    for causal in (False, True):
        o0, L0, _, _ = oracle.dense_attention_fw(q, k, v, causal)
        g0 = oracle.dense_attention_bw(q, k, v, do, causal)
        zero = np.zeros((B, 1, N))
        o1, L1 = oracle.masked_attention_fw(q, k, v, zero, causal)
        g1 = oracle.masked_attention_bw(q, k, v, do, zero, causal)
        assert np.allclose(o1, o0, atol=1e-13) and np.allclose(L1, L0, atol=1e-13)
        for a, b in zip(g1, g0):
            assert np.allclose(a, b, atol=1e-13)
    # -inf on a key subset == dense attention on the compacted K / V (non-causal: positions do not matter)
    keep = rng.uniform(size=(B, N)) < 0.6
    keep[:, 0] = True
    mask = np.where(keep, 0.0, -np.inf)[:, None, :]
    o, L = oracle.masked_attention_fw(q, k, v, mask, False)
    dq, dk, dv = oracle.masked_attention_bw(q, k, v, do, mask, False)
    for b in range(B):
        idx = np.nonzero(keep[b])[0]
        oc, Lc, _, _ = oracle.dense_attention_fw(q[b], k[b][:, idx], v[b][:, idx], False)
        assert np.allclose(o[b], oc, atol=1e-13) and np.allclose(L[b], Lc, atol=1e-13)
        # gradients w.r.t. the kept keys match the compacted problem's, dropped keys get exactly zero
        s = np.matmul(q[b], np.swapaxes(k[b][:, idx], -1, -2)) / np.sqrt(d)
        p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
        dvc = np.matmul(np.swapaxes(p, -1, -2), do[b])
        assert np.allclose(dv[b][:, idx], dvc, atol=1e-13)
        drop = np.nonzero(~keep[b])[0]
        assert np.all(dv[b][:, drop] == 0) and np.all(dk[b][:, drop] == 0)
    # fully dropped rows (causal, key 0 masked: query 0 sees nothing): O = 0, L = -inf, finite gradients
    mask2 = np.zeros((B, 1, N)); mask2[:, :, 0] = -np.inf
    o2, L2 = oracle.masked_attention_fw(q, k, v, mask2, True)
    g2 = oracle.masked_attention_bw(q, k, v, do, mask2, True)
    assert np.all(o2[:, :, 0] == 0) and np.all(np.isneginf(L2[:, :, 0])) and np.all(np.isfinite(o2))
    assert all(np.all(np.isfinite(g)) for g in g2)


def test_dropout_oracle_identities():
    """Dropout (row f4) has no reference fixture; the oracle is tied to the pinned functions: an all-ones keep matrix is
    the unmasked operator (the reference's own test multiplies by ones, kernel_tests/test_flashattn_fw.py:66,71), the
    analytic backward matches finite differences with a fixed 0/1 matrix, and the hash mask keeps ~ (1 - rate)."""
    rng = np.random.default_rng(12)
    B, H, N, d = 1, 2, 24, 8
    q, k, v, do = (rng.uniform(-1, 1, (B, H, N, d)) for _ in range(4))
    ones = np.ones((B * H, N, N), dtype=bool)
    for causal in (False, True):
        o0, L0, _, _ = oracle.dense_attention_fw(q, k, v, causal)
        o1, L1 = oracle.dropout_attention_fw(q, k, v, ones, 1.0, None, causal)
        assert np.allclose(o1, o0, atol=1e-13) and np.allclose(L1, L0, atol=1e-13)
        for a, b in zip(oracle.dropout_attention_bw(q, k, v, do, ones, 1.0, None, causal),
                        oracle.dense_attention_bw(q, k, v, do, causal)):
            assert np.allclose(a, b, atol=1e-13)
    keep = oracle.dropout_keep_mask(B * H, N, 0.3, 1234)
    scale = 1.0 / 0.7
    f = lambda q_, k_, v_: float((oracle.dropout_attention_fw(q_, k_, v_, keep, scale, None, True)[0] * do).sum())
    dq, dk, dv = oracle.dropout_attention_bw(q, k, v, do, keep, scale, None, True)
    eps = 1e-6
    for arr, grad, idx in ((q, dq, (0, 1, 5, 3)), (k, dk, (0, 0, 2, 7)), (v, dv, (0, 1, 9, 0))):
        hi, lo = arr.copy(), arr.copy()
        hi[idx] += eps
        lo[idx] -= eps
        args_hi = [hi if a is arr else a for a in (q, k, v)]
        args_lo = [lo if a is arr else a for a in (q, k, v)]
        fd = (f(*args_hi) - f(*args_lo)) / (2 * eps)
        assert abs(fd - grad[idx]) < 1e-6, (fd, grad[idx])
    big = oracle.dropout_keep_mask(4, 256, 0.25, 7)
    assert abs(big.mean() - 0.75) < 0.01
    assert oracle.dropout_keep_mask(2, 16, 0.0, 3).all()
    # decorrelated across heads and seeds
    assert abs((big[0] == big[1]).mean() - (0.75 ** 2 + 0.25 ** 2)) < 0.02
