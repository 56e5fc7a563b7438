"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the committed golden
vectors.  Tolerances (max-abs, stated per SURVEY.md section 8d):
  fp32 path : 1e-4 on O, dQ, dK, dV, L; FA-1 m equals the row max within 1e-5
              (the reference's own GPU tests use 1e-3 fw / 1e-2 bw: kernel_tests/test_flashattn_fw.py:23, _bw.py:19)
  bf16 path : 1e-3 on fp32-stored O, dQ, dK, dV, L vs the fp64 oracle on the SAME bf16-rounded inputs (north_star), causal and
              not.  P and dS enter the second MFMA of each product as bf16 (relative quantisation 2^-9 = 1.95e-3); rows with
              many keys average that out.  Rows with fewer than 64 admissible keys (the first rows under the causal mask,
              N < 64, rows thinned by a key mask or dropout) do not: there the kernels issue the product twice, with P / dS
              split into two bf16 fragments (Atom::pack_lo; round 1 had loosened these cases to 4e-3 instead).
"""
import os

import numpy as np
import pytest

import oracle
from gpu_util import maxabs, oracle_heads, rand_u, to_np

pytestmark = pytest.mark.gpu

TOL32 = 1e-4
TOLBF = 1e-3
TOLBF_CAUSAL = TOLBF   # round 1: 4e-3 (see the header); kept as a name so the causal cases stay visible
FLT_MAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def ops():
    from flash_attention_minitorch_amd import CudaKernelOps
    return CudaKernelOps


@pytest.fixture(scope="module")
def dev():
    import torch
    from flash_attention_minitorch_amd import device_ops
    assert torch.cuda.is_available()
    return device_ops


def _golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    bh, n, d = g["q"].shape
    shp = (1, bh, n, d)
    return g, shp


def operand_rounding_envelope(q, k):
    """Round 3: the MFMA-slot kernels (bf16, d = 64 / 128) fold c = tau*log2(e) into one bf16 operand: one more 2^-9 relative rounding
    of every q_d (k_d), which moves a score S' = c q.k (log2 units) by up to 2^-9 * sum_d |c q_d k_d| >= 2^-9 |S'|, i.e. P by that
    relative amount.  Nothing at the north star's U(-1, 1) inputs (|S'| < 4: the 1e-3 bound holds, every other test); on adversarial
    inputs with scores of tens the tests allow the slot kernels 2^-10 * max |S'| relative to the tensor's scale (half that worst
    case) and hold the kernels with fp32 scaling (OPTS_EXACT_SCALE) to the bound they always had."""
    d = q.shape[-1]
    smax = float(np.max(np.abs(np.einsum("bnd,bmd->bnm", q.astype(np.float64), k.astype(np.float64))))) * 1.4426950408889634 / np.sqrt(d)
    return max(5e-3, smax / 1024.0)


VARIANTS = [("flash_attn_fw", "flash_attn_bw", 1), ("flash_attn2_fw", "flash_attn2_bw", 2),
            ("flash_attn_causal_fw", "flash_attn_causal_bw", 1)]


# ---------------------------------------------------------------- reference FFI vs the reference's own outputs
@pytest.mark.parametrize("name", ["c0_b1h2n128d64", "ragged_n40d32", "ragged_n327d34"])
@pytest.mark.parametrize("fw,bw,conv", VARIANTS)
def test_host_abi_matches_reference_golden(ops, golden_dir, name, fw, bw, conv):
    g, shp = _golden(golden_dir, name)
    q, k, v, do = (g[x].reshape(shp) for x in ("q", "k", "v", "do_rand"))
    o, l, m = getattr(ops, fw)(q, k, v, False)
    assert o.shape == shp and l.shape == shp[:3] and m.shape == shp[:3] and o.dtype == np.float32
    assert maxabs(o[0], g["fa1_o_f64"]) < TOL32
    if conv == 1:   # FA-1 side outputs: l = sum exp(s - m), m = row max   (src/flash_attn_fw.cu:259-276)
        assert maxabs(m[0], g["fa1_m_f64"]) < 1e-5
        assert np.max(np.abs(l[0] / g["fa1_l_f64"] - 1)) < 1e-5
        assert maxabs(m[0] + np.log(l[0]), g["fa1_m_f64"] + np.log(g["fa1_l_f64"])) < TOL32
    else:           # FA-2: l = logsumexp, m untouched (src/flash_attn2_fw.cu:279-294)
        assert maxabs(l[0], g["fa2_L_f64"]) < TOL32
        assert np.all(m == -FLT_MAX)
    dq, dk, dv, cm = getattr(ops, bw)(q, k, v, o, do, l, m, False)
    assert cm is False
    fam = "fa1" if conv == 1 else "fa2"
    assert maxabs(dq[0], g[f"{fam}_dq_rand_f64"]) < TOL32
    assert maxabs(dk[0], g[f"{fam}_dk_rand_f64"]) < TOL32
    assert maxabs(dv[0], g[f"{fam}_dv_rand_f64"]) < TOL32


@pytest.mark.parametrize("name", ["c0_b1h2n128d64", "ragged_n40d32", "ragged_n327d34"])
@pytest.mark.parametrize("fw,bw,conv", VARIANTS)
def test_host_abi_causal_and_ones_grad(ops, golden_dir, name, fw, bw, conv):
    """All the reference's GPU tests run causal=True with out_grad = ones (kernel_tests/test_flashattn_bw.py:32)."""
    g, shp = _golden(golden_dir, name)
    q, k, v = (g[x].reshape(shp) for x in ("q", "k", "v"))
    do = np.ones(shp, np.float32)
    o, l, m = getattr(ops, fw)(q, k, v, np.array([1.0]))
    ref = oracle_heads(q[0], k[0], v[0], do[0], True, range(shp[1]))
    assert maxabs(o[0], ref["o"]) < TOL32
    L = m[0] + np.log(l[0]) if conv == 1 else l[0]
    assert maxabs(L, ref["L"]) < TOL32
    if conv == 1:
        assert maxabs(m[0], ref["m"]) < 1e-5
    dq, dk, dv, _ = getattr(ops, bw)(q, k, v, o, do, l, m, np.array([1.0]))
    assert maxabs(dq[0], ref["dq"]) < TOL32
    assert maxabs(dk[0], ref["dk"]) < TOL32
    assert maxabs(dv[0], ref["dv"]) < TOL32


# ---------------------------------------------------------------- the reference's own test shapes, full batch
def test_reference_comb_test_shape(ops):
    """kernel_tests/test_flashattn_comb.py:86-90: B=128, H=8, N=40, d=32, non-causal flash_attn + backward with
    out_grad = ones (N is not a multiple of any tile size)."""
    rng = np.random.default_rng(86)
    shp = (128, 8, 40, 32)
    q, k, v = (rand_u(rng, shp) for _ in range(3))
    do = np.ones(shp, np.float32)
    o, l, m = ops.flash_attn_fw(q, k, v, False)
    dq, dk, dv, _ = ops.flash_attn_bw(q, k, v, o, do, l, m, False)
    f = lambda a: a.reshape(1024, 40, -1)
    heads = range(0, 1024, 17)
    ref = oracle_heads(f(q), f(k), f(v), f(do), False, heads)
    idx = list(heads)
    assert maxabs(f(o)[idx], ref["o"]) < TOL32
    assert maxabs(m.reshape(1024, 40)[idx], ref["m"]) < 1e-5
    assert maxabs(f(dq)[idx], ref["dq"]) < TOL32
    assert maxabs(f(dk)[idx], ref["dk"]) < TOL32
    assert maxabs(f(dv)[idx], ref["dv"]) < TOL32


def test_reference_odd_head_dim_shape(ops):
    """kernel_tests/test_flashattn_2_fw.py:133-140: B=8, H=8, N=327, d=34 (FA-2 forward, causal); d is padded to 64
    inside the host launcher, tau stays sqrt(1/34)."""
    rng = np.random.default_rng(133)
    shp = (8, 8, 327, 34)
    q, k, v, do = (rand_u(rng, shp) for _ in range(4))
    o, l, m = ops.flash_attn2_fw(q, k, v, True)
    dq, dk, dv, _ = ops.flash_attn2_bw(q, k, v, o, do, l, m, True)
    f = lambda a: a.reshape(64, 327, -1)
    heads = [0, 9, 31, 63]
    ref = oracle_heads(f(q), f(k), f(v), f(do), True, heads)
    assert maxabs(f(o)[heads], ref["o"]) < TOL32
    assert maxabs(l.reshape(64, 327)[heads], ref["L"]) < TOL32
    assert np.all(m == -FLT_MAX)
    assert maxabs(f(dq)[heads], ref["dq"]) < TOL32
    assert maxabs(f(dk)[heads], ref["dk"]) < TOL32
    assert maxabs(f(dv)[heads], ref["dv"]) < TOL32


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_reference_odd_head_dim_shape_device_resident(dev, dtype):
    """The same shape (N = 327, d = 34, causal; kernel_tests/test_flashattn_2_fw.py:133-140) through the device-resident operator
    surface: any d <= 128 is zero-padded to the next of {32, 64, 128} on the device (fa_mi355x_*_padded), tau keeps the caller's d.
    Also d = 100 (padded to 128) and d = 20 (padded to 32), FA-1 side outputs."""
    import torch
    from flash_attention_minitorch_amd import _lib
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    tol = TOL32 if dtype == "f32" else TOLBF
    for (BH, N, d, causal, variant) in ((8, 327, 34, True, _lib.FA_VARIANT_FA2), (3, 200, 100, False, _lib.FA_VARIANT_FA1),
                                        (4, 129, 20, True, _lib.FA_VARIANT_FA2)):
        rng = np.random.default_rng(133 + d)
        arrs = [rand_u(rng, (BH, N, d)) for _ in range(4)]
        if dtype == "bf16":
            arrs = [oracle.bf16_round(a) for a in arrs]
        t = [torch.from_numpy(a).to("cuda", tdt) for a in arrs]
        o, l, m = dev.flash_attn_fwd(*t[:3], causal, variant)
        dq, dk, dv = dev.flash_attn_bwd(*t[:3], o, t[3], l, m, causal, variant)
        assert o.shape == (BH, N, d) and dq.shape == (BH, N, d) and o.is_contiguous()
        ref = oracle_heads(*arrs, causal, range(BH))
        L = l if variant == _lib.FA_VARIANT_FA2 else m + torch.log(l)
        for nm, got in (("o", o), ("L", L), ("dq", dq), ("dk", dk), ("dv", dv)):
            assert maxabs(to_np(got), ref[nm]) < tol, (d, nm, maxabs(to_np(got), ref[nm]))
    # the autograd contract on an odd head dim
    q, k, v = (torch.from_numpy(a).to("cuda", tdt).requires_grad_(True) for a in arrs[:3])
    out = dev.flash_attn2(q, k, v, True)
    out.backward(torch.from_numpy(arrs[3]).to("cuda"))
    assert maxabs(to_np(q.grad), ref["dq"]) < (tol if dtype == "f32" else 1e-2)   # (bf16: the gradient is cast to the input dtype)


# ---------------------------------------------------------------- BASELINE.json configs[1], configs[2] (fp32, full size)
def test_c1_fa1_forward_fp32_full(ops):
    rng = np.random.default_rng(1001)
    shp = (8, 8, 1024, 64)
    q, k, v = (rand_u(rng, shp) for _ in range(3))
    o, l, m = ops.flash_attn_fw(q, k, v, False)
    ref = oracle_heads(q.reshape(64, 1024, 64), k.reshape(64, 1024, 64), v.reshape(64, 1024, 64), None, False, range(64))
    assert maxabs(o.reshape(64, 1024, 64), ref["o"]) < TOL32
    assert maxabs(m.reshape(64, 1024), ref["m"]) < 1e-5
    assert maxabs((m + np.log(l)).reshape(64, 1024), ref["L"]) < TOL32


@pytest.mark.parametrize("causal", [False, True])
def test_c2_fa1_forward_backward_fp32_full(ops, causal):
    rng = np.random.default_rng(1002)
    shp = (8, 8, 2048, 64)
    q, k, v, do = (rand_u(rng, shp) for _ in range(4))
    fw, bw = (ops.flash_attn_causal_fw, ops.flash_attn_causal_bw) if causal else (ops.flash_attn_fw, ops.flash_attn_bw)
    o, l, m = fw(q, k, v, causal)
    dq, dk, dv, _ = bw(q, k, v, o, do, l, m, causal)
    f = lambda a: a.reshape(64, 2048, -1)
    heads = range(0, 64, 3)   # 22 of 64 heads; every (b, h) runs the same code, the sample bounds oracle time
    ref = oracle_heads(f(q), f(k), f(v), f(do), causal, heads)
    idx = list(heads)
    assert maxabs(f(o)[idx], ref["o"]) < TOL32
    assert maxabs(f(dq)[idx], ref["dq"]) < TOL32
    assert maxabs(f(dk)[idx], ref["dk"]) < TOL32
    assert maxabs(f(dv)[idx], ref["dv"]) < TOL32


@pytest.mark.parametrize("N", [256, 512, 1280, 257, 300, 1000])   # (the last three: ragged last key block / last stage)
@pytest.mark.parametrize("bnhd", [False, True])
@pytest.mark.parametrize("causal", [False, True])
def test_fp32_one_pass_backward(dev, N, bnhd, causal):
    """fp32, d = 64, N >= 256: the backward of a launch that fills the chip is ONE kernel (bwd_onepass_f32_kernel: the reference's five
    products, src/flash_attn2_bw.cu:94-247, dQ by fp32 atomics into a q_grad the LIBRARY zero-fills).  Against the fp64 oracle at the
    fp32 tolerance, against the two-kernel path (option 4 = 4: dk, dv bitwise -- the same per-key arithmetic -- and dq to summation
    order), both layouts and both side-output conventions, with and without the causal mask (diagonal stages: idle waves, the masked wave, the
    shortened dQ sum), on gradient buffers that hold NaN on entry."""
    import torch
    from flash_attention_minitorch_amd import _lib
    B, H, d = 2, 3, 64
    ONE, TWO = (0, 0, 0, 0, 5), (0, 0, 0, 0, 4)   # (6 heads do not fill the chip: the default would take two kernels here, see below)
    f32, fa2 = _lib.FA_DTYPE_F32, _lib.FA_VARIANT_FA2
    assert _lib.plan(B * H, N, d, causal, fa2, f32, dev.STAGE_ALL, ONE) == ["bwd_prep_kernel", "bwd_onepass_f32_kernel"]
    two = _lib.plan(B * H, N, d, causal, fa2, f32, dev.STAGE_ALL, TWO)
    assert "bwd_onepass_f32_kernel" not in two and len(two) >= 2, two
    assert "bwd_onepass_f32_kernel" not in _lib.plan(B * H, 200, d, False, fa2, f32, dev.STAGE_ALL, ONE)
    # the default takes the one-pass kernel when its launch (batch * ceil(N / 256) workgroups of one per CU) runs in rounds >= 80 % full
    # (below one workgroup per CU the query sweep of a key block is cut into 2, 4 or 8 parts: 16 heads x 8 blocks x 2, 4 x 8 x 8)
    for bh, want in ((64, True), (32, True), (16, True), (40, True), (4, True), (3, False), (2, False), (512, True)):
        assert ("bwd_onepass_f32_kernel" in _lib.plan(bh, 2048, d, causal, fa2, f32, dev.STAGE_ALL, None)) == want, bh
    rng = np.random.default_rng(77 + N)
    arrs = [rand_u(rng, (B * H, N, d)) for _ in range(4)]
    ref = oracle_heads(*arrs, causal, range(B * H))
    t4 = [torch.from_numpy(a).to("cuda").view(B, H, N, d) for a in arrs]
    if bnhd:
        q, k, v, do = (t.permute(0, 2, 1, 3).contiguous() for t in t4)
        back = lambda g: to_np(g.permute(0, 2, 1, 3).contiguous()).reshape(B * H, N, d)
    else:
        q, k, v, do = (t.reshape(B * H, N, d) for t in t4)
        back = to_np
    for variant in (_lib.FA_VARIANT_FA1, _lib.FA_VARIANT_FA2):
        if bnhd:
            o, l, m = dev.flash_attn_fwd_bnhd(q, k, v, causal, variant)
            g1 = dev.flash_attn_bwd_bnhd(q, k, v, o, do, l, m, causal, variant, opts=ONE)
            g2 = dev.flash_attn_bwd_bnhd(q, k, v, o, do, l, m, causal, variant, opts=TWO)
        else:
            o, l, m = dev.flash_attn_fwd(q, k, v, causal, variant)
            nan = lambda: tuple(torch.full(q.shape, float("nan"), dtype=torch.float32, device="cuda") for _ in range(3))
            g1 = dev.flash_attn_bwd(q, k, v, o, do, l, m, causal, variant, grads=nan(), opts=ONE)
            g2 = dev.flash_attn_bwd(q, k, v, o, do, l, m, causal, variant, grads=nan(), opts=TWO)
        for nm, a, b in zip(("dq", "dk", "dv"), g1, g2):
            assert maxabs(back(a), ref[nm]) < TOL32, (nm, variant)
            if nm == "dq" or N > 1024:   # (N = 1280 with 6 heads: the forced one-pass call cuts every sweep into 8 parts, dk / dv are sums too)
                assert float((a - b).abs().max()) < 5e-6 * max(1.0, float(b.abs().max()))
            else:
                assert torch.equal(a, b), nm


@pytest.mark.parametrize("BH,N,causal", [(16, 2048, False), (16, 2048, True), (4, 2048, False), (8, 1024, True), (20, 1000, False),
                                         (7, 2304, True)])
def test_fp32_one_pass_backward_split_sweeps(dev, BH, N, causal):
    """Launches below one workgroup per CU: the DEFAULT cuts the query sweep of every key block into 2 / 4 / 8 parts (one workgroup
    each) and sums dK, dV over the parts with atomics into gradients the library zero-fills, as dQ.  Against the fp64 oracle on
    sampled heads and against the two-kernel path on all of them, on gradient buffers that hold NaN on entry; ragged N and the causal
    mask (parts that start inside and behind the diagonal stages) included."""
    import torch
    from flash_attention_minitorch_amd import _lib
    assert "bwd_onepass_f32_kernel" in _lib.plan(BH, N, 64, causal, _lib.FA_VARIANT_FA2, _lib.FA_DTYPE_F32, dev.STAGE_ALL, None)
    rng = np.random.default_rng(BH * 10000 + N)
    arrs = [rand_u(rng, (BH, N, 64)) for _ in range(4)]
    q, k, v, do = (torch.from_numpy(a).to("cuda") for a in arrs)
    o, l, m = dev.flash_attn_fwd(q, k, v, causal)
    nan = lambda: tuple(torch.full(q.shape, float("nan"), dtype=torch.float32, device="cuda") for _ in range(3))
    g1 = dev.flash_attn_bwd(q, k, v, o, do, l, m, causal, grads=nan())
    g2 = dev.flash_attn_bwd(q, k, v, o, do, l, m, causal, grads=nan(), opts=(0, 0, 0, 0, 4))
    heads = sorted({0, BH // 2, BH - 1})
    ref = oracle_heads(*arrs, causal, heads)
    for nm, a, b in zip(("dq", "dk", "dv"), g1, g2):
        assert bool(torch.isfinite(a).all()), nm
        # (two summation orders of a few thousand fp32 terms: the oracle bound below is the parity check)
        assert float((a - b).abs().max()) < 5e-6 * max(1.0, float(b.abs().max())), nm
        assert maxabs(to_np(a)[heads], ref[nm]) < TOL32, nm


@pytest.mark.parametrize("BH,N", [(1, 128), (3, 129), (2, 200), (8, 1024), (5, 1000), (3, 1056), (7, 160)])
@pytest.mark.parametrize("causal", [False, True])
def test_fp32_split_key_forward(dev, BH, N, causal):
    """fp32, d = 64, launches that leave most of the chip idle: the default forward is fwd_splitk_f32_kernel (a workgroup = one 32-query
    block, wave w the key tiles w, w + 4, ...; classic online softmax per wave as src/flash_attn_fw.cu:163-245, the four partial (O, l, m)
    combined through LDS).  Against the fp64 oracle at the fp32 tolerance (O, L, and FA-1's m and l), both side-output conventions,
    ragged N, waves without a tile (N = 128, 160) or without an admissible key under the causal mask; and against the phased forward."""
    import torch
    from flash_attention_minitorch_amd import _lib
    assert _lib.plan(BH, N, 64, causal, _lib.FA_VARIANT_FA2, _lib.FA_DTYPE_F32, 0, None) == ["fwd_splitk_f32_kernel"]
    assert _lib.plan(BH, N, 64, causal, _lib.FA_VARIANT_FA2, _lib.FA_DTYPE_F32, 0, (0, 2)) == ["fwd_kernel"]
    assert _lib.plan(512, 1024, 64, causal, _lib.FA_VARIANT_FA2, _lib.FA_DTYPE_F32, 0, None) == ["fwd_kernel"]   # a launch that fills the chip
    rng = np.random.default_rng(BH * 7919 + N)
    arrs = [rand_u(rng, (BH, N, 64)) for _ in range(3)]
    q, k, v = (torch.from_numpy(a).to("cuda") for a in arrs)
    ref = oracle_heads(*arrs, None, causal, range(BH))
    for variant in (_lib.FA_VARIANT_FA1, _lib.FA_VARIANT_FA2):
        o, l, m = dev.flash_attn_fwd(q, k, v, causal, variant)
        o2, _, _ = dev.flash_attn_fwd(q, k, v, causal, variant, opts=(0, 2))
        assert maxabs(to_np(o), ref["o"]) < TOL32
        assert float((o - o2).abs().max()) < 1e-5
        if variant == _lib.FA_VARIANT_FA1:
            assert maxabs(to_np(m), ref["m"]) < 1e-5 and maxabs(to_np(l), ref["l"]) < TOL32 * max(1.0, float(np.max(ref["l"])))
            assert maxabs(to_np(m) + np.log(to_np(l)), ref["L"]) < TOL32
        else:
            assert maxabs(to_np(l), ref["L"]) < TOL32


# ---------------------------------------------------------------- bf16 device path: metric shape M and configs[3]
def _bf16_case(dev, B, H, N, d, causal, heads, seed):
    import torch
    rng = np.random.default_rng(seed)
    arrs = [oracle.bf16_round(rand_u(rng, (B * H, N, d))) for _ in range(4)]
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    o, L, _ = dev.flash_attn_fwd(tq, tk, tv, causal=causal)
    dq, dk, dv = dev.flash_attn_bwd(tq, tk, tv, o, tdo, L, causal=causal)
    torch.cuda.synchronize()
    ref = oracle_heads(arrs[0], arrs[1], arrs[2], arrs[3], causal, heads)
    idx = torch.tensor(list(heads), device="cuda")
    errs = {}
    for nm, got in (("o", o), ("L", L), ("dq", dq), ("dk", dk), ("dv", dv)):
        errs[nm] = maxabs(to_np(got[idx]), ref[nm])
    return errs, (o, L, dq, dk, dv), (tq, tk, tv, tdo)


@pytest.mark.parametrize("causal", [False, True])
def test_metric_shape_bf16_fa2_forward_backward(dev, causal):
    """B=8 H=8 N=4096 d=64 bf16 FA-2 fw+bw: the shape BASELINE.json's metric is quoted on."""
    # heads: every in-group position of the tiled dK/dV build (four consecutive heads per workgroup) and three XCDs
    errs, _, _ = _bf16_case(dev, 8, 8, 4096, 64, causal, [0, 1, 2, 3, 29, 37, 62, 63], 1004)
    for nm, e in errs.items():
        assert e < (TOLBF_CAUSAL if causal else TOLBF), (nm, e)


def test_c3_bf16_d128_forward_backward(dev):
    """configs[3]: B=16 H=16 N=4096 d=128 bf16 FA-2 fw+bw (d=128 is outside the reference FA-2 kernel's own
    envelope, src/flash_attn2_fw.cu:13,43; parity rests on the dense oracle)."""
    errs, _, _ = _bf16_case(dev, 16, 16, 4096, 128, False, [5, 102, 171, 250], 1003)
    for nm, e in errs.items():
        assert e < TOLBF, (nm, e)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("d", [32, 64, 128])
@pytest.mark.parametrize("N", [1, 33, 129, 200, 384])
@pytest.mark.parametrize("causal", [False, True])
def test_device_path_small_shapes(dev, dtype, d, N, causal):
    """Ragged N (tail masking), every head dim fast path, both dtypes, FA-1 and FA-2 side outputs."""
    import torch
    from flash_attention_minitorch_amd import _lib
    rng = np.random.default_rng(N * 1000 + d)
    BH = 3
    arrs = [rand_u(rng, (BH, N, d)) for _ in range(4)]
    if dtype == "bf16":
        arrs = [oracle.bf16_round(a) for a in arrs]
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", tdt) for a in arrs)
    tol = (TOLBF_CAUSAL if causal else TOLBF) if dtype == "bf16" else TOL32
    ref = oracle_heads(*arrs, causal, range(BH))
    for variant in (_lib.FA_VARIANT_FA1, _lib.FA_VARIANT_FA2):
        o, l, m = dev.flash_attn_fwd(tq, tk, tv, causal=causal, variant=variant)
        dq, dk, dv = dev.flash_attn_bwd(tq, tk, tv, o, tdo, l, m, causal=causal, variant=variant)
        L = to_np(m) + np.log(to_np(l)) if variant == _lib.FA_VARIANT_FA1 else to_np(l)
        assert maxabs(to_np(o), ref["o"]) < tol
        assert maxabs(L, ref["L"]) < tol
        if variant == _lib.FA_VARIANT_FA1:
            assert maxabs(to_np(m), ref["m"]) < (1e-5 if dtype == "f32" else tol)
        assert maxabs(to_np(dq), ref["dq"]) < tol
        assert maxabs(to_np(dk), ref["dk"]) < tol
        assert maxabs(to_np(dv), ref["dv"]) < tol


@pytest.mark.parametrize("N", [255, 256, 257, 512, 513, 768, 1000, 1280, 2048])
@pytest.mark.parametrize("causal", [False, True])
def test_slot_kernels_block_and_ring_boundaries(dev, N, causal):
    """bf16, d = 64, FA-2: the slot-interleaved forward / dQ / dK-dV kernels around their structural boundaries --
    256-query workgroups, 128-key stages, the three-slot K/V ring wrapping (>= 4 stages), ragged tails and the causal
    diagonal inside a stage -- each compared with the phased kernels (tuning keys) and with the oracle.  Causal with N a
    multiple of 256 runs the causal builds of the forward / dQ slot kernels (unmasked sweep + the diagonal block per wave,
    query blocks p and nqb-1-p paired: one, an odd and an even number of blocks, rings wrapping); other N the masked builds."""
    import torch
    from flash_attention_minitorch_amd import _lib
    rng = np.random.default_rng(7000 + N)
    BH, d = 2, 64
    arrs = [oracle.bf16_round(rand_u(rng, (BH, N, d))) for _ in range(4)]
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    tol = TOLBF_CAUSAL if causal else TOLBF
    ref = oracle_heads(*arrs, causal, range(BH))
    outs = {}
    # options 1, 2 = 3 / option 0 = 5 force the slot kernels under the causal mask whatever the launch size; option 7: causal builds
    # with one block per workgroup in ranked order (2) or with blocks p and nqb-1-p paired (1)
    k0 = 5 if causal else 0
    variants = [("slot", (k0, 3, 3, 0, 0, 0, 0, 2)), ("phased", dev.OPTS_PHASED)]
    if causal and N % 256 == 0:
        variants.append(("slot_paired", (k0, 3, 3, 0, 0, 0, 0, 1)))
    for tag, opts in variants:
        o, l, m = dev.flash_attn_fwd(tq, tk, tv, causal=causal, opts=opts)
        dq, dk, dv = dev.flash_attn_bwd(tq, tk, tv, o, tdo, l, m, causal=causal, opts=opts)
        outs[tag] = [to_np(x) for x in (o, l, dq, dk, dv)]
    for tag in outs:
        if tag != "phased":
            for nm, got in zip(("o", "L", "dq", "dk", "dv"), outs[tag]):
                assert np.all(np.isfinite(got)), (tag, nm)
                assert maxabs(got, ref[nm]) < tol, (tag, nm, maxabs(got, ref[nm]))
    # Same arithmetic per element, different tiling of the key loop: the two builds agree far inside the tolerance (0.5 * tol) --
    # except where the slot build carries tau*log2(e) in its bf16 operand instead of an fp32 fma per score (round 3): the mask-free
    # forward / dQ builds (N a multiple of 128 without the mask, of 256 with it) and the dK/dV slot kernel always.  There both are
    # within tol of the oracle and within tol of each other (a tiling bug would be orders of magnitude off).
    folded = N % 256 == 0 or (not causal and N % 128 == 0)
    for nm, a, b in zip(("o", "L", "dq", "dk", "dv"), outs["slot"], outs["phased"]):
        lim = tol if (folded or nm in ("dk", "dv")) else 0.5 * tol
        assert maxabs(a, b) < lim, (nm, maxabs(a, b))


@pytest.mark.parametrize("dtype,d", [("bf16", 64), ("bf16", 128), ("f32", 32), ("f32", 64)])
@pytest.mark.parametrize("causal", [False, True])
def test_key_mask_forward_backward(dev, dtype, d, causal):
    """additive [B, N] key mask (src/softmax_kernel.cu:27-34,77-90 semantics) inside the fused
    kernels, FA-1 and FA-2 side outputs, against the oracle's masked_attention_* (itself tied to the pinned dense
    oracle by identities, tests/test_oracle_golden.py).  Padding (-inf tail), random dropped keys, a finite bias and,
    under the causal rule, rows whose every admissible key is dropped (O = 0, L = -inf, zero gradients)."""
    import torch
    from flash_attention_minitorch_amd import _lib
    rng = np.random.default_rng(40 + d)
    B, H, N = 2, 3, 200
    arrs = [rand_u(rng, (B, H, N, d)) for _ in range(4)]
    if dtype == "bf16":
        arrs = [oracle.bf16_round(a) for a in arrs]
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", tdt) for a in arrs)
    mask = np.zeros((B, N), dtype=np.float32)
    mask[0, 150:] = -np.inf                                    # padded tail
    mask[1, rng.uniform(size=N) < 0.3] = -np.inf               # scattered drops
    mask[1, 5:9] = np.float32(-1.5)                            # finite bias
    mask[1, 0] = -np.inf                                       # causal: query 0 of batch 1 has no key left
    tmask = torch.from_numpy(mask).cuda()
    tol = (TOLBF_CAUSAL if causal else TOLBF) if dtype == "bf16" else TOL32
    ro, rL = oracle.masked_attention_fw(*arrs[:3], mask[:, None, :], causal)
    rg = oracle.masked_attention_bw(*arrs, mask[:, None, :], causal)
    for variant in (_lib.FA_VARIANT_FA1, _lib.FA_VARIANT_FA2):
        o, l, m = dev.flash_attn_fwd_masked(tq, tk, tv, tmask, causal, variant)
        dq, dk, dv = dev.flash_attn_bwd_masked(tq, tk, tv, o, tdo, l, m, tmask, causal, variant)
        with np.errstate(divide="ignore", invalid="ignore"):
            L = to_np(m) + np.log(to_np(l)) if variant == _lib.FA_VARIANT_FA1 else to_np(l)
        dead = np.isneginf(rL)
        assert dead.any() == causal                              # only the causal case has fully dropped rows
        assert np.array_equal(np.isneginf(L), dead)
        assert maxabs(np.where(dead, 0, L), np.where(dead, 0, rL)) < tol
        assert np.all(np.isfinite(to_np(o))) and maxabs(to_np(o), ro) < tol
        for nm, got, ref in (("dq", dq, rg[0]), ("dk", dk, rg[1]), ("dv", dv, rg[2])):
            assert np.all(np.isfinite(to_np(got))), nm
            assert maxabs(to_np(got), ref) < tol, (nm, maxabs(to_np(got), ref))
        # dropped keys receive exactly zero gradient
        drop = np.isneginf(mask)
        for b in range(B):
            assert np.all(to_np(dk)[b][:, drop[b]] == 0) and np.all(to_np(dv)[b][:, drop[b]] == 0)
    # a NULL / all-zero mask is the unmasked operator, bit for bit (same kernels when NULL; same arithmetic when zero)
    o0, l0, _ = dev.flash_attn_fwd(tq, tk, tv, causal)
    oz, lz, _ = dev.flash_attn_fwd_masked(tq, tk, tv, torch.zeros((B, N), device="cuda"), causal)
    assert maxabs(to_np(oz), to_np(o0)) < tol * 0.5 and maxabs(to_np(lz), to_np(l0)) < tol * 0.5


@pytest.mark.parametrize("dtype,d", [("bf16", 64), ("bf16", 128), ("f32", 32)])
@pytest.mark.parametrize("causal", [False, True])
def test_dropout_forward_backward(dev, dtype, d, causal):
    """SURVEY.md row f4, dropout: the in-kernel stateless mask equals the oracle's NumPy restatement bit for bit, so
    forward and backward match the oracle's dropout attention on the same mask; the three kernels regenerate the SAME
    mask (dV, dK, dQ consistent with O); with and without a key mask; rate 0 is the masked operator."""
    import torch
    from flash_attention_minitorch_amd import _lib
    rng = np.random.default_rng(50 + d)
    B, H, N = 2, 2, 200
    rate, seed = 0.2, 0xC0FFEE
    scale = 1.0 / (1.0 - rate)
    arrs = [rand_u(rng, (B, H, N, d)) for _ in range(4)]
    if dtype == "bf16":
        arrs = [oracle.bf16_round(a) for a in arrs]
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", tdt) for a in arrs)
    keep = oracle.dropout_keep_mask(B * H, N, rate, seed)
    assert 0.75 < keep.mean() < 0.85
    mask = np.zeros((B, N), dtype=np.float32)
    mask[1, 160:] = -np.inf
    tol = (TOLBF_CAUSAL if causal else TOLBF) if dtype == "bf16" else TOL32
    tol *= scale
    for km, tkm in ((None, None), (mask, torch.from_numpy(mask).cuda())):
        okm = None if km is None else km[:, None, :]
        ro, rL = oracle.dropout_attention_fw(*arrs[:3], keep, scale, okm, causal)
        rg = oracle.dropout_attention_bw(*arrs, keep, scale, okm, causal)
        for variant in (_lib.FA_VARIANT_FA1, _lib.FA_VARIANT_FA2):
            o, l, m = dev.flash_attn_fwd_dropout(tq, tk, tv, rate, seed, scale, tkm, causal, variant)
            dq, dk, dv = dev.flash_attn_bwd_dropout(tq, tk, tv, o, tdo, l, m, rate, seed, scale, tkm, causal, variant)
            L = to_np(m) + np.log(to_np(l)) if variant == _lib.FA_VARIANT_FA1 else to_np(l)
            assert maxabs(to_np(o), ro) < tol, maxabs(to_np(o), ro)
            assert maxabs(L, rL) < tol
            for nm, got, ref in (("dq", dq, rg[0]), ("dk", dk, rg[1]), ("dv", dv, rg[2])):
                assert maxabs(to_np(got), ref) < tol, (nm, maxabs(to_np(got), ref))
    # a different seed is a different mask; rate 0 is the plain operator
    o_a, _, _ = dev.flash_attn_fwd_dropout(tq, tk, tv, rate, seed, scale, None, causal)
    o_b, _, _ = dev.flash_attn_fwd_dropout(tq, tk, tv, rate, seed + 1, scale, None, causal)
    assert maxabs(to_np(o_a), to_np(o_b)) > 10 * tol
    o_0, l_0, _ = dev.flash_attn_fwd_dropout(tq, tk, tv, 0.0, seed, 1.0, None, causal)
    o_p, l_p, _ = dev.flash_attn_fwd(tq, tk, tv, causal)
    assert maxabs(to_np(o_0), to_np(o_p)) < tol and maxabs(to_np(l_0), to_np(l_p)) < tol


@pytest.mark.parametrize("N", [64, 192, 320, 1024])
def test_forward_slot_kernel_d128(dev, N):
    """bf16, d = 128, FA-2, non-causal, N a multiple of its 64-key stage: the slot-interleaved forward (two sub-tiles
    per stage, four-slot ring, barrier at the top of the stage) against the phased kernel and the oracle; 1, 3, 5 and 16
    stages exercise the prologue's double DMA, the ring wrapping and the 256-query workgroup edge."""
    import torch
    from flash_attention_minitorch_amd import _lib
    rng = np.random.default_rng(8000 + N)
    BH, d = 3, 128
    arrs = [oracle.bf16_round(rand_u(rng, (BH, N, d))) for _ in range(3)]
    tq, tk, tv = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    ro, rL, _, _ = oracle.dense_attention_fw(*arrs)
    o_s, l_s, _ = dev.flash_attn_fwd(tq, tk, tv)
    o_p, l_p, _ = dev.flash_attn_fwd(tq, tk, tv, opts=(0, 2))
    assert maxabs(to_np(o_s), ro) < TOLBF and maxabs(to_np(l_s), rL) < TOLBF
    # the two kernels set their softmax reference on 32 vs 64 keys, so P is rounded to bf16 at different scales:
    # each is within TOLBF of the oracle, their difference within the sum
    assert maxabs(to_np(o_s), to_np(o_p)) < 1.5 * TOLBF and maxabs(to_np(l_s), to_np(l_p)) < 0.5 * TOLBF   # (see the comment above)


@pytest.mark.parametrize("BH,N", [(256, 512), (512, 256), (64, 4096), (320, 1024), (256, 2048)])
def test_tiled_dkdv_build_is_bitwise_the_one_head_build(dev, BH, N):
    """bf16, d = 64, non-causal, N a multiple of 256, launches of at least one workgroup per CU per head group: the dK/dV slot
    kernel's tiled build (key block kb of 2 / 2 / 4 / 5 / 8 consecutive heads per workgroup, the ring and the pipeline carried from
    head to head) and the dQ slot kernel's (query block qb of as many consecutive heads per workgroup, round 3) against the
    one-head-per-workgroup builds (option 5 = 1): the same arithmetic in the same order, so bit for bit; two heads (the first of a
    group and the last one) against the oracle."""
    import torch
    rng = np.random.default_rng(9100 + BH + N)
    d = 64
    arrs = [oracle.bf16_round(rand_u(rng, (BH, N, d))) for _ in range(4)]
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    o, l, _ = dev.flash_attn_fwd(tq, tk, tv)
    tiled = [to_np(x) for x in dev.flash_attn_bwd(tq, tk, tv, o, tdo, l)]
    plain = [to_np(x) for x in dev.flash_attn_bwd(tq, tk, tv, o, tdo, l, opts=(0, 0, 0, 0, 0, 1))]
    for nm, a, b in zip(("dq", "dk", "dv"), tiled, plain):
        assert np.array_equal(a, b), nm
    heads = [0, BH - 1]
    ref = oracle_heads(*arrs, False, heads)
    for nm, got in zip(("dq", "dk", "dv"), tiled):
        assert maxabs(got[heads], ref[nm]) < TOLBF, (nm, maxabs(got[heads], ref[nm]))


@pytest.mark.parametrize("BH,N", [(127, 256), (128, 256), (255, 256), (256, 256), (64, 512), (63, 512), (32, 1024), (257, 256)])
@pytest.mark.parametrize("causal", [False, True])
def test_default_dispatch_around_launch_size_thresholds(dev, BH, N, causal):
    """bf16, d = 64, default options: the launcher picks slot / phased, tiled / one-head, ranked / paired builds and the folded
    preprocess by launch size (128 / 256 blocks, one workgroup per CU per head group, batch*head a multiple of 8 or not).  Shapes on
    both sides of each threshold: every head against the phased kernels with the separate preprocess kernel (options (4, 2, 2, 0, 1):
    same arithmetic per element), first and last head against the oracle."""
    import torch
    rng = np.random.default_rng(9300 + BH + N)
    d = 64
    arrs = [oracle.bf16_round(rand_u(rng, (BH, N, d))) for _ in range(4)]
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    outs = {}
    for tag, opts in (("default", None), ("phased", (4, 2, 2, 0, 1))):
        o, l, _ = dev.flash_attn_fwd(tq, tk, tv, causal=causal, opts=opts)
        dq, dk, dv = dev.flash_attn_bwd(tq, tk, tv, o, tdo, l, None, causal=causal, opts=opts)
        outs[tag] = [to_np(x) for x in (o, l, dq, dk, dv)]
    for nm, a, b in zip(("o", "L", "dq", "dk", "dv"), outs["default"], outs["phased"]):
        assert np.all(np.isfinite(a)), nm
        assert maxabs(a, b) < TOLBF, (nm, maxabs(a, b))
    heads = [0, BH - 1]
    ref = oracle_heads(*arrs, causal, heads)
    for nm, got in zip(("o", "L", "dq", "dk", "dv"), outs["default"]):
        assert maxabs(got[heads], ref[nm]) < TOLBF, (nm, maxabs(got[heads], ref[nm]))


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("opts", [(3,), (4,), (5,), (0, 2), (0, 3), (0, 0, 2), (0, 0, 3), (0, 0, 0, 0, 1), (0, 0, 0, 0, 0, 1),
                                  (0, 0, 0, 0, 0, 0, 0, 1), (0, 0, 0, 0, 0, 0, 0, 2)])
def test_every_accepted_option_value_against_the_oracle(dev, opts, causal):
    """Every per-call option value the product library accepts (include/flash_attn_mi355x.h; the values that lost their A/B moved
    to the diagnostic library in round 3), one at a time, bf16 d = 64 at a size where the launch-size rules would otherwise pick for
    themselves: forward + backward against the fp64 oracle at the ordinary bound."""
    import torch
    BH, N, d = 16, 768, 64
    rng = np.random.default_rng(8800)
    arrs = [oracle.bf16_round(rand_u(rng, (BH, N, d))) for _ in range(4)]
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    o, l, m = dev.flash_attn_fwd(tq, tk, tv, causal=causal, opts=opts)
    dq, dk, dv = dev.flash_attn_bwd(tq, tk, tv, o, tdo, l, m, causal=causal, opts=opts)
    heads = [0, 7, 15]
    ref = oracle_heads(*arrs, causal, heads)
    for nm, got in (("o", o), ("L", l), ("dq", dq), ("dk", dk), ("dv", dv)):
        assert maxabs(to_np(got)[heads], ref[nm]) < TOLBF, (opts, nm, maxabs(to_np(got)[heads], ref[nm]))


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("causal", [False, True])
def test_scale_guard_routes_by_operand_size_without_a_host_sync(dev, causal, d):
    """Round 4 (VERDICT r3 missing 3): the default call is safe outside U(-1, 1).  device_ops issues GUARDED calls: one
    device-side pass over q and k (fa_mi355x_scale_guard), then every kernel that folds tau*log2(e) into a bf16 operand is launched
    beside its fp32-scaling twin and the workgroups of the wrong side return at once.  Checked by bitwise identity: at U(-1, 1) the
    default equals the explicitly folded call (option 8 = 1), at x2 and x6 it equals the explicit fp32-scaling call (option 8 = 2),
    whose x2 results meet 1e-3 * scale where the folded kernels do not (profiles/r03_prescale_accuracy.txt); the decision is the
    host formula's (pick_opts), taken on the device; the plain C entry points (no guard) scale in fp32."""
    import ctypes
    import torch
    from flash_attention_minitorch_amd import _lib
    rng = np.random.default_rng(8800 + d)
    BH, N = 4, 512
    for amp in (1.0, 1.25, 2.0, 6.0):
        arrs = [oracle.bf16_round(amp * rand_u(rng, (BH, N, d))) for _ in range(3)] + [oracle.bf16_round(rand_u(rng, (BH, N, d)))]
        t = [torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs]
        guard = dev.scale_guard(t[0], t[1])
        gq, gk = float(guard[:256].max()), float(guard[256:].max())
        assert abs(gq - float(t[0].float().pow(2).sum(-1).max())) < 1e-3 * gq and abs(gk - float(t[1].float().pow(2).sum(-1).max())) < 1e-3 * gk
        want_exact = dev.pick_opts(t[0], t[1]) == dev.OPTS_EXACT_SCALE
        assert want_exact == (amp >= 2.0) or amp == 1.25   # (1.25: folded at d = 64, fp32 scaling at d = 128, where U(-1, 1) sits at 0.7 of the budget)

        def run(opts, guard_arg):
            o, L, _ = dev.flash_attn_fwd(*t[:3], causal, opts=opts, guard=guard_arg)
            g = dev.flash_attn_bwd(*t[:3], o, t[3], L, None, causal, opts=opts, guard=guard_arg)
            return [to_np(x) for x in (o, L) + tuple(g)]

        default = run(None, "auto")
        shared = run(None, guard)                       # one guard pass for forward and backward
        # ... and without any separate pass: the forward FILLS a guard inside its own launch (optimistically, redone by its twin when
        # beyond the budget) and the backward reads it; the maxima it leaves are those of the separate pass
        g2 = dev.new_guard(t[0])
        o_, L_, _ = dev.flash_attn_fwd(*t[:3], causal, guard=g2, produce_guard=True)
        gr = dev.flash_attn_bwd(*t[:3], o_, t[3], L_, None, causal, guard=g2)
        produced = [to_np(x) for x in (o_, L_) + tuple(gr)]
        assert abs(float(g2[:256].max()) - gq) < 1e-5 * gq and abs(float(g2[256:].max()) - gk) < 1e-5 * gk
        assert all(np.array_equal(a, b) for a, b in zip(produced, default))
        folded = run(dev.OPTS_FOLDED_SCALE, None)
        exact = run(dev.OPTS_EXACT_SCALE, None)
        same = exact if want_exact else folded
        for a, b, c in zip(default, shared, same):
            assert np.array_equal(a, b) and np.array_equal(a, c), (amp, maxabs(a, c))
        bf = _lib.FA_DTYPE_BF16
        differ = _lib.plan(BH, N, d, causal, 2, bf, 0, dev.OPTS_FOLDED_SCALE) != _lib.plan(BH, N, d, causal, 2, bf, 0, dev.OPTS_EXACT_SCALE)
        if differ:
            assert not np.array_equal(folded[0], exact[0])   # (they ARE different kernels; a causal launch this small runs the phased ones anyway)
        # the plain C entry point carries no guard: fp32 scaling
        o2 = torch.empty((BH, N, d), dtype=torch.float32, device="cuda")
        l2 = torch.empty((BH, N), dtype=torch.float32, device="cuda")
        vp = lambda x: ctypes.c_void_p(x.data_ptr())
        _lib.check(_lib.core().fa_mi355x_fwd(vp(t[0]), vp(t[1]), vp(t[2]), vp(o2), vp(l2), None, BH, N, d, int(causal), 2, 1, None))
        assert np.array_equal(to_np(o2), exact[0]) and np.array_equal(to_np(l2), exact[1])
        if amp == 2.0:   # the point of it: at x2 the default now has the fp32-scaling kernels' error (5e-3 * scale at this short N,
            # where the bf16 rounding of P is averaged over 512 keys only), and on L, which sums every key of a row, it is well below
            # the folded kernels' (2.9e-3 against 1e-6 at the metric shape, profiles/r03_prescale_accuracy.txt)
            ref = oracle_heads(*arrs, causal, range(BH))
            for nm, got in zip(("o", "L", "dq", "dk", "dv"), default):
                scale = max(1.0, float(np.max(np.abs(ref[nm]))))
                assert maxabs(got, ref[nm]) < 5e-3 * scale, (nm, maxabs(got, ref[nm]), scale)
            if differ:
                assert maxabs(default[1], ref["L"]) < 0.5 * maxabs(folded[1], ref["L"])


@pytest.mark.parametrize("mode", [1, 2])
def test_causal_slot_builds_do_not_touch_rows_beyond_a_waves_horizon(dev, mode):
    """in the causal slot builds a wave of query block 0 with no stage to sweep (forward: waves 0-1, dQ: waves 0-3) ran
    its prologue and drain periods on the diagonal block's first stage with P = 0 / dS = 0, i.e. 0 * V (rows 32..63) and 0 * K (rows
    96..127) on rows beyond its causal horizon: NaN if such a row holds Inf, where the reference (which masks,
    src/flash_attn_fw.cu:152-159, and skips tiles wholly above the diagonal) is finite.  V row 40 = Inf: queries below 32 (another
    32-row tile) must come out of the forward finite and unchanged; K row 100 = Inf: queries below 96 out of the dQ kernel.  Folded
    scale (mode 1) and fp32 scaling (mode 2: the phased forward, the dQ slot kernel's fp32-scaling sweep)."""
    import torch
    from flash_attention_minitorch_amd import _lib
    rng = np.random.default_rng(4242)
    BH, N, d = 128, 512, 64      # (256 query blocks: the causal slot builds are the default selection)
    arrs = [oracle.bf16_round(rand_u(rng, (BH, N, d))) for _ in range(4)]
    t = [torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs]
    opts = (5, 3, 3, 0, 0, 0, 0, 0, mode)
    names = _lib.plan(BH, N, d, True, 2, _lib.FA_DTYPE_BF16, 0, opts) + _lib.plan(BH, N, d, True, 2, _lib.FA_DTYPE_BF16, 7, opts)
    assert "bwd_dq_slot_kernel" in names and (mode == 2 or "fwd_slot_kernel" in names)
    o0, L0, _ = dev.flash_attn_fwd(*t[:3], True, opts=opts)
    dq0 = dev.flash_attn_bwd(*t[:3], o0, t[3], L0, None, True, opts=opts)[0]
    bad_v = t[2].clone()
    bad_v[:, 40] = float("inf")
    if mode == 1:   # (the phased forward of mode 2 multiplies whole 64-key tiles, as the reference multiplies its own tiles: row 40 is in the first)
        o1, L1, _ = dev.flash_attn_fwd(t[0], t[1], bad_v, True, opts=opts)
        assert torch.isfinite(o1[:, :32]).all() and torch.equal(o0[:, :32], o1[:, :32]) and torch.equal(L0, L1)
    bad_k = t[1].clone()
    bad_k[:, 100] = float("inf")
    dq1 = dev.flash_attn_bwd(t[0], bad_k, t[2], o0, t[3], L0, None, True, opts=opts)[0]
    assert torch.isfinite(dq1[:, :96]).all() and torch.equal(dq0[:, :96], dq1[:, :96])


def test_pick_opts_keeps_the_north_star_domain_on_the_fast_kernels(dev):
    """device_ops.pick_opts (the scale guard's decision taken on the host, once per tensor family): U(-1, 1) operands may run the
    folded-scale MFMA-slot kernels unguarded, operands a few times larger are sent to the kernels with fp32 scaling, and with that
    choice the x6 inputs of test_large_magnitude_inputs_stay_finite meet the 5e-3 bound."""
    import torch
    rng = np.random.default_rng(61)
    BH, N, d = 4, 512, 64
    mk = lambda s: torch.from_numpy(oracle.bf16_round(s * rand_u(rng, (BH, N, d)))).to("cuda", torch.bfloat16)
    assert dev.pick_opts(mk(1.0), mk(1.0)) == dev.OPTS_FOLDED_SCALE
    assert dev.pick_opts(mk(1.0).float(), mk(1.0).float()) is None and dev.pick_opts(mk(6.0), mk(6.0)) == dev.OPTS_EXACT_SCALE
    assert dev.pick_opts(mk(2.0), mk(2.0)) == dev.OPTS_EXACT_SCALE
    arrs = [oracle.bf16_round(6.0 * rand_u(rng, (BH, N, d))) for _ in range(3)] + [oracle.bf16_round(rand_u(rng, (BH, N, d)))]
    t = [torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs]
    opts = dev.pick_opts(t[0], t[1])
    o, L, _ = dev.flash_attn_fwd(*t[:3], opts=opts)
    dq, dk, dv = dev.flash_attn_bwd(*t[:3], o, t[3], L, opts=opts)
    ref = oracle_heads(*arrs, False, range(BH))
    for nm, got in (("o", o), ("L", L), ("dq", dq), ("dk", dk), ("dv", dv)):
        scale = max(1.0, float(np.max(np.abs(ref[nm]))))
        assert maxabs(to_np(got), ref[nm]) < 5e-3 * scale, (nm, maxabs(to_np(got), ref[nm]), scale)


def test_rejected_option_values(dev):
    import torch
    from flash_attention_minitorch_amd import _lib
    q = torch.zeros((2, 256, 64), device="cuda", dtype=torch.bfloat16)
    for bad in ((1,), (2,), (0, 6), (0, 0, 1), (0, 0, 4), (0, 0, 0, 1), (0, 0, 0, 0, 2), (0, 0, 0, 0, 0, 0, 1), (93,)):
        with pytest.raises(_lib.FlashAttnLibraryError, match="diagnostic"):
            dev.flash_attn_fwd(q, q, q, opts=bad)


@pytest.mark.parametrize("BH,N", [(24, 512), (5, 768), (40, 256), (16, 2560), (72, 1024)])
@pytest.mark.parametrize("order", [1, 2])
def test_causal_slot_builds_block_order(dev, BH, N, order):
    """The causal builds of the slot kernels map workgroup ids to (batch*head, block) either paired (option 7 = 1: blocks p and
    nqb-1-p in one workgroup) or ranked (2: one block per workgroup, heads taken in chunks per XCD, longest block first): every
    block of every head must be visited exactly once whatever batch*head (a multiple of 8 or not, fewer or more heads per XCD than
    a chunk) and the block count (odd, even, more than a chunk's share).  All heads against the phased kernels (same arithmetic
    per element), three heads against the oracle."""
    import torch
    rng = np.random.default_rng(9000 + BH + N)
    d = 64
    arrs = [oracle.bf16_round(rand_u(rng, (BH, N, d))) for _ in range(4)]
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    outs = {}
    for tag, opts in (("slot", (5, 3, 3, 0, 0, 0, 0, order)), ("phased", dev.OPTS_PHASED)):
        o, l, _ = dev.flash_attn_fwd(tq, tk, tv, causal=True, opts=opts)
        dq, dk, dv = dev.flash_attn_bwd(tq, tk, tv, o, tdo, l, None, causal=True, opts=opts)
        outs[tag] = [to_np(x) for x in (o, l, dq, dk, dv)]
    for nm, a, b in zip(("o", "L", "dq", "dk", "dv"), outs["slot"], outs["phased"]):
        assert np.all(np.isfinite(a)), nm
        assert maxabs(a, b) < TOLBF, (nm, maxabs(a, b))
    heads = [0, BH // 2, BH - 1]
    ref = oracle_heads(*arrs, True, heads)
    for nm, got in zip(("o", "L", "dq", "dk", "dv"), outs["slot"]):
        assert maxabs(got[heads], ref[nm]) < TOLBF, (nm, maxabs(got[heads], ref[nm]))


@pytest.mark.parametrize("N", [256, 768, 1024, 2304])
def test_causal_forward_slot_kernel_d128(dev, N):
    """bf16, d = 128, FA-2, causal, N a multiple of 256: the causal build of the slot-interleaved forward (unmasked sweep of the
    keys in front of the workgroup's diagonal block, the block wave by wave, query blocks p and nqb-1-p paired; option 1 = 3
    forces it whatever the launch size) against the phased kernel and the oracle: 1, 3, 4 and 9 query blocks."""
    import torch
    rng = np.random.default_rng(8100 + N)
    BH, d = 3, 128
    arrs = [oracle.bf16_round(rand_u(rng, (BH, N, d))) for _ in range(3)]
    tq, tk, tv = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    ro, rL, _, _ = oracle.dense_attention_fw(*arrs, causal=True)
    o_s, l_s, _ = dev.flash_attn_fwd(tq, tk, tv, causal=True, opts=(0, 3))
    o_p, l_p, _ = dev.flash_attn_fwd(tq, tk, tv, causal=True, opts=(0, 2))
    assert maxabs(to_np(o_s), ro) < TOLBF and maxabs(to_np(l_s), rL) < TOLBF
    assert maxabs(to_np(o_p), ro) < TOLBF and maxabs(to_np(l_p), rL) < TOLBF
    assert maxabs(to_np(o_s), to_np(o_p)) < 1.5 * TOLBF


def test_one_pass_backward_in_the_diagnostic_library():
    """The one-pass backward (csrc/fa_bwd_fused.h: dQ formed in the key-stationary kernel and summed across the key-block workgroups
    of a head by an ordered hand-off; the reference's single pass, src/flash_attn2_bw.cu:94-247) left the product library in round 3
    (it lost its A/B at every size and was the one kernel with scratch and a spin protocol) and lives in the diagnostic build.
    tools/check_fused.py, run as a CHILD process (this process never loads the diagnostic library), checks it against the fp64
    oracle (1e-3), against the two-kernel backward, bitwise repeatability and the hand-off status word at six shapes incl. B=8, H=8,
    N=4096."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_fused.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_chained_one_pass_backward_in_the_diagnostic_library():
    """Round 4: the five-product backward of the reference (src/flash_attn2_bw.cu:94-247: S, dP, dV, dK, dQ in ONE
    key-stationary pass, dQ summed over key blocks with atomicAdd at :228) as csrc/fa_bwd_chain.h builds it: a workgroup takes
    consecutive key blocks of a head and carries its running dQ tiles through memory; fp32 atomics only from the last block of each
    chain, none when a workgroup covers a whole head.  It measured slower than the two-kernel backward (profiles/r04_chain_backward.txt)
    and keeps 36 B of scratch, so it lives in the diagnostic build.  tools/check_chain.py, run as a CHILD process, checks it at eight
    shapes (1-4 chains per head, 1-16 blocks per chain, B=8 H=8 N=4096 among them) against the fp64 oracle (1e-3), the two-kernel
    backward (2e-3) and itself (run to run: 1e-5; the atomics' arrival order moves dq in the last bits only)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FA_MI355X_DIAG="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_chain.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype,d", [("f32", 64), ("bf16", 128), ("bf16", 32)])
@pytest.mark.parametrize("N", [33, 200])
def test_ragged_tail_with_very_negative_logsumexp(dev, dtype, d, N):
    """keys past N in the last tile of the phased dQ kernel read K = 0, so P = exp(-L); with a row logsumexp
    below about -88 that is +inf and dS = inf * 0 poisoned the row's dQ.  q = -s * k drives every score far below -100."""
    import torch
    rng = np.random.default_rng(9300 + N + d)
    BH = 2
    k = rand_u(rng, (BH, N, d)) + np.float32(1.5)            # all keys alike and of one sign: every score is very negative
    q = (-16.0 * np.ones((BH, N, d))).astype(np.float32) * np.abs(rand_u(rng, (BH, N, d)) + np.float32(1.5))
    v, do = rand_u(rng, (BH, N, d)), rand_u(rng, (BH, N, d))
    arrs = [q, k, v, do]
    if dtype == "bf16":
        arrs = [oracle.bf16_round(a) for a in arrs]
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", tdt) for a in arrs)
    ref = oracle_heads(*arrs, False, range(BH))
    assert np.max(ref["L"]) < -100.0
    o, l, m = dev.flash_attn_fwd(tq, tk, tv)
    dq, dk, dv = dev.flash_attn_bwd(tq, tk, tv, o, tdo, l, m)
    # |q| is up to 40 here, so the bound is relative to each tensor's magnitude (bf16: P, dS enter the second MFMA at 2^-9 relative)
    tol = 8 * TOLBF if dtype == "bf16" else TOL32
    for nm, got in (("dq", dq), ("dk", dk), ("dv", dv)):
        g = to_np(got)
        assert np.all(np.isfinite(g)), nm
        scale = max(1.0, float(np.max(np.abs(ref[nm]))))
        assert maxabs(g, ref[nm]) < tol * scale, (nm, maxabs(g, ref[nm]), scale)


@pytest.mark.parametrize("shape,causal", [((8, 8, 4096, 64), False), ((16, 8, 2048, 34), True), ((3, 5, 1500, 64), False)])
def test_host_abi_pipeline_multi_chunk(ops, shape, causal):
    """The host-pointer launchers cut a call into chunks of batch*head (H2D / kernels / D2H on three streams, caller arrays pinned
    in place): shapes that take 2-8 chunks, one with zero-padded head dim (d = 34 -> 64) and one whose chunks are ragged
    (15 heads), forward and backward, sampled heads against the oracle."""
    import ctypes
    from flash_attention_minitorch_amd import _lib
    rng = np.random.default_rng(3300 + shape[2])
    q, k, v, do = (rand_u(rng, shape) for _ in range(4))

    def pin_stats():
        a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        _lib.core().fa_mi355x_host_pin_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    ok0, fb0 = pin_stats()
    o, l, m = ops.flash_attn2_fw(q, k, v, causal)
    dq, dk, dv, _ = ops.flash_attn2_bw(q, k, v, o, do, l, m, causal)
    ok1, fb1 = pin_stats()
    # the caller's arrays are copied pageable by default (in-place pinning is opt-in, FA_MI355X_HOST_PIN=1: test below): nothing registered
    assert (ok1, fb1) == (ok0, fb0) == (0, 0), (ok0, fb0, ok1, fb1)
    B, H, N, d = shape
    f = lambda a: a.reshape(B * H, N, -1)
    heads = [0, B * H // 2, B * H - 1]
    ref = oracle_heads(f(q), f(k), f(v), f(do), causal, heads)
    assert maxabs(f(o)[heads], ref["o"]) < TOL32 and maxabs(l.reshape(B * H, N)[heads], ref["L"]) < TOL32
    assert np.all(m == -FLT_MAX)
    for nm, g in (("dq", dq), ("dk", dk), ("dv", dv)):
        assert maxabs(f(g)[heads], ref[nm]) < TOL32, nm
    # same call again with q, k, v aliased (one array registered three times cannot be pinned twice: pageable fallback)
    o2, l2, _ = ops.flash_attn2_fw(q, q, q, causal)
    ro, rL, _, _ = oracle.dense_attention_fw(f(q)[:1], f(q)[:1], f(q)[:1], causal)
    assert maxabs(f(o2)[:1], ro) < TOL32 and maxabs(l2.reshape(B * H, N)[:1], rL) < TOL32


def test_host_abi_in_place_pinning_is_opt_in():
    """FA_MI355X_HOST_PIN=1 (a CHILD process: the switch is read once): the host launchers register the caller's arrays in place
    (page-aligned merged ranges >= 4 MiB), none falls back to pageable copies, and the results are those of the default path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import ctypes, numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from flash_attention_minitorch_amd import CudaKernelOps as ops, _lib\n"
        "import oracle\n"
        "rng = np.random.default_rng(5)\n"
        "q, k, v, do = (rng.uniform(-1, 1, (4, 8, 1024, 64)).astype(np.float32) for _ in range(4))\n"
        "o, l, m = ops.flash_attn2_fw(q, k, v, False)\n"
        "dq, dk, dv, _ = ops.flash_attn2_bw(q, k, v, o, do, l, m, False)\n"
        "a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)\n"
        "_lib.core().fa_mi355x_host_pin_stats(ctypes.byref(a), ctypes.byref(b))\n"
        "assert a.value > 0 and b.value == 0, (a.value, b.value)\n"
        "f = lambda t: t.reshape(32, 1024, 64)\n"
        "ro = oracle.dense_attention_fw(f(q)[5:6], f(k)[5:6], f(v)[5:6], False)[0]\n"
        "rg = oracle.dense_attention_bw(f(q)[5:6], f(k)[5:6], f(v)[5:6], f(do)[5:6], False)\n"
        "assert np.max(np.abs(f(o)[5:6] - ro)) < 1e-4\n"
        "for g, r in zip((dq, dk, dv), rg): assert np.max(np.abs(f(g)[5:6] - r)) < 1e-4\n"
        "print('pinned ok', a.value)\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, FA_MI355X_HOST_PIN="1"))
    assert r.returncode == 0 and "pinned ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_native_backtrace_hook_fires_on_abort(tmp_path):
    """tests/abort_trace.c, as conftest.py installs it on GPU runs, on the GPU box itself (gpu_util.check_abort_hook; the CPU suite runs
    the same check)."""
    from gpu_util import check_abort_hook
    check_abort_hook(tmp_path)


def test_host_abi_arrays_that_share_pages(golden_dir):
    """The host launchers pin the caller's arrays in place (hipHostRegister).  NumPy arrays of a few MiB come from the malloc heap once
    glibc has raised its mmap threshold, so the arrays of ONE call can be adjacent and share boundary pages.  Round 2 registered every
    array on its own: overlapping registrations then made a later hipHostUnregister fail (surfacing as the NEXT call's
    "pointer does not correspond to a registered memory region" + exit) or left the runtime with a stale pinned range (an abort in a
    later, unrelated copy): about one GPU-suite run in six.  Round 3 registers page-aligned MERGED ranges.  Here every array of a
    forward and a backward call is carved, unaligned and back to back, out of ONE buffer (l and m >= 1 MiB so that they are pinned
    too), through the reference's own FFI symbols, twice, then an ordinary call follows."""
    import ctypes
    from flash_attention_minitorch_amd import _lib
    from flash_attention_minitorch_amd.cuda_kernel_ops import _FW_ARGTYPES, _BW_ARGTYPES, CudaKernelOps
    BH, N, d = 128, 2048, 32
    nt, nr = BH * N * d, BH * N
    sizes = [nt] * 8 + [nr] * 2                       # q k v out dout dq dk dv | l m
    buf = np.zeros(sum(sizes) + 64 * len(sizes) + 1024, np.float32)
    views, off = [], 3                                # 12 bytes off the buffer's alignment, 4-byte gaps' worth of odd spacing
    for i, n in enumerate(sizes):
        views.append(buf[off:off + n])
        off += n + 1 + 2 * i                          # adjacent: consecutive arrays share their boundary page
    q, k, v, out, dout, dq, dk, dv, l, m = views
    rng = np.random.default_rng(515)
    for a in (q, k, v, dout):
        a[:] = rand_u(rng, a.shape)
    fw, bw = _lib.load("flash_attn2_fw.so"), _lib.load("flash_attn2_bw.so")
    fw.launch_flash_attn_fw.argtypes, bw.launch_flash_attn_bw.argtypes = _FW_ARGTYPES, _BW_ARGTYPES
    fw.launch_flash_attn_fw.restype = bw.launch_flash_attn_bw.restype = None
    null = ctypes.c_void_p(0)
    for _ in range(2):
        m[:] = -FLT_MAX
        fw.launch_flash_attn_fw(q, k, v, out, l, m, BH, N, d, False, null)
        bw.launch_flash_attn_bw(q, k, v, out, dout, dq, dk, dv, l, m, BH, N, d, False, null)
    f = lambda a: a.reshape(BH, N, -1)
    heads = [0, 77, 127]
    ref = oracle_heads(f(q), f(k), f(v), f(dout), False, heads)
    assert maxabs(f(out)[heads], ref["o"]) < TOL32 and maxabs(l.reshape(BH, N)[heads], ref["L"]) < TOL32
    for nm, g in (("dq", dq), ("dk", dk), ("dv", dv)):
        assert maxabs(f(g)[heads], ref[nm]) < TOL32, nm
    # ... and the next, ordinary call (fresh arrays) and a device-path round trip are unaffected
    q4 = rand_u(rng, (2, 2, 256, 64))
    o4, l4, _ = CudaKernelOps.flash_attn2_fw(q4, q4, q4, True)
    ro, rL, _, _ = oracle.dense_attention_fw(q4, q4, q4, True)
    assert maxabs(o4, ro) < TOL32 and maxabs(l4, rL) < TOL32


# ---------------------------------------------------------------- row f3: flash vs vanilla attention ON THE GPU
def _vanilla():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import vanilla_gpu
    return vanilla_gpu


@pytest.mark.parametrize("N", [128, 1024])
@pytest.mark.parametrize("fw,bw", [("flash_attn_fw", "flash_attn_bw"), ("flash_attn_causal_fw", "flash_attn_causal_bw"),
                                   ("flash_attn2_fw", "flash_attn2_bw")])
def test_flash_matches_gpu_vanilla_attention_reference_shapes(ops, fw, bw, N):
    """The reference pins flash == vanilla on the GPU: kernel_tests/test_flashattn_fw.py:23-156 (B=1, H=8, d=64, causal_mask=True,
    atol = rtol = 1e-3) and test_flashattn_bw.py:19-210 (out_grad = ones, atol 1e-2, rtol 1e-3), for its three operator variants.
    Vanilla here: torch-ROCm fp32 matmul + softmax with the reference's -FLT_MAX causal mask (tools/vanilla_gpu.py)."""
    import torch
    vg = _vanilla()
    rng = np.random.default_rng(2300 + N)
    shp = (1, 8, N, 64)
    q, k, v = (rand_u(rng, shp) for _ in range(3))
    do = np.ones(shp, np.float32)
    o, l, m = getattr(ops, fw)(q, k, v, True)
    dq, dk, dv, _ = getattr(ops, bw)(q, k, v, o, do, l, m, True)
    ro, rdq, rdk, rdv = (to_np(t) for t in vg.vanilla_fw_bw(*(torch.from_numpy(a).cuda() for a in (q, k, v, do)), True))
    np.testing.assert_allclose(o, ro, atol=1e-3, rtol=1e-3)
    for got, ref in ((dq, rdq), (dk, rdk), (dv, rdv)):
        np.testing.assert_allclose(got, ref, atol=1e-2, rtol=1e-3)
    # (and far inside those bounds: both sides are fp32)
    assert maxabs(o, ro) < TOL32 and max(maxabs(dq, rdq), maxabs(dk, rdk), maxabs(dv, rdv)) < 2e-4


def test_flash_matches_gpu_vanilla_attention_comb_shape(ops):
    """kernel_tests/test_flashattn_comb.py:86-90: B=128, H=8, N=40, d=32, non-causal, out_grad = ones."""
    import torch
    vg = _vanilla()
    rng = np.random.default_rng(2386)
    shp = (128, 8, 40, 32)
    q, k, v = (rand_u(rng, shp) for _ in range(3))
    do = np.ones(shp, np.float32)
    o, l, m = ops.flash_attn_fw(q, k, v, False)
    dq, dk, dv, _ = ops.flash_attn_bw(q, k, v, o, do, l, m, False)
    ro, rdq, rdk, rdv = (to_np(t) for t in vg.vanilla_fw_bw(*(torch.from_numpy(a).cuda() for a in (q, k, v, do)), False))
    np.testing.assert_allclose(o, ro, atol=1e-3, rtol=1e-3)
    for got, ref in ((dq, rdq), (dk, rdk), (dv, rdv)):
        np.testing.assert_allclose(got, ref, atol=1e-2, rtol=1e-3)


@pytest.mark.parametrize("causal", [False, True])
def test_device_bf16_flash_matches_gpu_vanilla_attention(dev, causal):
    """The device-resident bf16 path against the same GPU vanilla attention evaluated in fp32 on the bf16-rounded inputs."""
    import torch
    vg = _vanilla()
    gen = torch.Generator(device="cuda").manual_seed(5)
    mk = lambda: ((torch.rand((16, 1024, 64), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, l, _ = dev.flash_attn_fwd(q, k, v, causal)
    dq, dk, dv = dev.flash_attn_bwd(q, k, v, o, do, l, None, causal)
    ro, rdq, rdk, rdv = vg.vanilla_fw_bw(q.float(), k.float(), v.float(), do.float(), causal)
    for nm, a, b in (("o", o, ro), ("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv)):
        assert float((a - b).abs().max()) < TOLBF, nm


def test_random_shapes_bf16(dev):
    """Seeded sweep over sequence lengths that straddle every structural size of the bf16 kernels (32-key sub-tiles, 64 / 128
    key stages, 128 / 256 query workgroups, 3 / 4 slot rings), head dims, causal flag and batch*head counts that do and do
    not divide by the 8 XCDs -- whichever kernel the dispatcher picks (slot, masked-slot or phased) must meet the bf16 bound."""
    import torch
    rng = np.random.default_rng(20261004)
    cases = []
    for _ in range(28):
        d = int(rng.choice([32, 64, 64, 64, 128, 128]))
        N = int(rng.choice([rng.integers(1, 70), rng.integers(100, 300), 64 * rng.integers(1, 9), 128 * rng.integers(1, 6),
                            rng.integers(300, 700)]))
        cases.append((N, d, bool(rng.integers(0, 2)), int(rng.choice([1, 3, 8, 16]))))
    for N, d, causal, BH in cases:
        arrs = [oracle.bf16_round(rand_u(rng, (BH, N, d))) for _ in range(4)]
        tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
        o, L, _ = dev.flash_attn_fwd(tq, tk, tv, causal)
        dq, dk, dv = dev.flash_attn_bwd(tq, tk, tv, o, tdo, L, None, causal)
        heads = range(min(BH, 3))
        ref = oracle_heads(*arrs, causal, heads)
        tol = TOLBF_CAUSAL if causal else TOLBF
        for nm, got in (("o", o), ("L", L), ("dq", dq), ("dk", dk), ("dv", dv)):
            g = to_np(got)[: len(heads)]
            assert np.all(np.isfinite(to_np(got))), (N, d, causal, BH, nm)
            assert maxabs(g, ref[nm]) < tol, (N, d, causal, BH, nm, maxabs(g, ref[nm]))


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("causal", [False, True])
def test_large_magnitude_inputs_stay_finite(dev, causal, d):
    """Scores of order +-100 (inputs x 6): exp2 arguments reach +-150 and nothing may overflow to inf / NaN on the way.  The phased
    kernels move their reference on most rows (P = exp2(c*s - c*m_ref) is computed BEFORE the guard is checked); the slot kernels'
    reference-free sweep (round 3) over- or underflows on many rows and their waves take the cold path (fwd_redo_rows).  The
    softmax is nearly one-hot here, so the bf16 bound is taken relative to the output scale.
    Accuracy envelope: the kernels with fp32 scaling (OPTS_EXACT_SCALE, and since round 4 the DEFAULT call, whose scale guard routes
    operands of this size to them) keep 5e-3 * scale; the slot kernels, which carry tau*log2(e) in a bf16 operand and run here only
    because option 8 = 1 vouches for the operands, must stay finite and inside operand_rounding_envelope (about 0.1 * scale)."""
    import torch
    rng = np.random.default_rng(77)
    BH, N = 2, 512
    # (d = 128: the forward slot kernel and the plain phased backward kernels carry the scale in their operand too; OPTS_EXACT_SCALE
    # runs the phased forward and the split-operand backward builds there)
    amp = 6.0 if d == 64 else 5.0   # (scores of the same order for either head dim)
    arrs = [oracle.bf16_round(amp * rand_u(rng, (BH, N, d))) for _ in range(3)] + [oracle.bf16_round(rand_u(rng, (BH, N, d)))]
    tq, tk, tv, tdo = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in arrs)
    ref = oracle_heads(*arrs, causal, range(BH))
    # the slot kernels whatever the launch size (causal: their causal builds) AND whatever the operands (option 8 = 1: no guard)
    slot = (5 if causal else 0, 3, 3, 0, 0, 0, 0, 0, 1)
    env = operand_rounding_envelope(arrs[0], arrs[1])
    # round 4: the DEFAULT call runs under the scale guard, which sends these operands to the fp32-scaling kernels: 5e-3 * scale again
    for opts, rel in ((dev.OPTS_EXACT_SCALE, 5e-3), (None, 5e-3), (slot, env)):
        o, L, _ = dev.flash_attn_fwd(tq, tk, tv, causal, opts=opts)
        dq, dk, dv = dev.flash_attn_bwd(tq, tk, tv, o, tdo, L, None, causal, opts=opts)
        for nm, got in (("o", o), ("L", L), ("dq", dq), ("dk", dk), ("dv", dv)):
            g = to_np(got)
            assert np.all(np.isfinite(g)), (opts, nm)
            scale = max(1.0, float(np.max(np.abs(ref[nm]))))
            assert maxabs(g, ref[nm]) < rel * scale, (opts, nm, maxabs(g, ref[nm]), scale)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("sign", [-1.0, 1.0])
def test_forward_cold_path_on_extreme_scores(dev, causal, sign):
    """Every row's scores sit near -135 or +135 in log2 units (q rows ~ 6u, k rows ~ +-6u): exp2(S') under- / overflows for EVERY key,
    so every wave of the reference-free slot forward leaves its sweep with a row sum outside [2^-96, 2^96] (0, inf or NaN) and redoes
    its rows in the wave-local cold path (fwd_redo_rows: classic running maximum, fp32 scaling, split P), whose results are held to
    the ordinary bound.  d = 64 and 128, non-causal and the causal builds."""
    import torch
    rng = np.random.default_rng(91)
    for d in (64, 128):
        BH, N = 2, 512
        u = rand_u(rng, (1, 1, d))
        q = oracle.bf16_round(6.0 * u + 0.25 * rand_u(rng, (BH, N, d)))
        k = oracle.bf16_round(sign * 6.0 * u * (8.0 / np.sqrt(d)) + 0.25 * rand_u(rng, (BH, N, d)))   # tau q.k ~ +-96 for either d
        v = oracle.bf16_round(rand_u(rng, (BH, N, d)))
        tq, tk, tv = (torch.from_numpy(a).to("cuda", torch.bfloat16) for a in (q, k, v))
        o, L, _ = dev.flash_attn_fwd(tq, tk, tv, causal, opts=(0, 3))
        ref = oracle_heads(q, k, v, None, causal, range(BH))
        smax = float(np.max(np.abs(ref["L"])))
        assert smax * 1.4427 > 110, smax   # the construction really leaves the sweep's range
        # causal: the waves of query block 0 (rows 0..255) never leave the range (no sweep: the diagonal block moves its reference the
        # classic way), so rows 64..255 keep the scaled operand there: L inside operand_rounding_envelope; all other rows are exact
        cold = slice(256, None) if causal else slice(None)
        for nm, got in (("o", o), ("L", L)):
            g = to_np(got)
            assert np.all(np.isfinite(g)), (d, nm)
            assert maxabs(g[:, cold], ref[nm][:, cold]) < (TOLBF if nm == "o" else 1e-5 * smax), (d, nm, maxabs(g[:, cold], ref[nm][:, cold]))
            assert maxabs(g, ref[nm]) < operand_rounding_envelope(q, k), (d, nm, maxabs(g, ref[nm]))


def test_long_sequence(dev):
    """N = 8192 against the oracle on one head, and N = 32768 through size-independent properties (no N^2 memory on
    either side: the reference's only sequence-length limit is time)."""
    import torch
    errs, _, _ = _bf16_case(dev, 1, 2, 8192, 64, False, [1], 2001)
    for nm, e in errs.items():
        assert e < TOLBF, (nm, e)
    errs, _, _ = _bf16_case(dev, 1, 2, 8192, 64, True, [0], 2002)
    for nm, e in errs.items():
        assert e < TOLBF_CAUSAL, (nm, e)
    torch.manual_seed(1)
    N = 32768
    mk = lambda: ((torch.rand((2, N, 64), device="cuda") - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    for causal in (False, True):
        o1, L1, _ = dev.flash_attn_fwd(q, k, torch.ones_like(v), causal=causal)
        assert (o1 - 1).abs().max().item() < (2e-3 if causal else 2e-4)   # causal: rows with 2-4 keys, bf16 P
        o, L, _ = dev.flash_attn_fwd(q, k, v, causal=causal)
        dq, dk, dv = dev.flash_attn_bwd(q, k, v, o, do, L, causal=causal)
        for t in (o, L, dq, dk, dv):
            assert torch.isfinite(t).all()
        assert (dv.sum(dim=1) - do.float().sum(dim=1)).abs().max().item() < 0.2     # columns of P^T sum: sum_n dV = sum_n dO
        lhs = (dq.double() * q.double()).sum(dim=(1, 2)); rhs = (dk.double() * k.double()).sum(dim=(1, 2))
        assert (lhs - rhs).abs().max().item() < 5e-2
        if not causal:   # L >= log(N) + min score, <= log(N) + max score
            assert (L - float(np.log(N))).abs().max().item() < 1.5


# ---------------------------------------------------------------- size-independent properties at full size
def test_properties_at_metric_shape(dev):
    import torch
    torch.manual_seed(0)
    B, H, N, d = 8, 8, 4096, 64
    mk = lambda: ((torch.rand((B * H, N, d), device="cuda") - 0.5) * 2).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    o, L, _ = dev.flash_attn_fwd(q, k, v)
    dq, dk, dv = dev.flash_attn_bwd(q, k, v, o, do, L)
    bad = []

    def check(name, ok, detail=""):
        if not ok:
            bad.append(f"{name} {detail}")

    # determinism: no atomics anywhere, so a second launch is bitwise identical
    o2, L2, _ = dev.flash_attn_fwd(q, k, v)
    dq2, dk2, dv2 = dev.flash_attn_bwd(q, k, v, o2, do, L2)
    for nm, a, b in (("o", o, o2), ("L", L, L2), ("dq", dq, dq2), ("dk", dk, dk2), ("dv", dv, dv2)):
        check(f"bitwise-repeatable {nm}", torch.equal(a, b))
        check(f"finite {nm}", bool(torch.isfinite(a).all()))
    # rows of P sum to one: V = ones -> O = ones (P is bf16 in the numerator, fp32 in the denominator: 2^-9/sqrt(N));
    # dO = 0 -> all gradients are exactly zero
    o1, _, _ = dev.flash_attn_fwd(q, k, torch.ones_like(v))
    e = (o1 - 1).abs().max().item()
    check("V=ones gives O=ones", e < 2e-4, f"{e:.3e}")
    z = dev.flash_attn_bwd(q, k, v, o, torch.zeros_like(do), L)
    check("dO=0 gives zero grads", all(t.abs().max().item() == 0 for t in z))
    # linearity in V (exact in bf16 for a power of two): O(q, k, 2v) = 2 O(q, k, v); L unchanged
    ob, Lb, _ = dev.flash_attn_fwd(q, k, (v.float() * 2).to(torch.bfloat16))
    check("O linear in V", torch.equal(ob, 2 * o) and torch.equal(Lb, L))
    # linearity of the backward in dO (delta and dP double exactly; products are rounded once more: 1e-6)
    g2 = dev.flash_attn_bwd(q, k, v, o, (do.float() * 2).to(torch.bfloat16), L)
    for nm, a, b in zip(("dq", "dk", "dv"), g2, (dq, dk, dv)):
        e = (a - 2 * b).abs().max().item()
        check(f"backward linear in dO ({nm})", e < 1e-4, f"{e:.3e}")
    # sum_n dV[n, :] = sum_n dO[n, :]  (every query distributes weight 1 over the keys)
    e = (dv.sum(dim=1) - do.float().sum(dim=1)).abs().max().item()
    check("column sums of dV", e < 5e-2, f"{e:.3e}")
    # softmax shift invariance: sum_j dS_ij = 0  =>  sum(dQ * Q) = sum(dK * K) per head
    lhs = (dq.double() * q.double()).sum(dim=(1, 2)); rhs = (dk.double() * k.double()).sum(dim=(1, 2))
    e = (lhs - rhs).abs().max().item()
    check("sum(dQ.Q) == sum(dK.K)", e < 2e-2, f"{e:.3e} (|lhs| up to {lhs.abs().max().item():.3e})")
    # batch*head independence: a slice computed alone equals the slice of the full launch
    sl = slice(17, 19)
    o_s, L_s, _ = dev.flash_attn_fwd(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous())
    check("batch*head independence", torch.equal(o_s, o[sl]) and torch.equal(L_s, L[sl]))
    # permuting the keys (with their values) leaves O unchanged up to summation order (non-causal)
    perm = torch.randperm(N, device="cuda")
    o_p, L_p, _ = dev.flash_attn_fwd(q[:4].contiguous(), k[:4, perm].contiguous(), v[:4, perm].contiguous())
    e1, e2 = (o_p - o[:4]).abs().max().item(), (L_p - L[:4]).abs().max().item()
    check("key permutation invariance", e1 < 1e-3 and e2 < 1e-4, f"{e1:.3e} {e2:.3e}")
    assert not bad, bad


def test_online_softmax_rescale_branch_is_exercised(dev):
    """A row maximum that jumps at a late K/V tile forces the O / l rescale path (bounded random data alone rarely
    moves the maximum after the first tiles): cdna_hip_programming.md section 5.4 rule 26."""
    import torch
    rng = np.random.default_rng(5)
    BH, N, d = 2, 512, 64
    q, k, v, do = (rand_u(rng, (BH, N, d)) for _ in range(4))
    for (row, key, scale) in ((3, 70, 6.0), (100, 300, 9.0), (257, 511, 12.0), (300, 129, 5.0)):
        k[:, key] = q[:, row] * scale
    arrs = [oracle.bf16_round(a) for a in (q, k, v, do)]
    env = operand_rounding_envelope(arrs[0], arrs[1])
    for causal in (False, True):
        for tdt, tol in ((torch.bfloat16, TOLBF), (torch.float32, TOL32)):
            # (0, 3, 3, .., 1): the slot kernels whatever the launch size and the operands (causal, N = 512: the diagonal-block phase moves the reference);
            # OPTS_EXACT_SCALE: the phased kernels, whose forward moves its reference at the spiked tiles
            for opts in ((None, (5 if causal else 0, 3, 3, 0, 0, 0, 0, 0, 1), dev.OPTS_EXACT_SCALE) if tdt == torch.bfloat16 else (None,)):
                t = [torch.from_numpy(a).to("cuda", tdt) for a in arrs]
                o, L, _ = dev.flash_attn_fwd(*t[:3], causal=causal, opts=opts)
                dq, dk, dv = dev.flash_attn_bwd(*t[:3], o, t[3], L, causal=causal, opts=opts)
                ref = oracle_heads(*arrs, causal, range(BH))
                for nm, got in (("o", o), ("L", L), ("dq", dq), ("dk", dk), ("dv", dv)):
                    # the spiked keys make |K| ~ 12 and gather P ~ 1 from many rows, so gradients reach O(10):
                    # the tolerance is relative to the tensor's scale here
                    # (the spiked rows put nearly all their weight on ONE key, so bf16 P / dS are not averaged: 5e-3)
                    # (the slot kernels carry tau*log2(e) in a bf16 operand: scores of ~45 on the spiked keys: operand_rounding_envelope)
                    scale = max(1.0, float(np.max(np.abs(ref[nm]))))
                    exact = tdt != torch.bfloat16 or opts is None or opts == dev.OPTS_EXACT_SCALE   # (None: the guard sees |k| ~ 12 |q|)
                    lim = ((5e-3 if exact else env) if tdt == torch.bfloat16 else tol) * scale
                    assert maxabs(to_np(got), ref[nm]) < lim, (causal, tdt, opts, nm, maxabs(to_np(got), ref[nm]), scale)


@pytest.mark.parametrize("N", [200, 256])   # ragged (masked slot builds / phased) and stage-aligned (mask-free slot builds)
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("causal", [False, True])
def test_bnhd_layout_matches_permuted_copy(dev, dtype, causal, N):
    """(B, N, H, d) in and out, no head-split copies.  Must be bit-identical to running the
    [B*H][N][d] path on permute(0,2,1,3).contiguous() copies (what minitorch/modules_transfomer.py:67-89 does)."""
    import torch
    from flash_attention_minitorch_amd import _lib
    torch.manual_seed(3)
    B, H, d = 2, 3, 64
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    q, k, v, do = (((torch.rand((B, N, H, d), device="cuda") - 0.5) * 2).to(tdt) for _ in range(4))
    perm = lambda t: t.permute(0, 2, 1, 3).contiguous()
    for variant in (_lib.FA_VARIANT_FA1, _lib.FA_VARIANT_FA2):
        o, l, m = dev.flash_attn_fwd_bnhd(q, k, v, causal, variant)
        dq, dk, dv = dev.flash_attn_bwd_bnhd(q, k, v, o, do, l, m, causal, variant)
        o_r, l_r, m_r = dev.flash_attn_fwd(perm(q), perm(k), perm(v), causal, variant)
        g_r = dev.flash_attn_bwd(perm(q), perm(k), perm(v), o_r, perm(do), l_r, m_r, causal, variant)
        assert o.shape == (B, N, H, d) and l.shape == (B, H, N)
        assert torch.equal(perm(o), o_r) and torch.equal(l, l_r)
        if m is not None:
            assert torch.equal(m, m_r)
        for a, b in zip((dq, dk, dv), g_r):
            assert torch.equal(perm(a), b)
        # ... and against the ORACLE directly (a HIP-vs-HIP comparison alone is not evidence)
        f = lambda t: to_np(perm(t).float())
        ref = oracle_heads(f(q).reshape(B * H, N, d), f(k).reshape(B * H, N, d), f(v).reshape(B * H, N, d),
                           f(do).reshape(B * H, N, d), causal, range(B * H))
        tol = TOLBF if dtype == "bf16" else TOL32
        L = to_np(l) if variant == _lib.FA_VARIANT_FA2 else to_np(m) + np.log(to_np(l))
        assert maxabs(f(o).reshape(B * H, N, d), ref["o"]) < tol and maxabs(L.reshape(B * H, N), ref["L"]) < tol
        for nm, a in (("dq", dq), ("dk", dk), ("dv", dv)):
            assert maxabs(f(a).reshape(B * H, N, d), ref[nm]) < tol, nm


@pytest.mark.parametrize("causal", [False, True])
def test_bnhd_layout_at_chip_filling_size(dev, causal):
    """[B][N][H][d] in place at a launch size that takes the builds small launches never reach: the tiled dK/dV kernel (key block kb of
    consecutive heads per workgroup: the next head sits H*d elements further, not N*d) and, under the causal mask, the causal slot
    builds with ranked block order.  Bit-identical to the [B*H][N][d] path on permuted copies; four heads against the oracle."""
    import torch
    torch.manual_seed(5)
    B, H, N, d = 32, 8, 512, 64
    q, k, v, do = (((torch.rand((B, N, H, d), device="cuda") - 0.5) * 2).to(torch.bfloat16) for _ in range(4))
    perm = lambda t: t.permute(0, 2, 1, 3).contiguous()
    o, l, m = dev.flash_attn_fwd_bnhd(q, k, v, causal)
    dq, dk, dv = dev.flash_attn_bwd_bnhd(q, k, v, o, do, l, m, causal)
    o_r, l_r, m_r = dev.flash_attn_fwd(perm(q), perm(k), perm(v), causal)
    g_r = dev.flash_attn_bwd(perm(q), perm(k), perm(v), o_r, perm(do), l_r, m_r, causal)
    assert torch.equal(perm(o), o_r) and torch.equal(l, l_r)
    for a, b in zip((dq, dk, dv), g_r):
        assert torch.equal(perm(a), b)
    heads = [0, 7, 100, B * H - 1]
    f = lambda t: to_np(perm(t).float()).reshape(B * H, N, d)
    ref = oracle_heads(f(q), f(k), f(v), f(do), causal, heads)
    for nm, a in (("o", o), ("dq", dq), ("dk", dk), ("dv", dv)):
        assert maxabs(f(a)[heads], ref[nm]) < TOLBF, nm


def test_autograd_functions_follow_reference_contract(dev):
    """Flash_Attn / Flash_Attn2 / Flash_Attn_Causal: forward returns o, backward yields one grad per tensor input
    (minitorch/tensor_functions.py:462-497)."""
    import torch
    rng = np.random.default_rng(9)
    arrs = [rand_u(rng, (1, 2, 96, 64)) for _ in range(4)]
    ref = oracle_heads(*(a[0] for a in arrs), True, range(2))
    for fn in (dev.flash_attn, dev.flash_attn2, dev.flash_attn_causal):
        q, k, v = (torch.from_numpy(a).cuda().requires_grad_(True) for a in arrs[:3])
        o = fn(q, k, v, True)
        o.backward(torch.from_numpy(arrs[3]).cuda())
        assert maxabs(to_np(o)[0], ref["o"]) < TOL32
        assert maxabs(to_np(q.grad)[0], ref["dq"]) < TOL32
        assert maxabs(to_np(k.grad)[0], ref["dk"]) < TOL32
        assert maxabs(to_np(v.grad)[0], ref["dv"]) < TOL32


def test_device_path_rejects_bad_input(dev):
    import torch
    from flash_attention_minitorch_amd import _lib
    q = torch.zeros((2, 16, 130), device="cuda")
    with pytest.raises(ValueError, match="d > 128"):   # the reference asserts d <= 128 (src/flash_attn_fw.cu:43)
        dev.flash_attn_fwd(q, q, q)
    with pytest.raises(_lib.FlashAttnLibraryError, match="no CPU fallback"):
        c = torch.zeros((2, 16, 64))
        dev.flash_attn_fwd(c, c, c)
    with pytest.raises(ValueError):
        a = torch.zeros((2, 16, 64), device="cuda")
        dev.flash_attn_fwd(a, a[:, :8], a)
