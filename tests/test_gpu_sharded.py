"""The N > 1 path with the DEFAULT compute function (the HIP kernels) on one GPU: sharded_flash_attn2_fwd / _bwd /
_fwd_overlapped in a single-process, world-size-1 RCCL group on cuda:0, at the shape one rank of BASELINE.json configs[4]
(B=128, H=16, N=4096, d=128 forward over 8 GPUs: 256 (batch, head) pairs per rank) holds.  The gloo tests
(tests/test_sharded_cpu.py) cover the partitioning with world sizes 2 and 3; this covers what they cannot: the real kernels
behind the collective calls, and the overlapped gather's chunking on the real device."""
import os
import socket

import numpy as np
import pytest

import oracle
from gpu_util import maxabs, oracle_heads, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group():
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _mk(bh, n, d, seed):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(seed)
    return [((torch.rand((bh, n, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16) for _ in range(4)]


def test_sharded_default_compute_small_slice_against_oracle(group):
    """A c4-slice-shaped input with few heads (N = 4096, d = 128, bf16): forward, overlapped forward and backward through
    sharded.py's default compute functions, against the fp64 oracle."""
    import torch
    from flash_attention_minitorch_amd import sharded
    BH, N, d = 4, 4096, 128
    q, k, v, do = _mk(BH, N, d, 11)
    o, L = sharded.sharded_flash_attn2_fwd(q, k, v, BH)
    o2, L2 = sharded.sharded_flash_attn2_fwd_overlapped(q, k, v, BH, chunks=2)
    o_loc, L_loc = sharded.sharded_flash_attn2_fwd(q, k, v, BH, gather=False)
    dq, dk, dv = sharded.sharded_flash_attn2_bwd(q, k, v, o_loc, do, L_loc, BH)
    g2 = sharded.sharded_flash_attn2_bwd_overlapped(q, k, v, o_loc, do, L_loc, BH, chunks=2)
    torch.cuda.synchronize()
    assert torch.equal(o, o2) and torch.equal(L, L2) and torch.equal(o, o_loc)
    assert all(torch.equal(a, b) for a, b in zip((dq, dk, dv), g2))
    heads = [0, 3]
    arrs = [to_np(t.float())[heads] for t in (q, k, v, do)]
    ref = oracle_heads(*arrs, False, range(len(heads)))
    assert maxabs(to_np(o)[heads], ref["o"]) < 1e-3 and maxabs(to_np(L)[heads], ref["L"]) < 1e-3
    for nm, g in (("dq", dq), ("dk", dk), ("dv", dv)):
        assert maxabs(to_np(g)[heads], ref[nm]) < 1e-3, nm


def test_sharded_default_compute_full_c4_rank_slice_properties(group):
    """The full per-rank slice of configs[4] (BH = 256, N = 4096, d = 128, bf16 forward): size-independent properties.
    V = 1 gives O = 1 exactly-ish; the overlapped gather (4 chunks) returns the one-shot gather's tensors bit for bit;
    sampled heads agree with the oracle."""
    import torch
    from flash_attention_minitorch_amd import sharded
    BH, N, d = 256, 4096, 128
    q, k, v, _ = _mk(BH, N, d, 12)
    o, L = sharded.sharded_flash_attn2_fwd(q, k, v, BH)
    o4, L4 = sharded.sharded_flash_attn2_fwd_overlapped(q, k, v, BH, chunks=4)
    torch.cuda.synchronize()
    assert o.shape == (BH, N, d) and L.shape == (BH, N)
    assert torch.equal(o, o4) and torch.equal(L, L4)
    ones = torch.ones_like(v)
    o1, _ = sharded.sharded_flash_attn2_fwd(q, k, ones, BH)
    assert float((o1 - 1.0).abs().max()) < 1e-3
    heads = [5, 250]
    arrs = [to_np(t.float())[heads] for t in (q, k, v)]
    ro, rL, _, _ = oracle.dense_attention_fw(*arrs)
    assert maxabs(to_np(o)[heads], ro) < 1e-3 and maxabs(to_np(L)[heads], rL) < 1e-3


def test_bf16_output_and_staged_gather_view(group):
    """Round 4 (VERDICT r3 missing 4): configs[4] is "FA-2 fw bf16" and SURVEY.md section 8e sizes its gather at 256 MiB per rank: the
    forward can store O as bf16 (option 9: one rounding of the fp32 result, so |bf16 O - fp32 O| <= 2^-9 |O| exactly as torch's own
    rounding gives it), sharded.fwd_bf16_out plugs that into the sharded forwards, and the overlapped gather lands every piece with
    all_gather_into_tensor in a [chunk][rank][cs] staging tensor that as_view hands back without a copy."""
    import torch
    from flash_attention_minitorch_amd import device_ops, sharded
    for d, BH, N in ((128, 8, 1024), (64, 8, 512)):
        q, k, v, _ = _mk(BH, N, d, 21 + d)
        o32, L32 = sharded.sharded_flash_attn2_fwd(q, k, v, BH)
        o16, L16 = sharded.sharded_flash_attn2_fwd(q, k, v, BH, compute_fn=sharded.fwd_bf16_out)
        assert o16.dtype == torch.bfloat16 and o16.shape == o32.shape and torch.equal(L16, L32)
        assert torch.equal(o16, o32.to(torch.bfloat16))          # the kernel's store IS the rounding of its fp32 result
        assert o16.numel() * o16.element_size() * 2 == o32.numel() * o32.element_size()
        for causal in (False, True):                               # the fp32-scaling twins (phased kernels) have the same epilogue
            a, _, _ = device_ops.flash_attn_fwd(q, k, v, causal, opts=device_ops.OPTS_EXACT_SCALE)
            b, _, _ = device_ops.flash_attn_fwd(q, k, v, causal, opts=device_ops.OPTS_EXACT_SCALE, out_dtype=torch.bfloat16)
            assert torch.equal(b, a.to(torch.bfloat16))
        ov, Lv = sharded.sharded_flash_attn2_fwd_overlapped(q, k, v, BH, chunks=4, compute_fn=sharded.fwd_bf16_out, as_view=True)
        torch.cuda.synchronize()
        assert ov.shape == (1, 4, BH // 4, N, d)     # [rank][chunk][cs][N][d]: a permuted view of the [chunk][rank][cs] stage
        assert torch.equal(ov.reshape(BH, N, d), o16) and torch.equal(Lv.reshape(BH, N), L32)
    with pytest.raises(ValueError):
        device_ops.flash_attn_bwd(q, k, v, o16, q, L16)          # the backward needs the fp32 O
