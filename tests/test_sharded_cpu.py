"""N > 1 path on CPU: world_size-2 (and 3, ragged split) gloo runs of the batch*head shard + all-gather
(flash_attention_minitorch_amd/sharded.py).  The HIP kernels cannot run here, so the per-rank compute function is
injected and backed by the CPU oracle -- only the partitioning / collective logic is under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from flash_attention_minitorch_amd import sharded


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_fwd(q, k, v, causal):
    o, L, _, _ = oracle.dense_attention_fw(q.numpy(), k.numpy(), v.numpy(), causal)
    return torch.from_numpy(o.astype(np.float32)), torch.from_numpy(L.astype(np.float32))


def _oracle_bwd(q, k, v, o, do, L, causal):
    g = oracle.dense_attention_bw(q.numpy(), k.numpy(), v.numpy(), do.numpy(), causal)
    return tuple(torch.from_numpy(a.astype(np.float32)) for a in g)


def _worker(rank, world, port, bh_total, causal, out_dir, N=24, d=16, chunks=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)   # every rank draws the full problem, then keeps its slice
        full = [torch.from_numpy(rng.uniform(-1, 1, (bh_total, N, d)).astype(np.float32)) for _ in range(4)]
        b, e = sharded.shard_range(bh_total, rank, world)
        q, k, v, do = (t[b:e].contiguous() for t in full)
        o, L = sharded.sharded_flash_attn2_fwd(q, k, v, bh_total, causal, compute_fn=_oracle_fwd)
        o_loc, L_loc = sharded.sharded_flash_attn2_fwd(q, k, v, bh_total, causal, gather=False, compute_fn=_oracle_fwd)
        dq, dk, dv = sharded.sharded_flash_attn2_bwd(q, k, v, o_loc, do, L_loc, bh_total, causal,
                                                      compute_fn=_oracle_bwd)
        assert o.shape == (bh_total, N, d) and L.shape == (bh_total, N)
        # gather hidden under the compute of the next piece: same tensors, bit for bit (falls back when the split is ragged)
        o2, L2 = sharded.sharded_flash_attn2_fwd_overlapped(q, k, v, bh_total, causal,
                                                            chunks=chunks or (3 if (e - b) % 3 == 0 else 2), compute_fn=_oracle_fwd)
        assert torch.equal(o2, o) and torch.equal(L2, L)
        g2 = sharded.sharded_flash_attn2_bwd_overlapped(q, k, v, o_loc, do, L_loc, bh_total, causal,
                                                        chunks=chunks or (3 if (e - b) % 3 == 0 else 2), compute_fn=_oracle_bwd)
        assert all(torch.equal(a, b_) for a, b_ in zip(g2, (dq, dk, dv)))
        # round 4: every piece lands by all_gather_into_tensor in a [chunk][rank][cs] staging tensor; as_view returns it permuted to
        # [rank][chunk][cs] (global batch*head order, no copy); a compute_fn with a bf16 O halves the bytes of that gather
        if (e - b) * world == bh_total and (e - b) % 2 == 0:
            ov, Lv = sharded.sharded_flash_attn2_fwd_overlapped(q, k, v, bh_total, causal, chunks=2, compute_fn=_oracle_fwd, as_view=True)
            assert ov.shape[:3] == (world, 2, (e - b) // 2) and torch.equal(ov.reshape(o.shape), o) and torch.equal(Lv.reshape(L.shape), L)
            bf = lambda q_, k_, v_, c_: (lambda oo, LL: (oo.to(torch.bfloat16), LL))(*_oracle_fwd(q_, k_, v_, c_))
            o16, _ = sharded.sharded_flash_attn2_fwd_overlapped(q, k, v, bh_total, causal, chunks=2, compute_fn=bf)
            assert o16.dtype == torch.bfloat16 and torch.equal(o16, o.to(torch.bfloat16))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), o=o.numpy(), L=L.numpy(), dq=dq.numpy(), dk=dk.numpy(),
                 dv=dv.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bh_total,causal", [(2, 6, False), (2, 6, True), (3, 7, False)])
def test_shard_and_gather_matches_single_process(tmp_path, world, bh_total, causal):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, bh_total, causal, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    q, k, v, do = (rng.uniform(-1, 1, (bh_total, 24, 16)).astype(np.float32) for _ in range(4))
    o, L, _, _ = oracle.dense_attention_fw(q, k, v, causal)
    dq, dk, dv = oracle.dense_attention_bw(q, k, v, do, causal)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        for name, ref in (("o", o), ("L", L), ("dq", dq), ("dk", dk), ("dv", dv)):
            assert np.max(np.abs(got[name] - ref)) < 1e-5, (r, name)


def test_configs4_split_eight_ranks_overlapped_gather(tmp_path):
    """BASELINE.json configs[4]'s partition: 2048 (batch, head) pairs over 8 ranks = 256 per rank, the forward's gather hidden under the
    compute in 4 pieces of 64 (what bench.py --config c4 --gpus 8 runs over RCCL), here on gloo with tiny heads (N = 8, d = 8)."""
    world, bh_total, N, d = 8, 2048, 8, 8
    port = _free_port()
    mp.spawn(_worker, args=(world, port, bh_total, False, str(tmp_path), N, d, 4), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    q, k, v, do = (rng.uniform(-1, 1, (bh_total, N, d)).astype(np.float32) for _ in range(4))
    o, L, _, _ = oracle.dense_attention_fw(q, k, v, False)
    dq, dk, dv = oracle.dense_attention_bw(q, k, v, do, False)
    for r in (0, 3, 7):
        got = np.load(tmp_path / f"rank{r}.npz")
        for name, ref in (("o", o), ("L", L), ("dq", dq), ("dk", dk), ("dv", dv)):
            assert got[name].shape == ref.shape and np.max(np.abs(got[name] - ref)) < 1e-5, (r, name)


def test_shard_bounds_are_contiguous_and_balanced():
    for total in (0, 1, 7, 64, 2048):
        for world in (1, 2, 3, 8):
            b = sharded.shard_bounds(total, world)
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1
    assert sharded.shard_range(2048, 3, 8) == (768, 1024)     # BASELINE.json configs[4]: 256 (b,h) per GPU
    with pytest.raises(ValueError):
        sharded.shard_bounds(4, 0)
