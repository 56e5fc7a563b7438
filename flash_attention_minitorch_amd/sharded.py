"""Batch*head sharding of the attention path across the GPUs of one node.

Every (batch, head) pair is independent in forward and backward (the reference kernels index by
``blockIdx.x = batch*head`` only, ``src/flash_attn_fw.cu:25-35``), so the flattened BH axis is cut into
contiguous slices, one per rank (one process per GPU), and the only communication is ONE all-gather of the
outputs -- never a reduction.  ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI inside a node);
the same code runs on "gloo" for the CPU tests.

The reference has no counterpart (no distributed code of any kind).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(bh_total: int, world: int) -> Sequence[Tuple[int, int]]:
    """Contiguous, balanced [begin, end) slices of the BH axis; the first (bh_total % world) ranks get one extra."""
    if bh_total < 0 or world <= 0:
        raise ValueError("bh_total >= 0 and world > 0 required")
    base, extra = divmod(bh_total, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < extra else 0)
        out.append((b, e))
        b = e
    return out


def shard_range(bh_total: int, rank: int, world: int) -> Tuple[int, int]:
    return shard_bounds(bh_total, world)[rank]


def all_gather_bh(local: torch.Tensor, bh_total: int, group=None) -> torch.Tensor:
    """Gather rank-local [bh_local, ...] slices into the full [bh_total, ...] tensor on every rank (one collective)."""
    world = dist.get_world_size(group)
    bounds = shard_bounds(bh_total, world)
    sizes = [e - b for b, e in bounds]
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError(f"local slice has {local.shape[0]} rows, expected {sizes[dist.get_rank(group)]}")
    local = local.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((bh_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    # ragged split: pad every slice to the largest, gather once, drop the padding
    mx = max(sizes)
    padded = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def _default_fwd(q, k, v, causal):
    from . import device_ops, _lib
    o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal, _lib.FA_VARIANT_FA2)
    return o, L


def fwd_bf16_out(q, k, v, causal):
    """compute_fn for the sharded forwards: the HIP kernels store O as bf16 (one rounding of the fp32 result, 2^-9 relative), so the
    gather moves half the bytes: BASELINE.json configs[4] (FA-2 fw bf16, B=128 H=16 N=4096 d=128 over 8 ranks) gathers 256 MiB per
    rank instead of 512.  L stays fp32.  Inference / activation-consumer use: the backward needs the fp32 O."""
    import torch as _t
    from . import device_ops, _lib
    o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal, _lib.FA_VARIANT_FA2, out_dtype=_t.bfloat16)
    return o, L


def _default_bwd(q, k, v, o, do, L, causal):
    from . import device_ops, _lib
    return device_ops.flash_attn_bwd(q, k, v, o, do, L, None, causal, _lib.FA_VARIANT_FA2)


def sharded_flash_attn2_fwd(q_local, k_local, v_local, bh_total: int, causal: bool = False, gather: bool = True,
                            group=None, compute_fn: Optional[Callable] = None):
    """FA-2 forward on this rank's [bh_local, N, d] slice; returns (O, L) gathered over BH when ``gather``.

    ``compute_fn(q, k, v, causal) -> (o, L)`` defaults to the HIP kernels; the CPU (gloo) tests inject a checker.
    """
    fn = compute_fn or _default_fwd
    o, L = fn(q_local, k_local, v_local, causal)
    if not gather:
        return o, L
    return all_gather_bh(o, bh_total, group), all_gather_bh(L, bh_total, group)


def _staged_gather(pieces_of, bh_total, world, chunks, cs, group, as_view):
    """Shared by the overlapped forward / backward: piece c of every rank is gathered with all_gather_into_tensor straight into
    stage[c] = [rank][cs][...] (a contiguous destination: the collective writes the final bytes itself, no flat temporary and no copy
    out, which the list form of all_gather needs on the NCCL / RCCL backend when its destinations are not rank-contiguous).  The stage
    is [chunk][rank][cs][...]; global batch*head row r * (chunks * cs) + c * cs + i is stage[c][r][i]: returned as that permuted VIEW
    ([rank][chunk][cs][...], zero copies) when ``as_view``, else as one contiguous [bh_total, ...] tensor (one final permute-copy)."""
    stages, pending, keep = None, [], []
    for c in range(chunks):
        pieces = tuple(p.contiguous() for p in pieces_of(c))
        if stages is None:
            stages = tuple(torch.empty((chunks, world, cs) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device) for p in pieces)
        for st, p in zip(stages, pieces):
            pending.append(dist.all_gather_into_tensor(st[c].view((world * cs,) + tuple(p.shape[1:])), p, group=group, async_op=True))
        keep.append(pieces)   # the pieces must outlive their collectives
    for work in pending:
        work.wait()
    views = tuple(st.permute(1, 0, 2, *range(3, st.dim())) for st in stages)
    if as_view:
        return views
    return tuple(v.reshape((bh_total,) + tuple(v.shape[3:])) for v in views)


def sharded_flash_attn2_fwd_overlapped(q_local, k_local, v_local, bh_total: int, causal: bool = False, chunks: int = 4,
                                       group=None, compute_fn: Optional[Callable] = None, as_view: bool = False):
    """The same result as sharded_flash_attn2_fwd(..., gather=True) with the all-gather hidden under the compute
    (at the 8-GPU config the gather of one rank's slice over its xGMI links costs about as much as its
    forward).  The local slice is cut into ``chunks`` pieces along batch*head; the gather of piece c is issued asynchronously
    (RCCL runs it on its own stream, ordered after the kernels already queued) and overlaps the kernels of piece c+1.  Every piece
    is gathered with all_gather_into_tensor into a [chunk][rank][cs] staging tensor (see _staged_gather); ``as_view`` returns
    (O, L) as [rank][chunk][cs][...] views of it (no copy at all), the default one contiguous [bh_total, ...] copy each.
    ``compute_fn=fwd_bf16_out`` halves the bytes of the O gather.  Needs an even split (bh_total divisible by the world size, the
    local slice by ``chunks``); otherwise it falls back to the one-shot gather."""
    fn = compute_fn or _default_fwd
    world = dist.get_world_size(group)
    bounds = shard_bounds(bh_total, world)
    bh_local = q_local.shape[0]
    if len({e - b for b, e in bounds}) != 1 or chunks <= 1 or bh_local % chunks != 0:
        return sharded_flash_attn2_fwd(q_local, k_local, v_local, bh_total, causal, True, group, compute_fn)
    cs = bh_local // chunks

    def pieces_of(c):
        sl = slice(c * cs, (c + 1) * cs)
        return fn(q_local[sl], k_local[sl], v_local[sl], causal)

    return _staged_gather(pieces_of, bh_total, world, chunks, cs, group, as_view)


def sharded_flash_attn2_bwd(q_local, k_local, v_local, o_local, do_local, L_local, bh_total: int,
                            causal: bool = False, gather: bool = True, group=None,
                            compute_fn: Optional[Callable] = None):
    """FA-2 backward on this rank's slice; gradients of different (b, h) never overlap, so gathering them is
    again an all-gather, not an all-reduce."""
    fn = compute_fn or _default_bwd
    dq, dk, dv = fn(q_local, k_local, v_local, o_local, do_local, L_local, causal)
    if not gather:
        return dq, dk, dv
    return tuple(all_gather_bh(g, bh_total, group) for g in (dq, dk, dv))


def sharded_flash_attn2_bwd_overlapped(q_local, k_local, v_local, o_local, do_local, L_local, bh_total: int,
                                       causal: bool = False, chunks: int = 4, group=None,
                                       compute_fn: Optional[Callable] = None, as_view: bool = False):
    """sharded_flash_attn2_bwd(..., gather=True) with the three gradient gathers hidden under the compute, as
    sharded_flash_attn2_fwd_overlapped does for O: the local slice is cut into ``chunks`` pieces along batch*head, the gathers of piece
    c (dQ, dK, dV: all-gathers, never a reduction -- gradients of different (b, h) do not overlap) run while piece c+1 computes, each
    straight into its [chunk][rank][cs] staging tensor (_staged_gather; ``as_view`` as there).  Needs an even split; otherwise it
    falls back to the one-shot gathers."""
    fn = compute_fn or _default_bwd
    world = dist.get_world_size(group)
    bounds = shard_bounds(bh_total, world)
    bh_local = q_local.shape[0]
    if len({e - b for b, e in bounds}) != 1 or chunks <= 1 or bh_local % chunks != 0:
        return sharded_flash_attn2_bwd(q_local, k_local, v_local, o_local, do_local, L_local, bh_total, causal, True, group,
                                       compute_fn)
    cs = bh_local // chunks

    def pieces_of(c):
        sl = slice(c * cs, (c + 1) * cs)
        return fn(q_local[sl], k_local[sl], v_local[sl], o_local[sl], do_local[sl], L_local[sl], causal)

    return _staged_gather(pieces_of, bh_total, world, chunks, cs, group, as_view)
