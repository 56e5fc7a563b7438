"""Helpers shared by the GPU parity tests (imported only under -m gpu)."""
import numpy as np

import oracle


def rand_u(rng, shape):
    # U(-1, 1) like the reference harness, test_utils.py:104-105
    return ((rng.random(shape, dtype=np.float32) - np.float32(0.5)) * np.float32(2)).astype(np.float32)


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def to_np(t):
    return t.detach().float().cpu().numpy()


def oracle_heads(q, k, v, do, causal, heads):
    """fp64 dense oracle on a subset of the flattened (BH, N, d) heads.  Returns dict of stacked arrays."""
    out = {n: [] for n in ("o", "L", "m", "l", "dq", "dk", "dv")}
    for hh in heads:
        o, L, m, l = oracle.dense_attention_fw(q[hh], k[hh], v[hh], causal)
        out["o"].append(o); out["L"].append(L); out["m"].append(m); out["l"].append(l)
        if do is not None:
            dq, dk, dv = oracle.dense_attention_bw(q[hh], k[hh], v[hh], do[hh], causal)
            out["dq"].append(dq); out["dk"].append(dk); out["dv"].append(dv)
    return {n: np.stack(a) for n, a in out.items() if a}
