#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own CPU model.

Runs only in the build container (needs /root/reference).  It imports
``kernel_tests/flash_attn_python.py`` from the reference checkout (torch CPU +
numpy only), feeds it seeded inputs and stores inputs + outputs as ``.npz``.
Nothing of the reference's text is stored: fixtures are data only.

Input distribution follows the reference harness: U(-1, 1) via ``(rand - 0.5) * 2``
(``test_utils.py:104-105``); ``dO = ones`` as ``kernel_tests/test_flashattn_bw.py:32``
plus a random dO.  Shapes: config c0 of BASELINE.json (B=1,H=2,N=128,d=64), the
reference's ragged test shapes N=40,d=32 (``kernel_tests/test_flashattn_comb.py:86-89``)
and N=327,d=34 (``kernel_tests/test_flashattn_2_fw.py:137-138``), and the notebook
shape N=334,d=233 seed 2 fp64 (``notebooks/flash_attention_backward.ipynb`` cell 1; dense only).

Usage:  python tests/golden/make_golden.py [--ref /root/reference]
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference(ref_root):
    sys.dont_write_bytecode = True  # the reference checkout is read-only
    path = os.path.join(ref_root, "kernel_tests", "flash_attn_python.py")
    spec = importlib.util.spec_from_file_location("ref_flash_attn_python", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rand_u(rng, shape):
    # (rand - 0.5) * 2  -> U(-1, 1), test_utils.py:104-105
    return ((rng.random(shape, dtype=np.float32) - np.float32(0.5)) * np.float32(2)).astype(np.float32)


def run_heads(fn, *arrs, nout):
    """Apply a per-head (N, d) reference function over the leading BH axis."""
    outs = [[] for _ in range(nout)]
    for h in range(arrs[0].shape[0]):
        res = fn(*[a[h] for a in arrs])
        if not isinstance(res, tuple):
            res = (res,)
        for o, r in zip(outs, res):
            o.append(r.detach().numpy() if isinstance(r, torch.Tensor) else np.asarray(r))
    return [np.stack(o) for o in outs]


def dense_autograd(ref, Q, K, V, dO):
    """Gradients of the reference's dense compute_attention via torch autograd."""
    dq, dk, dv = [], [], []
    for h in range(Q.shape[0]):
        q = Q[h].clone().requires_grad_(True)
        k = K[h].clone().requires_grad_(True)
        v = V[h].clone().requires_grad_(True)
        o = ref.compute_attention(q, k, v)
        o.backward(dO[h])
        dq.append(q.grad.numpy()); dk.append(k.grad.numpy()); dv.append(v.grad.numpy())
    return np.stack(dq), np.stack(dk), np.stack(dv)


def make_case(ref, name, BH, N, d, seed, tiled=True, dtypes=("f32", "f64"), do_kinds=("rand", "ones"),
              store_f32=False):
    rng = np.random.default_rng(seed)
    q = rand_u(rng, (BH, N, d)); k = rand_u(rng, (BH, N, d)); v = rand_u(rng, (BH, N, d))
    do_rand = rand_u(rng, (BH, N, d))
    do_ones = np.ones((BH, N, d), dtype=np.float32)
    out = dict(q=q, k=k, v=v, do_rand=do_rand, seed=np.int64(seed))
    for tag in dtypes:
        tdt = torch.float32 if tag == "f32" else torch.float64
        Q, K, V = (torch.from_numpy(a).to(tdt) for a in (q, k, v))
        (o_dense,) = run_heads(ref.compute_attention, Q, K, V, nout=1)
        out[f"o_dense_{tag}"] = o_dense
        for dname, do_np in (("rand", do_rand), ("ones", do_ones)):
            if dname not in do_kinds:
                continue
            dO = torch.from_numpy(do_np).to(tdt)
            gq, gk, gv = dense_autograd(ref, Q, K, V, dO)
            out[f"dq_dense_{dname}_{tag}"] = gq
            out[f"dk_dense_{dname}_{tag}"] = gk
            out[f"dv_dense_{dname}_{tag}"] = gv
        if not tiled:
            continue
        o1, l1, m1 = run_heads(ref.flash_attention, Q, K, V, nout=3)
        o2, L2 = run_heads(ref.flash_attention2, Q, K, V, nout=2)
        out[f"fa1_o_{tag}"] = o1; out[f"fa1_l_{tag}"] = l1; out[f"fa1_m_{tag}"] = m1
        out[f"fa2_o_{tag}"] = o2; out[f"fa2_L_{tag}"] = L2
        # l, m, L come back float64; the reference's backward needs them in Q's dtype
        # (kernel_tests/test_flashattn_bw.py:74-75,83 stores them into fp32 buffers first).
        l_t = torch.from_numpy(l1).to(tdt); m_t = torch.from_numpy(m1).to(tdt)
        L_t = torch.from_numpy(L2).to(tdt)
        O1 = torch.from_numpy(o1).to(tdt); O2 = torch.from_numpy(o2).to(tdt)
        dO = torch.from_numpy(do_rand).to(tdt)
        g1 = run_heads(ref.flash_attention_backward, Q, K, V, O1, dO, l_t, m_t, nout=3)
        g2 = run_heads(ref.flash_attention2_backward, Q, K, V, O2, dO, L_t, nout=3)
        for nm, a in zip(("dq", "dk", "dv"), g1):
            out[f"fa1_{nm}_rand_{tag}"] = a
        for nm, a in zip(("dq", "dk", "dv"), g2):
            out[f"fa2_{nm}_rand_{tag}"] = a
    if store_f32:  # outputs computed in fp64 but stored rounded to fp32 (keeps the fixture small)
        out = {kk: (vv.astype(np.float32) if getattr(vv, "dtype", None) == np.float64 else vv) for kk, vv in out.items()}
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    ref = load_reference(args.ref)
    torch.manual_seed(0)
    # c0: BASELINE.json configs[0]  B=1 H=2 N=128 d=64
    make_case(ref, "c0_b1h2n128d64", BH=2, N=128, d=64, seed=1000)
    # ragged N (not a multiple of 16) and small d: kernel_tests/test_flashattn_comb.py:86-89
    make_case(ref, "ragged_n40d32", BH=4, N=40, d=32, seed=1001)
    # odd d: kernel_tests/test_flashattn_2_fw.py:137-138 (one head; 4x4 tiles are slow)
    make_case(ref, "ragged_n327d34", BH=1, N=327, d=34, seed=1002, dtypes=("f64",))
    # notebook shape (d > 128: dense only), fp64
    make_case(ref, "nb_n334d233", BH=1, N=334, d=233, seed=2, tiled=False, dtypes=("f64",), do_kinds=("rand",),
              store_f32=True)


if __name__ == "__main__":
    main()
