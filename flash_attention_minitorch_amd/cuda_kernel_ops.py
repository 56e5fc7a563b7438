"""Host-array operator surface: the counterpart of ``CudaKernelOps.flash_attn*`` in the reference
(``minitorch/cuda_kernel_ops.py:527-677``), bound to the HIP libraries through the reference's own FFI
(``launch_flash_attn_fw`` / ``launch_flash_attn_bw``, host fp32 pointers).

minitorch itself is not importable on the GPU box (needs numba + pycuda), so tensors here are NumPy
float32 arrays of shape (B, H, N, d); anything exposing ``to_numpy()`` (a minitorch Tensor) is accepted
too.  ``causal_mask`` is the reference's 1-element tensor read with ``.item()``
(``minitorch/cuda_kernel_ops.py:529``) -- a bool / int / 1-element array works the same way.

Returns follow the reference exactly:
  ``flash_attn*_fw(q, k, v, causal_mask) -> (out (B,H,N,d), l (B,H,N), m (B,H,N))``
  ``flash_attn*_bw(q, k, v, out, out_grad, l, m, causal_mask) -> (q_grad, k_grad, v_grad, causal_mask)``
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

datatype = np.float32  # minitorch/cuda_kernel_ops.py:36


def _as_array(t) -> np.ndarray:
    if hasattr(t, "to_numpy"):
        t = t.to_numpy()
    return np.asarray(t)


def _causal_flag(causal_mask) -> bool:
    # int(causal_mask._tensor._storage.item()) == 1      (minitorch/cuda_kernel_ops.py:529)
    if hasattr(causal_mask, "_tensor"):
        return int(causal_mask._tensor._storage.item()) == 1
    return int(np.asarray(causal_mask).reshape(-1)[0]) == 1


def _stream():
    """torch.cuda.current_stream().cuda_stream when torch sees a GPU (minitorch/cuda_kernel_ops.py:535), else NULL."""
    try:
        import torch

        if torch.cuda.is_available():
            return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    except Exception:
        pass
    return ctypes.c_void_p(0)


_ND = np.ctypeslib.ndpointer(dtype=datatype, ndim=1, flags="C_CONTIGUOUS")
_FW_ARGTYPES = [_ND] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_bool, ctypes.c_void_p]
_BW_ARGTYPES = [_ND] * 10 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_bool, ctypes.c_void_p]


class CudaKernelOps:
    """Name kept from the reference so call sites read the same; the kernels are HIP (gfx950)."""

    @staticmethod
    def flash_attn_fw_generic(q, k, v, causal_mask, generic_lib):
        # minitorch/cuda_kernel_ops.py:527-581
        causal = _causal_flag(causal_mask)
        q, k, v = _as_array(q), _as_array(k), _as_array(v)
        assert q.ndim == 4, "expected (batch, nhead, from_len, to_len)"
        batch_size, nhead, from_len, to_len = q.shape  # to_len IS the head dim d
        assert q.shape == k.shape
        assert q.shape == v.shape
        assert q.strides == k.strides
        assert q.strides == v.strides
        bh = batch_size * nhead
        qf = np.ascontiguousarray(q, dtype=datatype).reshape(-1)
        kf = np.ascontiguousarray(k, dtype=datatype).reshape(-1)
        vf = np.ascontiguousarray(v, dtype=datatype).reshape(-1)
        out = np.zeros(bh * from_len * to_len, dtype=datatype)                       # :537
        l = np.zeros(bh * from_len, dtype=datatype)                                  # :538
        m = np.full(bh * from_len, -np.finfo(datatype).max, dtype=datatype)          # :539
        fn = generic_lib.launch_flash_attn_fw
        fn.argtypes = _FW_ARGTYPES
        fn.restype = None
        fn(qf, kf, vf, out, l, m, bh, from_len, to_len, causal, _stream())
        return (
            out.reshape(batch_size, nhead, from_len, to_len),
            l.reshape(batch_size, nhead, from_len),
            m.reshape(batch_size, nhead, from_len),
        )

    @staticmethod
    def flash_attn_bw_generic(q, k, v, out, out_grad, l, m, causal_mask, generic_lib):
        # minitorch/cuda_kernel_ops.py:583-653
        causal = _causal_flag(causal_mask)
        q, k, v, out, out_grad, l, m = (_as_array(t) for t in (q, k, v, out, out_grad, l, m))
        batch_size, nhead, from_len, to_len = q.shape
        assert q.shape == k.shape
        assert q.shape == v.shape
        assert q.shape == out.shape
        assert q.shape == out_grad.shape
        assert l.shape == (batch_size, nhead, from_len)
        assert m.shape == (batch_size, nhead, from_len)
        assert q.strides == k.strides
        assert q.strides == v.strides
        assert q.strides == out.strides
        assert q.strides == out_grad.strides
        bh = batch_size * nhead
        flat = lambda a: np.ascontiguousarray(a, dtype=datatype).reshape(-1)
        q_grad = np.zeros(bh * from_len * to_len, dtype=datatype)                    # :609-611
        k_grad = np.zeros_like(q_grad)
        v_grad = np.zeros_like(q_grad)
        fn = generic_lib.launch_flash_attn_bw
        fn.argtypes = _BW_ARGTYPES
        fn.restype = None
        fn(flat(q), flat(k), flat(v), flat(out), flat(out_grad), q_grad, k_grad, v_grad, flat(l), flat(m),
           bh, from_len, to_len, causal, _stream())
        shape = (batch_size, nhead, from_len, to_len)
        return q_grad.reshape(shape), k_grad.reshape(shape), v_grad.reshape(shape), causal_mask

    # variant selectors: minitorch/cuda_kernel_ops.py:655-677 -- the LIBRARY picks the variant
    @staticmethod
    def flash_attn_fw(q, k, v, causal_mask):
        return CudaKernelOps.flash_attn_fw_generic(q, k, v, causal_mask, _lib.load("flash_attn_fw.so"))

    @staticmethod
    def flash_attn_bw(q, k, v, out, out_grad, l, m, causal_mask):
        return CudaKernelOps.flash_attn_bw_generic(q, k, v, out, out_grad, l, m, causal_mask,
                                                   _lib.load("flash_attn_bw.so"))

    @staticmethod
    def flash_attn2_fw(q, k, v, causal_mask):
        return CudaKernelOps.flash_attn_fw_generic(q, k, v, causal_mask, _lib.load("flash_attn2_fw.so"))

    @staticmethod
    def flash_attn2_bw(q, k, v, out, out_grad, l, m, causal_mask):
        return CudaKernelOps.flash_attn_bw_generic(q, k, v, out, out_grad, l, m, causal_mask,
                                                   _lib.load("flash_attn2_bw.so"))

    @staticmethod
    def flash_attn_causal_fw(q, k, v, causal_mask):
        return CudaKernelOps.flash_attn_fw_generic(q, k, v, causal_mask, _lib.load("flash_attn_causal_fw.so"))

    @staticmethod
    def flash_attn_causal_bw(q, k, v, out, out_grad, l, m, causal_mask):
        return CudaKernelOps.flash_attn_bw_generic(q, k, v, out, out_grad, l, m, causal_mask,
                                                   _lib.load("flash_attn_causal_bw.so"))
