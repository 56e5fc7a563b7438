"""MI355X-native FlashAttention forward/backward behind minitorch's flash-attn operator surface.

Only what the hot path needs lives here:

* ``csrc/``            hand-written HIP kernels for gfx950 + the C ABI (``include/flash_attn_mi355x.h``)
* ``cuda_kernels/``    build output of ``compile_cuda.sh``: the six library names the reference opens
                       (``minitorch/cuda_kernel_ops.py:30-35``) + ``libflash_attn_mi355x.so``
* ``cuda_kernel_ops``  host-array operator surface, same names / argument meaning as the reference's
                       ``CudaKernelOps.flash_attn*_fw / _bw`` (``minitorch/cuda_kernel_ops.py:527-677``)
* ``device_ops``       device-resident (torch-ROCm tensors) entry points + autograd Functions
* ``modules_transformer``  the in-model caller: MultiHeadAttention data flow on [B][N][H][d], no head-split copies (row f1)
* ``sharded``          batch*head shard across the GPUs of one node (RCCL all-gather)

The product path never imports ``oracle/`` and has no CPU fallback: if the HIP libraries are missing,
loading fails loudly.
"""
from . import _lib  # noqa: F401
from .cuda_kernel_ops import CudaKernelOps  # noqa: F401

__all__ = ["CudaKernelOps", "_lib"]
__version__ = "0.1"
