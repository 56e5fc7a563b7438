"""ctypes loader for the HIP libraries built by ``compile_cuda.sh``.

Mirrors the six ``ctypes.CDLL`` handles of the reference (``minitorch/cuda_kernel_ops.py:30-35``), but with
paths resolved relative to this package instead of the current directory, and lazily, so importing the
package on a machine without the build (or without a GPU) does not fail until an op is called.
"""
from __future__ import annotations

import ctypes
import os

try:  # torch first: its bundled HIP runtime must be the one (and only) libamdhip64 in the process,
    import torch  # noqa: F401  so torch stream handles / device pointers are valid in our launches.
except Exception:  # pragma: no cover - torch-less host-array use
    torch = None

KERNEL_DIR = os.environ.get(
    "FA_MI355X_KERNEL_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), "cuda_kernels")
)

# FA_MI355X_DIAG=1 (tools/ only) loads the diagnostic build instead: stamp / ablation kernels and fa_mi355x_set_tuning live there,
# never in the product library the tests, the shims and bench.py use.
DIAG = bool(os.environ.get("FA_MI355X_DIAG"))
CORE_NAME = "libflash_attn_mi355x_diag.so" if DIAG else "libflash_attn_mi355x.so"
VARIANT_LIBS = (
    "flash_attn_fw.so",
    "flash_attn_bw.so",
    "flash_attn2_fw.so",
    "flash_attn2_bw.so",
    "flash_attn_causal_fw.so",
    "flash_attn_causal_bw.so",
)

FA_VARIANT_FA1 = 1
FA_VARIANT_FA2 = 2
FA_DTYPE_F32 = 0
FA_DTYPE_BF16 = 1
FA_OK = 0
FA_LAYOUT_BHND = 0
FA_LAYOUT_BNHD = 1

_handles: dict = {}


class FlashAttnLibraryError(RuntimeError):
    pass


def lib_path(name: str) -> str:
    return os.path.join(KERNEL_DIR, name)


def load(name: str) -> ctypes.CDLL:
    """dlopen one of the built libraries; raises loudly when it is missing (no fallback path exists)."""
    h = _handles.get(name)
    if h is not None:
        return h
    path = lib_path(name)
    if not os.path.exists(path):
        raise FlashAttnLibraryError(
            f"{path} not found: build the HIP libraries first (./compile_cuda.sh, or "
            f"python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback."
        )
    try:
        h = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL if name == CORE_NAME else ctypes.DEFAULT_MODE)
    except OSError as e:  # e.g. libamdhip64.so missing
        raise FlashAttnLibraryError(f"could not load {path}: {e}") from e
    _handles[name] = h
    return h


_vp, _i, _fp = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)


def core() -> ctypes.CDLL:
    """libflash_attn_mi355x.so with argtypes set for the device-pointer entry points."""
    h = load(CORE_NAME)
    if getattr(h, "_fa_typed", False):
        return h
    h.fa_mi355x_fwd.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
    h.fa_mi355x_fwd.restype = _i
    h.fa_mi355x_bwd.argtypes = [_vp] * 11 + [_i] * 6 + [_vp]
    h.fa_mi355x_bwd.restype = _i
    h.fa_mi355x_bwd_stages.argtypes = [_vp] * 11 + [_i] * 7 + [_vp]
    h.fa_mi355x_bwd_stages.restype = _i
    h.fa_mi355x_fwd_layout.argtypes = [_vp] * 6 + [_i] * 8 + [_vp]
    h.fa_mi355x_fwd_layout.restype = _i
    h.fa_mi355x_bwd_layout.argtypes = [_vp] * 11 + [_i] * 8 + [_vp]
    h.fa_mi355x_bwd_layout.restype = _i
    h.fa_mi355x_fwd_masked.argtypes = [_vp] * 7 + [_i] * 8 + [_vp]
    h.fa_mi355x_fwd_masked.restype = _i
    h.fa_mi355x_bwd_masked.argtypes = [_vp] * 12 + [_i] * 8 + [_vp]
    h.fa_mi355x_bwd_masked.restype = _i
    _f, _u = ctypes.c_float, ctypes.c_uint
    h.fa_mi355x_fwd_dropout.argtypes = [_vp] * 7 + [_f, _f, _u] + [_i] * 8 + [_vp]
    h.fa_mi355x_fwd_dropout.restype = _i
    h.fa_mi355x_bwd_dropout.argtypes = [_vp] * 11 + [_f, _f, _u] + [_vp] + [_i] * 8 + [_vp]
    h.fa_mi355x_bwd_dropout.restype = _i
    h.fa_mi355x_bwd_workspace_bytes.argtypes = [_i, _i, _i]
    h.fa_mi355x_bwd_workspace_bytes.restype = ctypes.c_size_t
    h.fa_mi355x_bwd_workspace_bytes_ex.argtypes = [_i, _i, _i, ctypes.POINTER(ctypes.c_int), _i]
    h.fa_mi355x_bwd_workspace_bytes_ex.restype = ctypes.c_size_t
    h.fa_mi355x_last_error.argtypes = []
    h.fa_mi355x_last_error.restype = ctypes.c_char_p
    h.fa_mi355x_version.argtypes = []
    h.fa_mi355x_version.restype = ctypes.c_char_p
    _ip = ctypes.POINTER(ctypes.c_int)
    h.fa_mi355x_fwd_ex.argtypes = [_vp] * 6 + [_i] * 6 + [_ip, _i, _vp]
    h.fa_mi355x_fwd_ex.restype = _i
    h.fa_mi355x_bwd_ex.argtypes = [_vp] * 11 + [_i] * 7 + [_ip, _i, _vp]
    h.fa_mi355x_bwd_ex.restype = _i
    h.fa_mi355x_bwd_status.argtypes = [_vp, _i, _i, _i, _ip]
    h.fa_mi355x_bwd_status.restype = _i
    h.fa_mi355x_fwd_scaled.argtypes = [_vp] * 6 + [_i] * 5 + [ctypes.c_float] + [_i] * 3 + [_vp]
    h.fa_mi355x_fwd_scaled.restype = _i
    h.fa_mi355x_bwd_scaled.argtypes = [_vp] * 11 + [_i] * 5 + [ctypes.c_float] + [_i] * 3 + [_vp]
    h.fa_mi355x_bwd_scaled.restype = _i
    h.fa_mi355x_fwd_padded.argtypes = [_vp] * 6 + [_i] * 7 + [_vp]
    h.fa_mi355x_fwd_padded.restype = _i
    h.fa_mi355x_bwd_padded.argtypes = [_vp] * 11 + [_i] * 7 + [_vp]
    h.fa_mi355x_bwd_padded.restype = _i
    h.fa_mi355x_guard_bytes.argtypes = []
    h.fa_mi355x_guard_bytes.restype = ctypes.c_size_t
    h.fa_mi355x_scale_guard.argtypes = [_vp, _vp, ctypes.c_long, _i, _i, _vp, _vp]
    h.fa_mi355x_scale_guard.restype = _i
    h.fa_mi355x_fwd_guarded.argtypes = [_vp] * 6 + [_i] * 5 + [ctypes.c_float] + [_i] * 3 + [_ip, _i, _vp, _i, _vp]
    h.fa_mi355x_fwd_guarded.restype = _i
    h.fa_mi355x_bwd_guarded.argtypes = [_vp] * 11 + [_i] * 5 + [ctypes.c_float] + [_i] * 4 + [_ip, _i, _vp, _vp]
    h.fa_mi355x_bwd_guarded.restype = _i
    h.fa_mi355x_plan.argtypes = [_i] * 7 + [_ip, _i, ctypes.c_char_p, ctypes.c_size_t]
    h.fa_mi355x_plan.restype = _i
    if DIAG:
        h.fa_mi355x_set_tuning.argtypes = [_i, _i]
        h.fa_mi355x_set_tuning.restype = _i
        h.fa_mi355x_debug_phase_cycles.argtypes = [_vp, _i]
        h.fa_mi355x_debug_phase_cycles.restype = _i
    h.fa_mi355x_measure_mfma_peak.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _vp]
    h.fa_mi355x_measure_mfma_peak.restype = _i
    h.fa_mi355x_probe.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]
    h.fa_mi355x_probe.restype = _i
    h._fa_typed = True
    return h


def opts_array(opts):
    """ctypes (pointer, count) of a per-call option list for the *_ex entry points (None: defaults)."""
    if not opts:
        return None, 0
    arr = (ctypes.c_int * len(opts))(*[int(x) for x in opts])
    return arr, len(opts)


def plan(batch, n, d, causal, variant, dtype, stages, opts=None):
    """Kernel names, in launch order, of the call with these arguments (fa_mi355x_plan: the library's own dispatch code with the
    launches skipped).  stages = 0: the forward; otherwise the backward stage mask."""
    arr, cnt = opts_array(opts)
    buf = ctypes.create_string_buffer(1024)
    check(core().fa_mi355x_plan(int(batch), int(n), int(d), int(bool(causal)), int(variant), int(dtype), int(stages), arr, cnt, buf, 1024))
    return [x for x in buf.value.decode().split(";") if x]


def check(status: int) -> None:
    if status != FA_OK:
        msg = core().fa_mi355x_last_error().decode()
        raise FlashAttnLibraryError(f"flash_attn_mi355x error {status}: {msg}")
