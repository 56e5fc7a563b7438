"""CPU-side checks of the drop-in boundary: every symbol include/flash_attn_mi355x.h declares is exported by
the built libraries, the six reference library names exist, argument validation works without a GPU, and the
host-array operator surface enforces the reference's preconditions.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from flash_attention_minitorch_amd import _lib
    return _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "flash_attn_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:launch_flash_attn|fa_mi355x)_\w+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for s in ("launch_flash_attn_fw", "launch_flash_attn_bw", "fa_mi355x_fwd", "fa_mi355x_bwd",
              "fa_mi355x_bwd_workspace_bytes", "fa_mi355x_launch_fw_host", "fa_mi355x_launch_bw_host",
              "fa_mi355x_bwd_stages", "fa_mi355x_fwd_layout", "fa_mi355x_bwd_layout", "fa_mi355x_fwd_masked", "fa_mi355x_bwd_masked", "fa_mi355x_fwd_dropout", "fa_mi355x_bwd_dropout", "fa_mi355x_last_error", "fa_mi355x_version", "fa_mi355x_set_tuning", "fa_mi355x_debug_phase_cycles", "fa_mi355x_measure_mfma_peak", "fa_mi355x_probe"):
        assert s in syms


def test_core_library_exports_every_declared_symbol(built):
    core = built.core()
    for s in _declared_symbols():
        if s.startswith("launch_flash_attn"):
            continue
        assert hasattr(core, s), s


@pytest.mark.parametrize("name", ["flash_attn_fw.so", "flash_attn2_fw.so", "flash_attn_causal_fw.so"])
def test_forward_shims_export_reference_symbol(built, name):
    lib = built.load(name)
    assert hasattr(lib, "launch_flash_attn_fw")      # src/flash_attn_fw.cu:302
    out = subprocess.run(["nm", "-D", "--defined-only", built.lib_path(name)], capture_output=True, text=True).stdout
    assert " T launch_flash_attn_fw" in out and "launch_flash_attn_bw" not in out


@pytest.mark.parametrize("name", ["flash_attn_bw.so", "flash_attn2_bw.so", "flash_attn_causal_bw.so"])
def test_backward_shims_export_reference_symbol(built, name):
    lib = built.load(name)
    assert hasattr(lib, "launch_flash_attn_bw")      # src/flash_attn_bw.cu:277
    out = subprocess.run(["nm", "-D", "--defined-only", built.lib_path(name)], capture_output=True, text=True).stdout
    assert " T launch_flash_attn_bw" in out and "launch_flash_attn_fw" not in out


def test_library_contains_gfx950_code_object(built):
    blob = open(built.lib_path(built.CORE_NAME), "rb").read()
    assert b"gfx950" in blob
    assert b"fwd_kernel" in blob and b"bwd_dkdv_kernel" in blob and b"bwd_dq_kernel" in blob


def test_argument_validation_without_gpu(built):
    core = built.core()
    assert core.fa_mi355x_version().decode().startswith("flash_attn_mi355x")
    assert core.fa_mi355x_bwd_workspace_bytes(64, 4096, 64) == 2 * 64 * 4096 * 4
    assert core.fa_mi355x_bwd_workspace_bytes(0, 4096, 64) == 0
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)
    # bad sizes / null pointers / unsupported d are rejected before any HIP call
    assert core.fa_mi355x_fwd(one, one, one, one, one, one, 0, 16, 64, 0, 2, 0, null) == 1
    assert core.fa_mi355x_fwd(null, one, one, one, one, one, 1, 16, 64, 0, 2, 0, null) == 1
    assert core.fa_mi355x_fwd(one, one, one, one, one, one, 1, 16, 64, 0, 7, 0, null) == 1
    assert core.fa_mi355x_fwd(one, one, one, one, one, one, 1, 16, 64, 0, 2, 9, null) == 1
    assert core.fa_mi355x_fwd(one, one, one, one, one, one, 1, 16, 34, 0, 2, 0, null) == 2
    assert b"32, 64, 128" in core.fa_mi355x_last_error()
    assert core.fa_mi355x_fwd(one, one, one, one, one, null, 1, 16, 64, 0, 1, 0, null) == 1   # FA-1 needs m
    with pytest.raises(built.FlashAttnLibraryError):
        built.check(2)


def test_missing_library_fails_loudly(built, monkeypatch, tmp_path):
    monkeypatch.setattr(built, "KERNEL_DIR", str(tmp_path))
    monkeypatch.setattr(built, "_handles", {})
    with pytest.raises(built.FlashAttnLibraryError, match="no CPU fallback"):
        built.load("flash_attn2_fw.so")


def test_operator_surface_preconditions(built):
    from flash_attention_minitorch_amd import CudaKernelOps
    q = np.zeros((1, 2, 8, 64), np.float32)
    k = np.zeros((1, 2, 9, 64), np.float32)
    with pytest.raises(AssertionError):   # minitorch/cuda_kernel_ops.py:531
        CudaKernelOps.flash_attn2_fw(q, k, q, True)
    with pytest.raises(AssertionError):   # :592-593 l/m shape
        CudaKernelOps.flash_attn2_bw(q, q, q, q, q, np.zeros((1, 2, 7), np.float32), np.zeros((1, 2, 8), np.float32), 1)
    from flash_attention_minitorch_amd.cuda_kernel_ops import _causal_flag
    assert _causal_flag(True) and _causal_flag(np.array([1.0])) and not _causal_flag(0) and not _causal_flag([0.0])


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "flash_attention_minitorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    code = ("import sys; import flash_attention_minitorch_amd, flash_attention_minitorch_amd.device_ops, "
            "flash_attention_minitorch_amd.sharded; assert not any(m == 'oracle' or m.startswith('oracle.') "
            "for m in sys.modules)")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
