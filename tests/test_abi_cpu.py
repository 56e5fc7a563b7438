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


def _declared_symbols(diag=False):
    """Entry points the header declares; those inside ``#ifdef FA_DIAG`` belong to the diagnostic build only."""
    text = open(os.path.join(ROOT, "include", "flash_attn_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    diag_blocks = re.findall(r"#ifdef FA_DIAG(.*?)#endif", text, flags=re.S)
    if diag:
        text = "\n".join(diag_blocks)
    else:
        text = re.sub(r"#ifdef FA_DIAG.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:launch_flash_attn|fa_mi355x)_\w+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for s in ("launch_flash_attn_fw", "launch_flash_attn_bw", "fa_mi355x_fwd", "fa_mi355x_bwd",
              "fa_mi355x_bwd_workspace_bytes", "fa_mi355x_launch_fw_host", "fa_mi355x_launch_bw_host",
              "fa_mi355x_bwd_stages", "fa_mi355x_fwd_layout", "fa_mi355x_bwd_layout", "fa_mi355x_fwd_masked", "fa_mi355x_bwd_masked", "fa_mi355x_fwd_dropout", "fa_mi355x_bwd_dropout", "fa_mi355x_last_error", "fa_mi355x_version", "fa_mi355x_fwd_ex", "fa_mi355x_bwd_ex", "fa_mi355x_bwd_status", "fa_mi355x_measure_mfma_peak", "fa_mi355x_probe", "fa_mi355x_plan", "fa_mi355x_fwd_padded", "fa_mi355x_bwd_padded", "fa_mi355x_fwd_scaled", "fa_mi355x_bwd_scaled"):
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


def _device_kernels(path):
    """(name, private_segment_fixed_size, vgpr_count) of every kernel in the gfx950 code object bundled in a built library."""
    import struct
    import tempfile
    objcopy, readelf = "/opt/rocm/lib/llvm/bin/llvm-objcopy", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + fat, path, os.path.join(td, "x")], check=True)
        b = open(fat, "rb").read()
        assert b.startswith(b"__CLANG_OFFLOAD_BUNDLE__")
        n = struct.unpack_from("<Q", b, 24)[0]
        off, co = 32, None
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", b, off)
            off += 24
            triple = b[off:off + tl].decode()
            off += tl
            if "gfx950" in triple:
                co = os.path.join(td, "dev.co")
                open(co, "wb").write(b[o:o + sz])
        assert co, "no gfx950 code object in " + path
        notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        out.append((name, scratch, vgpr))
    return out


def test_product_library_has_no_diagnostic_code_and_no_scratch(built):
    """stamp builds, the barrier-less ablation and fa_mi355x_set_tuning live in the FA_DIAG build only, and no
    kernel of the product library uses scratch memory (round 3: no exception left; the one-pass backward, which kept 8 B per lane,
    moved to the diagnostic build)."""
    path = built.lib_path("libflash_attn_mi355x.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    for s in _declared_symbols(diag=True):
        assert s not in syms, s
    assert "g_phase_cycles" not in syms
    kernels = _device_kernels(path)
    assert len(kernels) > 40
    for name, scratch, vgpr in kernels:
        assert vgpr <= 512
        assert scratch == 0, (name, scratch)
    # template arguments that only diagnostic instantiations carry: DIAG = 1 / 2 of the slot kernels, MODE 9 / 13 / 93 of dK/dV, ABL != 0
    names = " ".join(k[0] for k in kernels)
    assert not re.search(r"bwd_dkdv_kernelI\S*Li(9|13|93)ELb", names)
    assert "bwd_dkdv_slot_kernelIDF16bLi64ELi1E" not in names and "bwd_dq_slot_kernelIDF16bLi64ELi1E" not in names
    assert "bwd_dq_slot_kernelIDF16bLi64ELi2E" not in names
    assert "bwd_fused_kernel" not in names   # the one-pass backward: diagnostic build only since round 3
    assert len(kernels) < 98, len(kernels)   # round 3 library diet (133 before, 88 then; round 4: + two scale-guard kernels, + four builds of the fp32 one-pass backward)


def test_diagnostic_library_exports_the_diag_entry_points(built):
    path = built.lib_path("libflash_attn_mi355x_diag.so")
    assert os.path.exists(path)
    syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    for s in _declared_symbols(diag=True) + ["fa_mi355x_fwd_ex", "fa_mi355x_bwd_ex"]:
        assert s in syms, s


def test_library_contains_gfx950_code_object(built):
    blob = open(built.lib_path(built.CORE_NAME), "rb").read()
    assert b"gfx950" in blob
    assert b"fwd_kernel" in blob and b"bwd_dkdv_kernel" in blob and b"bwd_dq_kernel" in blob


def test_argument_validation_without_gpu(built):
    core = built.core()
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)
    assert core.fa_mi355x_version().decode().startswith("flash_attn_mi355x")
    # three row-constant vectors: -L/tau, -rowsum(dO*O), -L*log2(e)
    assert core.fa_mi355x_bwd_workspace_bytes(64, 4096, 128) == 3 * 64 * 4096 * 4
    assert core.fa_mi355x_bwd_workspace_bytes(64, 4100, 64) == 3 * 64 * 4100 * 4      # not a one-pass shape
    assert core.fa_mi355x_bwd_workspace_bytes(64, 4096, 64) == 3 * 64 * 4096 * 4
    assert core.fa_mi355x_bwd_workspace_bytes(0, 4096, 64) == 0
    # with options: the product library has no kernel that needs more (the chained one-pass backward, opts[4] = 3, whose slabs this
    # entry point sizes, lives in the diagnostic library)
    core.fa_mi355x_bwd_workspace_bytes_ex.restype = ctypes.c_size_t
    chain = (ctypes.c_int * 5)(0, 0, 0, 0, 3)
    assert core.fa_mi355x_bwd_workspace_bytes_ex(64, 4096, 64, chain, 5) == 3 * 64 * 4096 * 4
    assert core.fa_mi355x_bwd_workspace_bytes_ex(64, 4096, 64, None, 0) == 3 * 64 * 4096 * 4
    # round 4: the scale guard and the guarded entry points validate their arguments before any HIP call as well
    core.fa_mi355x_guard_bytes.restype = ctypes.c_size_t
    assert core.fa_mi355x_guard_bytes() == 2 * 256 * 4
    core.fa_mi355x_scale_guard.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert core.fa_mi355x_scale_guard(null, one, 16, 64, 1, one, null) == 1 and core.fa_mi355x_scale_guard(one, one, 0, 64, 1, one, null) == 1
    assert core.fa_mi355x_scale_guard(one, one, 16, 64, 7, one, null) == 1
    fw = core.fa_mi355x_fwd_guarded
    fw.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_float] + [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                                                                                    ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert fw(one, one, one, one, one, one, 1, 1, 16, 64, 0, 0.0, 0, 2, 1, None, 0, null, 1, null) == 1      # produce_guard without a guard
    assert b"produce_guard" in core.fa_mi355x_last_error()
    assert fw(one, one, one, one, one, one, 1, 1, 16, 64, 0, -1.0, 0, 2, 1, None, 0, null, 0, null) == 1     # negative softmax_scale
    assert fw(one, one, one, one, one, one, 1, 1, 16, 64, 5, 0.0, 0, 2, 1, None, 0, null, 0, null) == 1      # unknown layout
    assert fw(one, one, one, one, one, one, 1, 1, 16, 48, 0, 0.0, 0, 2, 1, None, 0, null, 0, null) == 2      # head dim
    mode9 = (ctypes.c_int * 10)(0, 0, 0, 0, 0, 0, 0, 0, 4, 0)
    assert fw(one, one, one, one, one, one, 1, 1, 16, 64, 0, 0.0, 0, 2, 1, mode9, 10, null, 0, null) == 1    # option 8 = 4 does not exist
    # per-call options: diagnostic values are rejected by the product library before any HIP call
    bad = (ctypes.c_int * 3)(93, 0, 0)
    assert core.fa_mi355x_fwd_ex(one, one, one, one, one, one, 1, 16, 64, 0, 2, 0, bad, 3, null) == 1
    assert b"diagnostic" in core.fa_mi355x_last_error()
    assert core.fa_mi355x_fwd_ex(one, one, one, one, one, one, 1, 16, 64, 0, 2, 0, None, 9, null) == 1
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


def test_native_backtrace_hook_fires_on_abort(tmp_path):
    """tests/abort_trace.c (the SIGABRT / SIGSEGV hook of the GPU runs) is itself tested: see gpu_util.check_abort_hook."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import check_abort_hook
    check_abort_hook(tmp_path)


def test_deterministic_switch_keeps_the_two_kernel_fp32_backward(built):
    """FA_MI355X_DETERMINISTIC=1 (read once per process: a child): the fp32 d = 64 backward stays on two kernels (dq bitwise repeatable)
    for callers of the reference ABI, which has no options argument; option 4 = 5 still forces the one-pass kernel.  fa_mi355x_plan
    runs the library's own dispatch without a GPU."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from flash_attention_minitorch_amd import _lib\n"
            "f32 = _lib.FA_DTYPE_F32\n"
            "print(_lib.plan(64, 2048, 64, False, 2, f32, 7, None), _lib.plan(64, 2048, 64, False, 2, f32, 7, (0, 0, 0, 0, 5)))\n") % root
    out = {}
    for val in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, FA_MI355X_DETERMINISTIC=val))
        assert r.returncode == 0, r.stderr[-1000:]
        out[val] = r.stdout.strip()
    assert out["0"] == "['bwd_prep_kernel', 'bwd_onepass_f32_kernel'] ['bwd_prep_kernel', 'bwd_onepass_f32_kernel']", out["0"]
    assert out["1"] == "['bwd_dq_kernel', 'bwd_dkdv_kernel'] ['bwd_prep_kernel', 'bwd_onepass_f32_kernel']", out["1"]
