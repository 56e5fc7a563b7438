"""MFMA operand-layout probes on the real GPU: what the atoms (csrc/fa_atoms.h) read from an LDS tile, checked with
exact small-integer data.  A symmetric operand would hide a transposed map, so every check uses asymmetric data
(cdna_hip_programming.md section 3: 'Always A=I-check with ASYMMETRIC B')."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def acc_row(i, h):
    return (i & 3) + 8 * (i >> 2) + 4 * h


def run_probe(tile, b, d, dtype):
    import torch
    from flash_attention_minitorch_amd import _lib

    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    code = _lib.FA_DTYPE_BF16 if dtype == "bf16" else _lib.FA_DTYPE_F32
    t_tile = torch.from_numpy(tile).to("cuda", tdt).contiguous()
    t_b = torch.from_numpy(b).to("cuda", tdt).contiguous()
    row_out = torch.zeros((d // 16, 64, 8), device="cuda")
    tr_out = torch.zeros((d // 32, 4, 64, 8), device="cuda")
    mma_out = torch.zeros((2, 64, 16), device="cuda")
    swap_out = torch.zeros((2, 64), device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(_lib.core().fa_mi355x_probe(p(t_tile), p(t_b), p(row_out), p(tr_out), p(mma_out), p(swap_out), d, code,
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in (row_out, tr_out, mma_out, swap_out)]


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("d", [32, 64, 128])
def test_row_and_transposed_fragment_maps(dtype, d):
    b = np.zeros((32, d), np.float32)
    for enc in ("row", "col"):
        tile = np.zeros((64, d), np.float32)
        tile[:] = np.arange(64)[:, None] if enc == "row" else np.arange(d)[None, :]
        row_out, tr_out, _, _ = run_probe(tile, b, d, dtype)
        for lane in range(64):
            r, h = lane & 31, lane >> 5
            for kc in range(d // 16):
                exp = [tile[32 + r, 16 * kc + 8 * h + j] for j in range(8)]
                assert list(row_out[kc, lane]) == exp, (enc, "row_frag", lane, kc)
            for ct in range(d // 32):
                for s in range(4):
                    exp = [tile[16 * s + 8 * (j >> 2) + 4 * h + (j & 3), 32 * ct + r] for j in range(8)]
                    assert list(tr_out[ct, s, lane]) == exp, (enc, "tr_frag", lane, ct, s)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("d", [32, 64, 128])
def test_mfma_product_and_accumulator_as_operand(dtype, d):
    rng = np.random.default_rng(d)
    tile = rng.integers(-1, 2, (64, d)).astype(np.float32)
    b = rng.integers(-1, 2, (32, d)).astype(np.float32)
    _, _, mma_out, swap_out = run_probe(tile, b, d, dtype)
    X = tile[:32] @ b.T                       # X[m][n]
    Y = tile[:32, :32].T @ X                  # Y[c][n] = sum_m tile[m][c] X[m][n]
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        for i in range(16):
            assert mma_out[0, lane, i] == X[acc_row(i, h), r], ("X", lane, i)
            assert mma_out[1, lane, i] == Y[acc_row(i, h), r], ("Y", lane, i)
        assert swap_out[0, lane] == (lane & 31) + 32
        assert swap_out[1, lane] == 2 * (lane & 31) + 32
