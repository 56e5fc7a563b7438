"""The kernel plan (fa_mi355x_plan) is the library's own dispatch code with the launches skipped; bench.py labels its per-kernel
timings and its roofline from it (bench.stage_plan).  Checked here on the GPU box (the launch-size rules read the CU count): the plan
of the three shapes the bench line reports, and that bench.py's stage list is built from exactly those names."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = [
    # (BH, N, d, causal, dtype): forward, whole backward
    ((64, 4096, 64, False, "bf16"), ["fwd_slot_kernel"], ["bwd_dq_slot_kernel", "bwd_dkdv_slot_kernel"]),      # metric shape M
    ((64, 4096, 64, True, "bf16"), ["fwd_slot_kernel"], ["bwd_dq_slot_kernel", "bwd_dkdv_slot_kernel"]),       # M under the causal mask
    ((256, 4096, 128, False, "bf16"), ["fwd_slot_kernel"], ["bwd_dq_kernel", "bwd_dkdv_kernel"]),              # configs[3]
]


@pytest.mark.parametrize("shape,fwd_names,bwd_names", CASES)
def test_plan_matches_what_bench_times(shape, fwd_names, bwd_names):
    import torch
    import bench
    from flash_attention_minitorch_amd import _lib, device_ops
    assert torch.cuda.is_available()
    BH, N, d, causal, dtype = shape
    dt = _lib.FA_DTYPE_BF16 if dtype == "bf16" else _lib.FA_DTYPE_F32
    fold = device_ops.OPTS_FOLDED_SCALE   # the kernels a call runs once the operands are vouched for (option 8 = 1) or inside the guard's budget
    assert _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, 0, fold) == fwd_names
    assert _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, device_ops.STAGE_ALL, fold) == bwd_names
    # round 4: WITHOUT evidence about the operands a call scales every score in fp32, as the reference does (the phased kernels) ...
    exact_fwd = _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, 0)
    assert exact_fwd[0] == "fwd_kernel" and exact_fwd == _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, 0, device_ops.OPTS_EXACT_SCALE)
    # ... and a GUARDED call (what device_ops and bench.py issue by default) launches both sides of every pair
    assert bench.guarded_call(BH, N, d, causal, dtype, None)
    guarded = _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, 0, bench.with_scale_mode(None, 3))
    assert guarded == fwd_names + exact_fwd
    gb = _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, device_ops.STAGE_ALL, bench.with_scale_mode(None, 3))
    # (the backward needs no twins: its d = 64 slot kernels carry both scalings in one launch, the d = 128 kernels scale in fp32)
    assert gb == bwd_names
    stages, k_fwd, k_dq, k_dkdv = bench.stage_plan(device_ops, BH, N, d, causal, dtype, None, lambda: None, lambda s=7: None, True)
    # no guard pass (the forward's launch fills the guard) and no preprocess kernel (the dQ launch does it and runs before dK/dV)
    assert [n for n, _ in stages] == fwd_names + bwd_names
    assert (k_fwd, k_dq, k_dkdv) == (fwd_names[0], bwd_names[0], bwd_names[1])
    # stage-split calls name the same kernels (plus the preprocess kernel when dQ is not in the call)
    assert _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, device_ops.STAGE_DKDV, fold) == [bwd_names[1]]
    assert _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, device_ops.STAGE_PREP, fold) == ["bwd_prep_kernel"]


def test_plan_of_option_and_feature_paths():
    from flash_attention_minitorch_amd import _lib, device_ops
    bf, fa2 = _lib.FA_DTYPE_BF16, _lib.FA_VARIANT_FA2
    # the phased kernels (fp32 scaling) on request
    assert _lib.plan(64, 4096, 64, False, fa2, bf, 0, device_ops.OPTS_EXACT_SCALE) == ["fwd_kernel"]
    assert _lib.plan(64, 4096, 64, False, fa2, bf, 7, device_ops.OPTS_EXACT_SCALE) == ["bwd_dq_slot_kernel", "bwd_dkdv_slot_kernel"]   # (their fp32-scaling sweep)
    assert _lib.plan(64, 4096, 64, False, fa2, bf, 7, device_ops.OPTS_PHASED) == ["bwd_dq_kernel", "bwd_dkdv_kernel"]
    # a ragged causal launch: phased kernels, each followed by its split-operand launch for the rows with few keys
    assert _lib.plan(2, 200, 128, True, fa2, bf, 0) == ["fwd_kernel", "fwd_kernel"]
    # fp32 (the reference's own dtype)
    f32 = _lib.FA_DTYPE_F32
    assert _lib.plan(64, 2048, 64, True, _lib.FA_VARIANT_FA1, f32, 7) == ["bwd_prep_kernel", "bwd_onepass_f32_kernel"]   # d = 64, N >= 256
    assert _lib.plan(64, 2000, 64, False, _lib.FA_VARIANT_FA2, f32, 7) == ["bwd_prep_kernel", "bwd_onepass_f32_kernel"]
    assert _lib.plan(64, 2048, 64, True, _lib.FA_VARIANT_FA1, f32, 7, (0, 0, 0, 0, 4)) == ["bwd_prep_kernel", "bwd_dkdv_kernel", "bwd_dq_kernel"]
    assert _lib.plan(64, 200, 64, True, _lib.FA_VARIANT_FA1, f32, 7) == ["bwd_dq_kernel", "bwd_dkdv_kernel"]
    assert _lib.plan(64, 2048, 32, False, _lib.FA_VARIANT_FA1, f32, 7) == ["bwd_dq_kernel", "bwd_dkdv_kernel"]
    assert _lib.plan(2, 2048, 64, False, _lib.FA_VARIANT_FA1, f32, 7) == ["bwd_dq_kernel", "bwd_dkdv_kernel"]   # 16 key blocks: even cut 8 ways, half the chip
    assert _lib.plan(2, 2048, 64, False, _lib.FA_VARIANT_FA1, f32, 7, (0, 0, 0, 0, 5)) == ["bwd_prep_kernel", "bwd_onepass_f32_kernel"]
    assert _lib.plan(16, 2048, 64, False, _lib.FA_VARIANT_FA1, f32, 7) == ["bwd_prep_kernel", "bwd_onepass_f32_kernel"]   # 128 blocks x 2 parts
