#!/usr/bin/env python3
"""bench.py -- attention fw+bw throughput of the HIP path on MI355X, one JSON line on rank 0.

    python bench.py --gpus 1 --steps 100 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one FlashAttention-2 forward + backward pass (fa_mi355x_fwd + fa_mi355x_bwd, device-pointer C ABI)
over one synthetic batch that is already resident in HBM.  Workload at N=1: the shape BASELINE.json's metric is
quoted on -- B=8, H=8, N=4096, d=64, bf16 inputs (fp32 outputs), non-causal.  With --gpus N every rank runs that
same batch on its own GPU (batch*head shards are independent: weak scaling, no data-path collective in the timed
region); the one RCCL all-gather of O the north star describes is timed separately and reported as
``gather_ms`` (never inside ``value``).

FLOP accounting: fw = 4*B*H*N^2*d, bw = 10*B*H*N^2*d; softmax flops not counted.

    python bench.py --config c4 [--gpus N ...]     BASELINE.json configs[4]: FA-2 forward, bf16, B=128 H=16 N=4096 d=128, the
                                                   batch*head axis cut over the ranks (strong scaling), with the RCCL gather

Extra objects on the JSON line:
  roofline     -- dominant kernel (longest average launch): algorithmic FLOPs per launch / HIP-event duration,
                  against the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline -- the oracle's NumPy fp32 vanilla attention fw+bw (oracle/attention_ref.py), timed on this host on a
                  bounded sample of heads of the same workload (rank 0, --gpus 1 only).
  vs_vanilla   -- the reference's own comparison (README.md:7, kernel_tests/test_flashattn_time.py) on this GPU: materialised-S
                  attention in torch-ROCm (tools/vanilla_gpu.py) beside the flash path, forward and forward+backward.
  variants     -- [total, fw, bw] ms of the reference's three operator variants (FA-1, FA-1 "causal", FA-2; all called with
                  causal_mask = True, fp32, as kernel_tests/test_flashattn_time.py:64-93 times them) and of BASELINE.json
                  configs[1..3] (rank 0, --gpus 1 only; after the timed region).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_F32_TFLOPS = 157.3     # fp32-input MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)   # the clock needs some tens of ms of load to settle
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--seqlen", type=int, default=4096)
    ap.add_argument("--headdim", type=int, default=64)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="BLAS threads of the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--no-kernel-breakdown", action="store_true")
    ap.add_argument("--no-sustained-peak", action="store_true")
    ap.add_argument("--config", choices=["metric", "c4"], default="metric",
                    help="metric: BASELINE.json's headline shape (default); c4: configs[4], forward sharded over the ranks")
    ap.add_argument("--no-extras", action="store_true", help="skip vs_vanilla / variants (rank 0, --gpus 1 only)")
    ap.add_argument("--opts", type=str, default="", help="comma-separated per-call kernel options (fa_mi355x_*_ex), A/B runs only")
    ap.add_argument("--out-bf16", action="store_true",
                    help="--config c4: the forward stores O as bf16 (one rounding of the fp32 result), so the gather moves 256 MiB per rank")
    ap.add_argument("--amp", type=float, default=1.0,
                    help="A/B runs only: q and k are drawn from U(-amp, amp) (the metric's domain is amp = 1; amp >= 1.3 sends the "
                         "guarded default to its fp32-scaling kernels)")
    ap.add_argument("--phased", action="store_true",
                    help="profiling A/B only: run the phased (round-1 v5) kernels instead of the MFMA-slot ones")
    return ap.parse_args()


def with_scale_mode(opts, mode):
    o = list(opts or ()) + [0] * 9
    o[8] = mode
    return tuple(o[:max(9, len(opts or ()))])


def guarded_call(BH, N, d, causal, dtype, opts):
    """Does the default call of this shape run under the scale guard (a kernel with the folded softmax scale AND its fp32-scaling
    twin are launched, include/flash_attn_mi355x.h)?  Asked of the library: the plan of a guarded call names more launches."""
    from flash_attention_minitorch_amd import _lib
    if opts is not None and len(opts) > 8 and opts[8] != 0:
        return False
    dt = _lib.FA_DTYPE_BF16 if dtype == "bf16" else _lib.FA_DTYPE_F32
    one = _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, 0, with_scale_mode(opts, 1))
    two = _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, 0, with_scale_mode(opts, 3))
    return len(two) > len(one)


def stage_plan(device_ops, BH, N, d, causal, dtype, opts, fwd, bwd, guarded=False):
    """The step as a list of (kernel name, callable) in the library's own launch order, from fa_mi355x_plan.  A backward plan without
    bwd_prep_kernel means the dQ launch preprocesses its own rows and runs first; bwd_onepass_f32_kernel is the fp32 d = 64 one-pass
    backward (bwd_fused_kernel the diagnostic library's bf16 one).
    guarded (a step under the scale guard): every stage is the launch of the named kernel plus the launch of its fp32-scaling twin,
    which returns at once for operands inside the guard's budget (the names are those of the chosen side); the forward stage also
    zero-fills the 2-KiB guard that its launch then fills."""
    from flash_attention_minitorch_amd import _lib
    dt = _lib.FA_DTYPE_BF16 if dtype == "bf16" else _lib.FA_DTYPE_F32
    popts = with_scale_mode(opts, 1) if guarded else opts
    plan = lambda stages: _lib.plan(BH, N, d, causal, _lib.FA_VARIANT_FA2, dt, stages, popts)
    main = lambda names: [n for n in names if n != "bwd_prep_kernel"][0]   # (a follow-up launch of the same kernel follows its main one)
    k_fwd = main(plan(0))
    whole = plan(device_ops.STAGE_ALL)
    k_dq, k_dkdv = main(plan(device_ops.STAGE_DQ)), main(plan(device_ops.STAGE_DKDV))
    onepass = [n for n in whole if n in ("bwd_fused_kernel", "bwd_onepass_f32_kernel")]
    if onepass:   # (fp32, d = 64: the one-pass backward is the default; k_dq / k_dkdv name the two-kernel path of a split call)
        stages = ((k_fwd, fwd), ("bwd_prep_kernel", lambda: bwd(device_ops.STAGE_PREP)),
                  (onepass[0], lambda: bwd(device_ops.STAGE_DKDV | device_ops.STAGE_DQ)))
    elif "bwd_prep_kernel" in whole:
        stages = ((k_fwd, fwd), ("bwd_prep_kernel", lambda: bwd(device_ops.STAGE_PREP)),
                  (k_dkdv, lambda: bwd(device_ops.STAGE_DKDV)), (k_dq, lambda: bwd(device_ops.STAGE_DQ)))
    else:
        stages = ((k_fwd, fwd), (k_dq, lambda: bwd(device_ops.STAGE_PREP | device_ops.STAGE_DQ)),
                  (k_dkdv, lambda: bwd(device_ops.STAGE_DKDV)))
    return stages, k_fwd, k_dq, k_dkdv


def run_extras(torch, device_ops, q, k, v, do, causal, B, H):
    """vs_vanilla at the bench shape and the per-variant [total, fw, bw] table; a few seconds, after the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import vanilla_gpu as vg
    from flash_attention_minitorch_amd import _lib
    BH, N, d = q.shape
    out = {}
    # --- flash vs vanilla (bf16 matmuls, fp32 softmax: the strongest vanilla torch offers) at this run's shape
    o, L, _ = device_ops.flash_attn_fwd(q, k, v, causal)
    ws = device_ops.bwd_workspace(q)
    grads = tuple(torch.empty(q.shape, dtype=torch.float32, device="cuda") for _ in range(3))
    # (the default calls run under the scale guard: the forward fills it, the backward takes the same guard)
    g0 = device_ops.new_guard(q)

    def f_fw():
        device_ops.flash_attn_fwd(q, k, v, causal, out=o, l=L, guard=g0, produce_guard=True)

    def f_fwbw():
        f_fw()
        device_ops.flash_attn_bwd(q, k, v, o, do, L, None, causal, workspace=ws, grads=grads, guard=g0)

    v_fw = lambda: vg.vanilla_attention(q, k, v, causal)
    v_fwbw = lambda: vg.vanilla_fw_bw(q, k, v, do, causal)
    r = {"vanilla_fw_ms": vg.time_ms(v_fw, 3), "flash_fw_ms": vg.time_ms(f_fw, 10),
         "vanilla_fwbw_ms": vg.time_ms(v_fwbw, 3), "flash_fwbw_ms": vg.time_ms(f_fwbw, 10)}
    r["max_abs_diff_fw"] = float((v_fw().float() - o).abs().max())
    r["speedup_fw"] = r["vanilla_fw_ms"] / r["flash_fw_ms"]
    r["speedup_fwbw"] = r["vanilla_fwbw_ms"] / r["flash_fwbw_ms"]
    r["vanilla"] = "torch-ROCm materialised S: bf16 matmuls (hipBLASLt), fp32 softmax, autograd backward; device resident"
    # the reference's SECOND comparator (minitorch/modules_transfomer.py:131-136, use_fused_kernel): the same materialised scores, but
    # mask + softmax as ONE fused kernel (src/softmax_kernel.cu:236-282 with its [B, 1, 1, N] padding mask, all zeros there)
    q4, k4, v4, do4 = (t.view(B, H, N, d) for t in (q, k, v, do))
    f_fw2 = lambda: vg.fused_softmax_attention(q4, k4, v4, causal)
    f_fwbw2 = lambda: vg.fused_softmax_fw_bw(q4, k4, v4, do4, causal)
    r["fused_softmax_fw_ms"], r["fused_softmax_fwbw_ms"] = vg.time_ms(f_fw2, 3), vg.time_ms(f_fwbw2, 3)
    r["speedup_fw_vs_fused_softmax"] = r["fused_softmax_fw_ms"] / r["flash_fw_ms"]
    r["speedup_fwbw_vs_fused_softmax"] = r["fused_softmax_fwbw_ms"] / r["flash_fwbw_ms"]
    r["fused_softmax"] = "the same with mask + softmax as one fused kernel (" + vg.fused_softmax_kind() + "): the reference's attn_softmax path"
    out["vs_vanilla"] = {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in r.items()}
    torch.cuda.empty_cache()
    # --- the reference's "breakup" figure (kernel_tests/test_flashattn_breakdown.py:44-66: B=8 H=8 d=64 fp32, causal mask on):
    # per-phase time of the vanilla forward beside the one fused launch, at N = 2048 (the S tensor is 1 GiB in fp32)
    gen = torch.Generator(device="cuda").manual_seed(11)
    mk32 = lambda: (torch.rand((64, 2048, 64), device="cuda", generator=gen) - 0.5) * 2
    q32, k32, v32 = mk32(), mk32(), mk32()
    bd = vg.vanilla_breakdown_ms(q32, k32, v32, True)
    o32, l32, _ = device_ops.flash_attn_fwd(q32, k32, v32, True)
    bd["fused_flash_attn2_fw"] = vg.time_ms(lambda: device_ops.flash_attn_fwd(q32, k32, v32, True, out=o32, l=l32), 10, 3)
    out["vs_vanilla"]["breakdown_ms"] = {"shape": "B8 H8 N2048 d64 fp32 causal (reference timing harness shape)",
                                         **{kk: round(vv, 4) for kk, vv in bd.items()}}
    del q32, k32, v32, o32, l32
    torch.cuda.empty_cache()

    # --- per-variant [total, fw, bw]
    def fw_only(BH_, N_, d_):   # [ms, TFLOP/s, fraction of the 2.5 PFLOP/s dense bf16 peak] of one forward launch
        gen = torch.Generator(device="cuda").manual_seed(9)
        mk = lambda: ((torch.rand((BH_, N_, d_), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
        qq, kk, vv = mk(), mk(), mk()
        oo, ll, _ = device_ops.flash_attn_fwd(qq, kk, vv, False)
        tf = vg.time_ms(lambda: device_ops.flash_attn_fwd(qq, kk, vv, False, out=oo, l=ll), 20, 5)
        tfl = 4.0 * BH_ * N_ * N_ * d_ / tf / 1e9
        return {"ms_tflops_frac": [round(tf, 4), round(tfl, 1), round(tfl / PEAK_BF16_TFLOPS, 4)]}

    def three(B, H, N_, d_, tdt, variant, caus, iters=10, warm=3):
        gen = torch.Generator(device="cuda").manual_seed(7)
        mk = lambda: ((torch.rand((B * H, N_, d_), device="cuda", generator=gen) - 0.5) * 2).to(tdt)
        qq, kk, vv, dd = mk(), mk(), mk(), mk()
        oo, ll, mm = device_ops.flash_attn_fwd(qq, kk, vv, caus, variant)
        w2 = device_ops.bwd_workspace(qq)
        gg = tuple(torch.empty(qq.shape, dtype=torch.float32, device="cuda") for _ in range(3))
        gd = device_ops.new_guard(qq)
        fw = lambda: device_ops.flash_attn_fwd(qq, kk, vv, caus, variant, out=oo, l=ll, m=mm, guard=gd, produce_guard=True)

        bw = lambda: device_ops.flash_attn_bwd(qq, kk, vv, oo, dd, ll, mm, caus, variant, workspace=w2, grads=gg, guard=gd)
        tf, tb = vg.time_ms(fw, iters, warm), vg.time_ms(bw, iters, warm)
        cf = 0.5 if caus else 1.0
        fl_fw, fl_bw = 4.0 * B * H * N_ * N_ * d_ * cf, 10.0 * B * H * N_ * N_ * d_ * cf
        tfl = [(fl_fw + fl_bw) / (tf + tb) / 1e9, fl_fw / tf / 1e9, fl_bw / tb / 1e9]
        peak = PEAK_F32_TFLOPS if tdt == torch.float32 else PEAK_BF16_TFLOPS   # dense MFMA peak of the arithmetic type
        dt_ = _lib.FA_DTYPE_F32 if tdt == torch.float32 else _lib.FA_DTYPE_BF16
        return {"ms_total_fw_bw": [round(tf + tb, 4), round(tf, 4), round(tb, 4)],
                "tflops_total_fw_bw": [round(x, 1) for x in tfl],
                "frac_of_mfma_peak": [round(x / peak, 4) for x in tfl], "mfma_peak_tflops": peak,
                # what the backward call launches (fa_mi355x_plan): fp32 d = 64 launches that fill the chip take the one-pass kernel
                "bw_kernels": _lib.plan(B * H, N_, d_, caus, variant, dt_, device_ops.STAGE_ALL, None)}

    f32, bf = torch.float32, torch.bfloat16
    FA1, FA2 = _lib.FA_VARIANT_FA1, _lib.FA_VARIANT_FA2
    out["variants"] = {
        "what": "[total, fw, bw]; device resident; the first three rows are the reference's timing harness "
                "(kernel_tests/test_flashattn_time.py:64-93: B=8 H=8 d=64, causal_mask=True, fp32) at N=2048",
        "flash_attn (FA-1) fp32 causal B8 H8 N2048 d64": three(8, 8, 2048, 64, f32, FA1, True),
        "flash_attn_causal (FA-1) fp32 causal B8 H8 N2048 d64": three(8, 8, 2048, 64, f32, FA1, True),
        "flash_attn2 (FA-2) fp32 causal B8 H8 N2048 d64": three(8, 8, 2048, 64, f32, FA2, True),
        "configs[1] FA-1 fp32 B8 H8 N1024 d64": three(8, 8, 1024, 64, f32, FA1, False),
        "configs[2] FA-1 fp32 B8 H8 N2048 d64": three(8, 8, 2048, 64, f32, FA1, False),
        "configs[3] FA-2 bf16 B16 H16 N4096 d128": three(16, 16, 4096, 128, bf, FA2, False),
        "metric shape causal FA-2 bf16 B8 H8 N4096 d64": three(8, 8, 4096, 64, bf, FA2, True),
        "configs[4] per-rank slice: FA-2 fw bf16 BH=256 N=4096 d=128": fw_only(256, 4096, 128),
    }
    torch.cuda.empty_cache()
    # --- the reference's ablation sweep (README.md:12-13, kernel_tests/test_flashattn_time.py:96-102: batch, heads, head dim; "batch
    # and head are equivalent"): FA-2 fw+bw TFLOP/s, bf16, N = 2048, non-causal, device resident
    abl = {"what": "FA-2 fw+bw TFLOP/s (14*B*H*N^2*d flops), bf16 in / fp32 out, N=2048, non-causal; key = B<batch>H<heads>d<head dim>"}
    for dd in (32, 64, 128):
        for BB in (4, 8, 16):
            for HH in (4, 8, 16):
                r3 = three(BB, HH, 2048, dd, bf, FA2, False, iters=4, warm=2)
                abl[f"B{BB}H{HH}d{dd}"] = r3["tflops_total_fw_bw"][0]
    out["ablation"] = abl
    return out


def main_c4(args):
    """BASELINE.json configs[4]: FlashAttention-2 forward, bf16, B=128 H=16 N=4096 d=128, batch*head sharded over the ranks
    (2048 pairs / world each: strong scaling), one RCCL all-gather of O; also the gather hidden under the compute."""
    import torch
    import torch.distributed as dist
    from flash_attention_minitorch_amd import device_ops, sharded
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("FA_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    backend = os.environ.get("FA_BENCH_BACKEND", "nccl")
    if backend == "nccl":   # (a one-rank group at --gpus 1: the same collective calls run everywhere)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    B, H, N, d = 128, 16, 4096, 128
    BH_total = B * H
    b0, b1 = sharded.shard_range(BH_total, rank, world)
    bh = b1 - b0
    gen = torch.Generator(device="cuda").manual_seed(2004 + rank)
    mk = lambda: ((torch.rand((bh, N, d), device="cuda", generator=gen) - 0.5) * 2).to(torch.bfloat16)
    q, k, v = mk(), mk(), mk()
    odt = torch.bfloat16 if args.out_bf16 else torch.float32
    out = torch.empty((bh, N, d), dtype=odt, device="cuda")
    L = torch.empty((bh, N), dtype=torch.float32, device="cuda")
    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / steps

    steps, warmup = args.steps, args.warmup
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
    it = [0]

    def fwd():   # (the default call: under the scale guard, formed inside the forward's own launch)
        device_ops.flash_attn_fwd(q, k, v, False, out=out, l=L, out_dtype=odt)

    def fwd_ev():
        e = ev[it[0] % steps]
        e[0].record()
        fwd()
        e[1].record()
        it[0] += 1

    for _ in range(warmup):
        fwd()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fwd_ev()
    barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    sec = float(tt.item()) / steps
    kern_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
    gsteps = max(2, min(5, steps))
    gather_s = timed(lambda: sharded.all_gather_bh(out, BH_total), gsteps, 1)
    chunks = 4 if bh % 4 == 0 else 1
    cfn = sharded.fwd_bf16_out if args.out_bf16 else None
    over_s = timed(lambda: sharded.sharded_flash_attn2_fwd_overlapped(q, k, v, BH_total, False, chunks=chunks, compute_fn=cfn, as_view=True),
                   gsteps, 1)
    flops_total = 4.0 * BH_total * N * N * d
    flops_rank = 4.0 * bh * N * N * d
    if rank == 0:
        ach = flops_rank / (kern_ms * 1e-3) / 1e12
        line = {
            "metric": "attn fw TFLOP/s at configs[4] (B=128,H=16,N=4096,d=128, batch*head sharded); % MFMA roofline",
            "value": round(flops_total / sec / 1e12, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(sec * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[4]: FlashAttention-2 forward, B={B} H={H} N={N} d={d}, bf16 in / "
                                   f"{'bf16' if args.out_bf16 else 'fp32'} out, "
                                   f"non-causal, {BH_total} (batch, head) pairs cut into {world} contiguous slices ({bh} on rank 0)",
                       "B": B, "H": H, "N": N, "d": d, "causal": False,
                       "parallelism": f"batch*head shard x{world}, one RCCL all-gather of O (timed beside the compute)"},
            "pct_mfma_roofline": round(100.0 * flops_total / sec / 1e12 / world / PEAK_BF16_TFLOPS, 2),
            "roofline": {"bound": "mfma", "kernel": "fwd_slot_kernel<bf16,128>", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                         "avg_launch_ms": round(kern_ms, 4), "flops_per_launch": flops_rank},
            "cpu_baseline": None,
            "gather_ms": round(gather_s * 1e3, 3),
            "gather_bytes_per_rank": int(out.numel() * out.element_size()),
            "out_dtype": "bf16" if args.out_bf16 else "fp32",
            "fw_with_gather_overlapped_ms": round(over_s * 1e3, 3),
            "fw_plus_gather_serial_ms": round((sec + gather_s) * 1e3, 3),
            "value_with_gather_overlapped": round(flops_total / over_s / 1e12, 2),
        }
        print(json.dumps(line), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if args.config == "c4":
        return main_c4(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # FA_BENCH_ONE_DEVICE / FA_BENCH_BACKEND: rehearsal of the N > 1 code path on a one-GPU box (every rank on cuda:0, gloo);
    # the driver's runs use neither
    if os.environ.get("FA_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from flash_attention_minitorch_amd import device_ops
    OPTS = device_ops.OPTS_PHASED if args.phased else None   # per-call kernel options (no process-wide state)
    if args.opts:
        OPTS = tuple(int(x) for x in args.opts.split(","))

    B, H, N, d = args.batch, args.heads, args.seqlen, args.headdim
    BH = B * H
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    causal = bool(args.causal)
    gen = torch.Generator(device="cuda").manual_seed(1004 + rank)
    mk = lambda: ((torch.rand((BH, N, d), device="cuda", generator=gen) - 0.5) * 2).to(tdt)  # U(-1,1), test_utils.py:104
    q, k, v, do = mk(), mk(), mk(), mk()
    if args.amp != 1.0:
        q, k = (q.float() * args.amp).to(tdt), (k.float() * args.amp).to(tdt)
    out = torch.empty((BH, N, d), dtype=torch.float32, device="cuda")
    L = torch.empty((BH, N), dtype=torch.float32, device="cuda")
    grads = tuple(torch.empty((BH, N, d), dtype=torch.float32, device="cuda") for _ in range(3))
    ws = device_ops.bwd_workspace(q)

    # The default call runs under the scale guard (include/flash_attn_mi355x.h): every step's forward forms the row norms of its q, k
    # inside its own launch (inside the timed region: a training step sees new q, k every time), its fp32-scaling twin and the
    # backward's launches read the result on the device.
    GUARDED = guarded_call(BH, N, d, causal, args.dtype, OPTS)
    guard = device_ops.new_guard(q, OPTS) if GUARDED else None

    def fwd():   # (fills the guard inside its own launch; the backward's launches read it)
        device_ops.flash_attn_fwd(q, k, v, causal, out=out, l=L, opts=OPTS, guard=guard, produce_guard=GUARDED)

    def bwd(stages=device_ops.STAGE_ALL):
        device_ops.flash_attn_bwd(q, k, v, out, do, L, None, causal, workspace=ws, grads=grads, stages=stages, opts=OPTS, guard=guard)

    # One step = forward + backward; the backward's kernels are launched one by one so that a HIP event can be recorded between
    # kernels INSIDE the timed region (same stream, same kernels, same order as fa_mi355x_bwd).  Which kernels those are, and in which
    # order, is asked of the library (fa_mi355x_plan runs its dispatch code with the launches skipped): kernel names as rocprofv3
    # shows them (fa::<name><...>).
    STAGES, K_FWD, K_DQ, K_DKDV = stage_plan(device_ops, BH, N, d, causal, args.dtype, OPTS, fwd, bwd, GUARDED)
    breakdown = not args.no_kernel_breakdown

    def step(ev=None, only=-1):
        # ev: one HIP event per kernel boundary (len(STAGES) + 1 of them); only >= 0: record just the two around stage `only`
        if ev is not None and only <= 0:
            ev[0].record()
        for i, (_, fn) in enumerate(STAGES):
            fn()
            if ev is not None and (only < 0 or i + 1 == only or i == only):
                ev[i + 1].record()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The clock needs some tens of ms of the SAME load to settle (MI355X_MICROARCH.md, DVFS give-back): whatever --warmup says, the
    # timed region is preceded by at least SETTLE_MS of back-to-back steps (untimed, reported as settle_ms), then the W warm-up steps.
    SETTLE_MS = 60.0
    ts = time.perf_counter()
    while True:
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        settle_ms = (time.perf_counter() - ts) * 1e3
        if settle_ms >= SETTLE_MS:
            break
    # Every kernel's launch duration, by HIP events between the launches, over --steps untimed steps of the settled load: the
    # `kernels_ms` key, and which kernel is the dominant one.  An event between two kernels costs the step about 3 us, so the timed
    # region below keeps only the two events around the dominant kernel (roofline.avg_launch_ms is measured THERE, live).
    new_events = lambda: [[torch.cuda.Event(enable_timing=True) for _ in range(len(STAGES) + 1)] for _ in range(args.steps)]
    pre_events = None
    dom_stage = -1
    if breakdown:
        pre_events = new_events()
        for i in range(args.steps):
            step(pre_events[i])
        torch.cuda.synchronize()
        per_stage = [sum(ev[i].elapsed_time(ev[i + 1]) for ev in pre_events) / args.steps for i in range(len(STAGES))]
        dom_stage = max(range(len(STAGES)), key=lambda i: per_stage[i])
    events = new_events() if breakdown else None
    for _ in range(args.warmup):
        step()

    def timed_block(ev_rows):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(ev_rows[i] if ev_rows is not None else None, dom_stage)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed_block(events if breakdown else None)   # the contractual block: EXACTLY --steps steps, barrier + synchronize on both sides
    # ... and five more blocks of the same length, so that the line carries its own spread (`value` stays the first block's)
    more_blocks = [timed_block(None) for _ in range(5)]

    cf = 0.5 if causal else 1.0
    flops_fw = 4.0 * BH * N * N * d * cf
    flops_bw = 10.0 * BH * N * N * d * cf
    ms_per_step = elapsed / args.steps * 1e3
    value = world * (flops_fw + flops_bw) / (elapsed / args.steps) / 1e12
    blocks_ms = [round(e / args.steps * 1e3, 4) for e in [elapsed] + more_blocks]
    med_ms = sorted(blocks_ms)[len(blocks_ms) // 2]

    # ---- per-kernel average launch duration from the HIP events recorded in the timed region -------------------
    kernels = {}
    kernel_rooflines = None
    roofline = None
    if breakdown:
        # algorithmic split of the backward's 10*B*H*N^2*d: dK/dV kernel owns S, dP, dV, dK (4 GEMMs), dQ kernel owns
        # dQ (1 GEMM); the dQ kernel's recomputation of S and dP is not algorithmic work and is not counted.
        alg = {"scale_guard_kernel": 0.0, K_FWD: flops_fw, "bwd_prep_kernel": 0.0, K_DKDV: 8.0 * BH * N * N * d * cf,
               K_DQ: 2.0 * BH * N * N * d * cf, "bwd_fused_kernel": flops_bw, "bwd_onepass_f32_kernel": flops_bw}
        for i, (name, _) in enumerate(STAGES):   # the dominant kernel: from the timed region; the others: the pass before it
            evs = events if i == dom_stage else pre_events
            ms = sum(ev[i].elapsed_time(ev[i + 1]) for ev in evs) / args.steps
            kernels[name] = (ms, alg[name])
        dom = STAGES[dom_stage][0]
        dur_ms, fl = kernels[dom]
        # every kernel against the same roofline, counted (algorithmic) and executed work: the dQ kernel recomputes S^T and dP^T
        # (6 GEMM units executed for the 2 it is credited with); stage times include the launches that return at the guard check
        executed = {K_FWD: flops_fw, K_DKDV: 8.0 * BH * N * N * d * cf, K_DQ: 6.0 * BH * N * N * d * cf, "bwd_fused_kernel": flops_bw,
                    "bwd_onepass_f32_kernel": flops_bw}
        pk = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        kernel_rooflines = {n: {"ms": round(ms_, 4), "algorithmic_tflops": round(fl_ / (ms_ * 1e-3) / 1e12, 1),
                                "frac": round(fl_ / (ms_ * 1e-3) / 1e12 / pk, 4),
                                "executed_tflops": round(executed.get(n, fl_) / (ms_ * 1e-3) / 1e12, 1),
                                "executed_frac": round(executed.get(n, fl_) / (ms_ * 1e-3) / 1e12 / pk, 4)}
                            for n, (ms_, fl_) in kernels.items() if fl_ > 0 and ms_ > 0}
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        achieved = fl / (dur_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        metric_shape = (B, H, N, d, args.dtype, causal) == (8, 8, 4096, 64, "bf16", False)   # what the PMC passes ran
        if metric_shape and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_source": None if traffic is None else "profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                    "this bench command (gfx950 wide-read correction applied), not a counter read in this run",
                    "avg_launch_ms": round(dur_ms, 4), "flops_per_launch": fl}

    # what this device sustains on a bare bf16 MFMA loop with random operands (power-limited clock), measured live
    sustained = None
    if rank == 0 and args.dtype == "bf16" and not args.no_sustained_peak:
        import ctypes
        from flash_attention_minitorch_amd import _lib
        tf, ghz = ctypes.c_double(0.0), ctypes.c_double(0.0)
        _lib.check(_lib.core().fa_mi355x_measure_mfma_peak(300.0, ctypes.byref(tf), ctypes.byref(ghz),
                                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        sustained = {"value": round(tf.value, 1), "unit": "TFLOP/s", "clock_GHz": round(ghz.value, 3),
                     "what": "bare v_mfma_f32_32x32x16_bf16 loop, random operands, two waves per SIMD on every CU, >= 0.3 s"}
        if roofline is not None and tf.value > 0:
            roofline["frac_of_sustained_mfma"] = round(roofline["achieved"] / tf.value, 4)

    # The one all-gather of O the north star describes: reported beside the metric, never inside it.  Every rank takes the same
    # path (no rank-local try/except: a rank that skipped a collective would leave the others in it).
    gather_ms = fw_with_gather_ms = None
    if world > 1:
        from flash_attention_minitorch_amd import sharded
        sharded.all_gather_bh(out, BH * world)
        barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            sharded.all_gather_bh(out, BH * world)
        barrier()
        gather_ms = (time.perf_counter() - t1) / 3 * 1e3
        # gather-inclusive forward: the gather of piece c hidden under the kernels of piece c+1
        sharded.sharded_flash_attn2_fwd_overlapped(q, k, v, BH * world, causal, chunks=4)
        barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            sharded.sharded_flash_attn2_fwd_overlapped(q, k, v, BH * world, causal, chunks=4)
        barrier()
        fw_with_gather_ms = (time.perf_counter() - t1) / 3 * 1e3

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        import oracle
        from threadpoolctl import threadpool_limits
        # a one-GPU box's CPU share is 16 cores: pin the BLAS pool to that and report it as `cores`
        nthreads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        hq, hk, hv, hdo = (t.float().cpu().numpy() for t in (q, k, v, do))
        with threadpool_limits(limits=nthreads):
            oracle.vanilla_attention_fw_bw_f32(hq[:1, :256], hk[:1, :256], hv[:1, :256], hdo[:1, :256], causal)  # warm
            c0 = time.perf_counter()
            nh = 0
            while nh < BH and (nh < 2 or time.perf_counter() - c0 < args.cpu_seconds):
                oracle.vanilla_attention_fw_bw_f32(hq[nh], hk[nh], hv[nh], hdo[nh], causal)
                nh += 1
            ct = time.perf_counter() - c0
        cpu_flops = 14.0 * nh * N * N * d * cf
        # SURVEY.md section 8d also asks for the one-thread time: one head, one BLAS thread
        with threadpool_limits(limits=1):
            c1 = time.perf_counter()
            oracle.vanilla_attention_fw_bw_f32(hq[0], hk[0], hv[0], hdo[0], causal)
            ct1 = time.perf_counter() - c1
        cpu_baseline = {"value": round(cpu_flops / ct / 1e12, 5), "unit": "TFLOP/s", "cores": nthreads,
                        "kind": "port",
                        "sample": f"{nh} of {BH} heads of the same workload (NumPy fp32 materialised-S attention fw+bw, "
                                  f"{nthreads} BLAS threads), {ct:.1f} s",
                        "host_cpu_count": os.cpu_count(),
                        "single_thread": {"value": round(14.0 * N * N * d * cf / ct1 / 1e12, 5), "unit": "TFLOP/s",
                                          "sample": f"1 head, 1 BLAS thread, {ct1:.1f} s"}}

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        extras = run_extras(torch, device_ops, q, k, v, do, causal, B, H)

    if rank == 0:
        line = {
            "metric": "attn fw+bw TFLOP/s at (B=8,H=8,N=4096,d=64); % MFMA roofline",
            "value": round(value, 2),
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"FlashAttention-2 fw+bw, B={B} H={H} N={N} d={d}, {args.dtype} in / fp32 out, "
                                   f"{'causal' if causal else 'non-causal'}, per GPU",
                       "B": B, "H": H, "N": N, "d": d, "causal": causal,
                       "parallelism": f"batch*head shard x{world}" if world > 1 else "single GPU"},
            "pct_mfma_roofline": round(100.0 * value / world / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS), 2),
            "roofline": roofline,
            "sustained_mfma_peak": sustained,
            "cpu_baseline": cpu_baseline,
            "kernels_ms": {n: round(t, 4) for n, (t, _) in kernels.items()},
            "kernel_rooflines": kernel_rooflines,
            "kernels_ms_source": "HIP events: roofline.kernel inside the timed region (two events per step); the others over the "
                                 "same number of untimed steps just before the warm-up (an event per kernel boundary)",
            "kernels_ms_sum_note": "the entries come from two passes (see kernels_ms_source) and the untimed pass carries an event at "
                                   "every kernel boundary (~3 us each): their sum is NOT ms_per_step and may exceed it by a few us",
            "scale_guard": {"guarded": GUARDED,
                            "what": "the forward launch forms the row norms of q, k itself (2-KiB memset in front of it); the forward, dQ "
                                    "and dK/dV stages each launch the named kernel and its fp32-scaling twin, which returns at once "
                                    "for operands inside the guard's budget: all of it inside the stages' times"
                                    if GUARDED else "no kernel of this call folds the softmax scale into an operand"},
            "settle_ms": round(settle_ms, 1),
            "ms_per_step_blocks": blocks_ms,
            "ms_per_step_median": med_ms,
            "value_median": round(world * (flops_fw + flops_bw) / (med_ms * 1e-3) / 1e12, 2),
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "fw_with_gather_ms": None if fw_with_gather_ms is None else round(fw_with_gather_ms, 3),
            "vs_vanilla": extras.get("vs_vanilla"),
            "variants": extras.get("variants"),
            "ablation": extras.get("ablation"),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
