#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>_*/ *_results.db, rocpd SQLite) into the small
files committed under profiles/: <tag>_kernel_stats.csv (per-kernel launch durations from the kernel trace),
<tag>_pmc.json (per-launch counter averages of our kernels) and profiles/hbm_traffic.json (bytes per launch, gfx950
FETCH_SIZE correction applied).   usage: tools/summarize_profile.py <tag>"""
import csv, glob, json, os, sqlite3, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
KNAMES = ("scale_guard_kernel", "fwd_slot_kernel", "fwd_kernel", "bwd_prep_kernel", "bwd_onepass_f32_kernel", "bwd_fused_kernel", "bwd_chain_kernel", "bwd_dkdv_slot_kernel", "bwd_dkdv_kernel",
          "bwd_dq_slot_kernel", "bwd_dq_kernel")

def short(name):   # mangled fa:: kernel name -> the name bench.py reports
    if "fa" not in name:
        return None
    for k in KNAMES:
        if k in name:
            return k
    return None

def db(sub):
    f = glob.glob(os.path.join(out, f"{tag}_{sub}", "**", "*_results.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None

# ---- kernel durations from the kernel trace of the bench command
d = db("trace")
steady = {}
if d:
    per = {}
    for name, dur in d.execute("select name, duration from kernels order by start"):
        n = short(name)
        if n:
            per.setdefault((n, name), []).append(dur)
    rows = []
    for (n, full), v in per.items():
        s = v[len(v) // 5:]          # the first fifth are the warm-up steps
        steady[n] = sum(s) / len(s) / 1e6
        rows.append({"Kernel": n, "Name": full, "Calls": len(v), "TotalDurationNs": sum(v), "AverageNs": sum(v) / len(v),
                     "MinNs": min(v), "MaxNs": max(v), "SteadyAverageNs": sum(s) / len(s)})
        print(f"{n:20s} calls {len(v):3d}  avg {sum(v)/len(v)/1e6:.4f} ms  min {min(v)/1e6:.4f}  steady avg {steady[n]:.4f} ms")
    with open(os.path.join(prof, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)

# ---- counters (one --pmc pass per group)
pmc = {}
for sub in ("fetch", "write", "sqA", "sqB"):
    d = db(sub)
    if not d:
        continue
    acc = {}   # (kernel, counter) -> {dispatch: sum over the counter's hardware instances}
    for kname, disp, cname, val in d.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
        n = short(kname)
        if n:
            dd = acc.setdefault((n, cname), {})
            dd[disp] = dd.get(disp, 0.0) + float(val)
    for (n, c), dd in acc.items():
        pmc.setdefault(n, {})[c] = sum(dd.values()) / len(dd)
if pmc:
    for n, dd in pmc.items():
        if dd.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in dd and "GRBM_GUI_ACTIVE" in dd:
            # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs (cycles)
            cyc = dd["GRBM_GUI_ACTIVE"] / 8.0
            dd["_mfma_busy_frac"] = dd["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
            if n in steady:
                dd["_clock_GHz_from_GRBM"] = cyc / (steady[n] * 1e6)
    json.dump({"_note": "per-launch averages over the launches of each kernel in `bench.py --steps 6 --warmup 2` (metric shape); one "
                        "rocprofv3 --pmc pass per counter group; steady_ms from the kernel trace of `bench.py --steps 20 --warmup 5`",
               "steady_ms": steady, **pmc}, open(os.path.join(prof, f"{tag}_pmc.json"), "w"), indent=1)
    traffic, detail = {}, {}
    for n, dd in pmc.items():
        if "FETCH_SIZE" in dd and "WRITE_SIZE" in dd:
            rd = dd["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE counts half the bytes of wide reads (MI355X_MICROARCH.md, HBM)
            wr = dd["WRITE_SIZE"] * 1024
            traffic[n] = int(rd + wr)
            detail[n] = {"FETCH_SIZE_KB_per_launch": dd["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": dd["WRITE_SIZE"],
                         "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr}
    import re
    if traffic and re.fullmatch(r"r\d+_v\d+", tag):   # only the plain metric-shape profiles (rNN_vM) rewrite hbm_traffic.json   # '...ph' (phased-kernel A/B) and causal profiles leave hbm_traffic.json
        # (which bench.py quotes for the non-causal metric shape) alone
        print("hbm bytes per launch:", traffic)
        traffic["_detail"] = detail
        traffic["_note"] = ("bytes per launch at B=8,H=8,N=4096,d=64 bf16; FETCH_SIZE (KB) x 1024 x 2 (gfx950 wide-read correction) + "
                            f"WRITE_SIZE (KB) x 1024; separate --pmc passes ({tag}, profiles/README.md)")
        json.dump(traffic, open(os.path.join(prof, "hbm_traffic.json"), "w"), indent=1)
    for n, dd in pmc.items():
        print(n, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in dd.items() if k.startswith("_") or k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT")})
