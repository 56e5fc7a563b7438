#!/usr/bin/env bash
# rocprofv3 evidence for the fp32 one-pass backward at BASELINE configs[2]: a kernel trace with --stats, then ONE counter pass (SQ group),
# each in its own run (gpurun refuses --pmc together with trace domains).  Writes gpurun_out/onepass_f32_{stats.csv,pmc.json}.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$R"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/op32_trace" -o t -f csv -- python3 "$R/tools/prof_onepass_f32.py" 20 > "$OUT/op32_trace.log" 2>&1
echo "trace rc=$?"; tail -n 1 "$OUT/op32_trace.log"
f=$(find "$OUT/op32_trace" -name '*kernel_stats.csv' | head -n 1); [ -n "$f" ] && cp "$f" "$OUT/onepass_f32_stats.csv" && head -n 8 "$OUT/onepass_f32_stats.csv"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS \
  -d "$OUT/op32_pmc" -o c -- python3 "$R/tools/prof_onepass_f32.py" 4 > "$OUT/op32_pmc.log" 2>&1
echo "pmc rc=$?"; tail -n 1 "$OUT/op32_pmc.log"
python3 - <<'PY'
import glob, os, sqlite3, json
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
res = {}
for f in glob.glob(os.path.join(out, "op32_pmc", "**", "*_results.db"), recursive=True):
    db = sqlite3.connect(f)
    acc = {}
    for kname, disp, cname, val in db.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
        kk = [k for k in ("bwd_onepass_f32_kernel", "bwd_prep_kernel", "fwd_kernel") if k in kname]
        if not kk: continue
        dd = acc.setdefault((kk[0], cname), {})
        dd[disp] = dd.get(disp, 0.0) + float(val)
    for (k, c), dd in acc.items():
        res.setdefault(k, {})[c] = sum(dd.values()) / len(dd)
json.dump(res, open(os.path.join(out, "onepass_f32_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
rm -rf "$OUT/op32_trace" "$OUT/op32_pmc"
