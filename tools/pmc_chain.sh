#!/usr/bin/env bash
# rocprofv3 counter passes of the chained one-pass backward and its timing ablations (diagnostic library), one pass per counter group
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp FA_MI355X_DIAG=1; cd "$R"
for v in ${VARIANTS:-"0,0,0,0,3" "0,0,0,0,3,1" "0,0,0,0,3,2304"}; do
  tag=$(echo "$v" | tr ',' '_' | tr -d '-'); tag=${tag:-split}
  i=0
  for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $grp -d "$OUT/pmcc_${tag}_$i" -o c -- python3 "$R/tools/prof_chain.py" "$v" 4 > "$OUT/pmcc_${tag}_$i.log" 2>&1
    echo "$tag group $i rc=$?"; tail -n 2 "$OUT/pmcc_${tag}_$i.log"
  done
done
python3 - <<'PY'
import glob, os, sqlite3, json
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
res = {}
for d in sorted(glob.glob(os.path.join(out, "pmcc_*_[0-9]"))):
    f = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if not f: continue
    db = sqlite3.connect(f[0])
    tag = os.path.basename(d)[5:-2]
    acc = {}
    try:
        rows = db.execute("select name, dispatch_id, counter_name, counter_value from pmc_events")
    except Exception as e:
        print(d, e); continue
    for kname, disp, cname, val in rows:
        if "fa" not in kname: continue
        kk = [k for k in ("bwd_chain_kernel", "bwd_dkdv_slot_kernel", "bwd_dq_slot_kernel", "bwd_prep_kernel") if k in kname]
        if not kk: continue
        dd = acc.setdefault((kk[0], cname), {})
        dd[disp] = dd.get(disp, 0.0) + float(val)
    for (k, c), dd in acc.items():
        res.setdefault(tag, {}).setdefault(k, {})[c] = sum(dd.values()) / len(dd)
json.dump(res, open(os.path.join(out, "pmc_chain.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True)[:6000])
PY
rm -rf "$OUT"/pmcc_*_[0-9]
