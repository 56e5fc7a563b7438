#!/usr/bin/env bash
# usage: tools/isa_mix.sh <mangled-kernel-substring>...   -> resource usage + per-basic-block instruction mix
# (compiles csrc/fa_api.hip with -save-temps into /tmp/isa unless FA_ISA_NOBUILD=1)
set -e
mkdir -p /tmp/isa && cd /tmp/isa
if [ -z "$FA_ISA_NOBUILD" ]; then
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize ${FA_EXTRA_FLAGS:-} \
  -Rpass-analysis=kernel-resource-usage -save-temps /root/repo/flash_attention_minitorch_amd/csrc/fa_api.hip -o /tmp/isa/core.so 2> res.txt
grep -E "error" -A3 res.txt | head -20 || true
fi
S=fa_api-hip-amdgcn-amd-amdhsa-gfx950.s
for K in "$@"; do
  echo "== $K"
  grep -A12 "Function Name: .*${K}" res.txt | grep -E " VGPRs:|AGPRs|Scratch|Occupancy|LDS Size" | sed 's/\[-Rpass.*//; s/.*remark: [^ ]* //' | tr '\n' ' '; echo
  start=$(grep -n "^_ZN2fa.*${K}.*:" $S | head -1 | cut -d: -f1)
  awk -v s="$start" 'NR>=s' $S | awk '/^\.Lfunc_end/{exit} {print}' > kern_$K.s
  awk '/^\.LBB[0-9_]+:/{lbl=$1} {c[lbl]++; if($1 ~ /^v_mfma/) m[lbl]++; else if($1 ~ /^v_exp/) e[lbl]++; else if($1 ~ /^v_/) v[lbl]++; if($1 ~ /^ds_/) d[lbl]++; if ($1 ~ /^s_/) s[lbl]++; if ($1 ~ /^(global|buffer)_/) g[lbl]++} END{for(l in c) if (c[l]>'${MINSZ:-30}') print l, "total",c[l],"mfma",m[l]+0,"exp",e[l]+0,"valu",v[l]+0,"ds",d[l]+0,"salu",s[l]+0,"vmem",g[l]+0}' kern_$K.s | sort -t_ -k2 -n
done
