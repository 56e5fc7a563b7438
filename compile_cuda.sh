#!/usr/bin/env bash
# Builds the MI355X (gfx950) FlashAttention libraries with hipcc.  Name kept from the reference
# (compile_cuda.sh / Makefile:28-50 there built the CUDA .so files with nvcc); the output directory
# holds the same six library names the reference's cuda_kernel_ops.py:30-35 opens, plus the core
# library they forward to.
#   OUT_DIR=minitorch/cuda_kernels ./compile_cuda.sh     # drop-in location inside a minitorch checkout
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
SRC="$HERE/flash_attention_minitorch_amd/csrc"
OUT_DIR="${OUT_DIR:-$HERE/flash_attention_minitorch_amd/cuda_kernels}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
ARCH="${FA_ARCH:-gfx950}"
mkdir -p "$OUT_DIR"

CORE="$OUT_DIR/libflash_attn_mi355x.so"
DIAG="$OUT_DIR/libflash_attn_mi355x_diag.so"   # stamp / ablation builds + fa_mi355x_set_tuning: tools/ only (FA_SKIP_DIAG=1 skips it)
FLAGS=(--offload-arch="$ARCH" -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize ${FA_EXTRA_FLAGS:-})
stale() { [ ! -f "$1" ] || [ -n "$(find "$SRC" "$HERE/include" -newer "$1" \( -name '*.h' -o -name '*.hip' \) -print -quit)" ]; }
pids=()
if stale "$CORE"; then
  echo "[compile_cuda.sh] hipcc --offload-arch=$ARCH  fa_api.hip -> $CORE"
  "$HIPCC" "${FLAGS[@]}" "$SRC/fa_api.hip" -o "$CORE" & pids+=($!)
else
  echo "[compile_cuda.sh] $CORE is up to date"
fi
if [ -z "${FA_SKIP_DIAG:-}" ] && stale "$DIAG"; then
  echo "[compile_cuda.sh] hipcc --offload-arch=$ARCH  -DFA_DIAG fa_api.hip -> $DIAG"
  "$HIPCC" "${FLAGS[@]}" -DFA_DIAG "$SRC/fa_api.hip" -o "$DIAG" & pids+=($!)
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done

shim() {  # name variant FW|BW
  "$HIPCC" -O2 -fPIC -shared -x c++ "$SRC/fa_shim.cpp" -DFA_SHIM_VARIANT="$2" -DFA_SHIM_"$3" \
      -L"$OUT_DIR" -lflash_attn_mi355x -Wl,-rpath,'$ORIGIN' -o "$OUT_DIR/$1.so"
}
shim flash_attn_fw        1 FW
shim flash_attn_bw        1 BW
shim flash_attn_causal_fw 1 FW
shim flash_attn_causal_bw 1 BW
shim flash_attn2_fw       2 FW
shim flash_attn2_bw       2 BW
echo "[compile_cuda.sh] built: $(ls "$OUT_DIR" | tr '\n' ' ')"
