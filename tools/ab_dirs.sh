#!/usr/bin/env bash
# Same-box A/B of two builds of the library: runs <command> alternately with the in-tree build and with the build in gpurun_ab/
# (FA_MI355X_KERNEL_DIR), ABAB, one process each.  usage: tools/ab_dirs.sh "<command>" [rounds]
cd "${GRAFT_REPO_ROOT:-.}"
for i in $(seq 1 "${2:-2}"); do
  echo "--- round $i: in-tree (new)"; bash -c "$1"
  echo "--- round $i: gpurun_ab (old)"; FA_MI355X_KERNEL_DIR="$PWD/gpurun_ab" bash -c "$1"
done
