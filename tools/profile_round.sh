#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of the bench command, then separate --pmc passes
# (never combined with trace domains).  Outputs land under gpurun_out/<tag>_*; tools/summarize_profile.py turns
# them into the files committed under profiles/.
#   usage: tools/profile_round.sh <tag>        e.g. tools/profile_round.sh r01_v6
set -uo pipefail
TAG="${1:-prof}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"
S="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-breakdown ${BENCH_ARGS:-}"
run() { echo "=== $*"; timeout -k 10 300 "$@" > "$OUT/${TAG}_last.log" 2>&1; rc=$?; tail -n 3 "$OUT/${TAG}_last.log"; echo "=== exit $rc"; return $rc; }
run rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_trace" -o t -- $B || exit 1
grep '"metric"' "$OUT/${TAG}_last.log" > "$OUT/${TAG}_bench_under_trace.json" || true
run rocprofv3 --pmc FETCH_SIZE -d "$OUT/${TAG}_fetch" -o c -- $S || exit 1
run rocprofv3 --pmc WRITE_SIZE -d "$OUT/${TAG}_write" -o c -- $S || exit 1
run rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/${TAG}_sqA" -o c -- $S || exit 1
run rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES -d "$OUT/${TAG}_sqB" -o c -- $S || exit 1
find "$OUT" -path "*${TAG}_*" -name "*.csv" | head -20
