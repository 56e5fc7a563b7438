#!/usr/bin/env bash
# Runs GPU steps one after another on the gpurun box, logging each under gpurun_out/.
# An ordinary failure (non-zero exit) does not stop the sequence; a step that TIMES OUT or is killed does
# (no further GPU step is started after a hang).
# usage: tools/gpu_steps.sh name1 "cmd1" name2 "cmd2" ...
mkdir -p gpurun_out
rc_all=0
while [ $# -ge 2 ]; do
  name="$1"; cmd="$2"; shift 2
  echo "=== step $name: $cmd"
  timeout -k 10 "${STEP_TIMEOUT:-420}" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== step $name exit $rc"; tail -n "${STEP_TAIL:-15}" "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / killed: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
