#!/bin/bash
# round-3 GPU session helper: run the listed steps in order, stop at the first step that timed out / was killed (no further GPU
# work after a hang), keep going after ordinary failures.  usage: scripts/r3_run.sh TAG  (steps are defined below per TAG)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
step() {  # step NAME SECONDS cmd...
  local name=$1 secs=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/$TAG.steps
  timeout -k 10 $secs "$@" > gpurun_out/${TAG}_$name.log 2> gpurun_out/${TAG}_$name.err
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/$TAG.steps
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in $name: stopping" | tee -a gpurun_out/$TAG.steps; exit 1; fi
  return 0
}
TAG=$1
