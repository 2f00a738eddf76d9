#!/bin/bash
# Run GPU steps one after another on a gpurun box, each under its own `timeout -k`, logging to gpurun_out/<name>.log.
# Stops at the first step that was killed at its limit (rc 124 / 137): after a hang nothing else touches the GPU.
#   tools/gpu_steps.sh "name|seconds|command" ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
for step in "$@"; do
  name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 $secs bash -c "$cmd" > $O/$name.log 2>&1
  rc=$?
  echo "=== $name rc=$rc $(( $(date +%s) - start ))s"; tail -4 $O/$name.log | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name was killed at its limit: stopping"; exit $rc; fi
done
exit 0
