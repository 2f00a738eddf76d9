#!/bin/bash
# Rebuild libaha_amd.so (the built .so travels with the snapshot) and run GPU steps on an MI355X box:
#   tools/grun.sh [gpurun timeout] "name|seconds|command" ...
T=1200
if [[ "$1" =~ ^[0-9]+$ ]]; then T=$1; shift; fi
make -C "$(dirname "$0")/../aha-_amd/csrc" -j8 2>&1 | grep -E "error|Error" && exit 1
args=""
for s in "$@"; do args="$args \"${s//\"/\\\"}\""; done
exec /usr/local/graft/bin/gpurun --timeout $T -- "tools/gpu_steps.sh $args"
