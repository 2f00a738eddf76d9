#!/bin/bash
# Does the vision tower really overlap with the LM chain?  Headline loop with different tower GEMM variants.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
out=$O/overlap_sweep.txt; : > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 240 python bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>>$O/overlap.err \
        | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])" | tee -a $out; }
run
run --no-overlap
run --tile-dma 10
run --tile-dma 10 --no-overlap
run --tile-dma 2
run --tile-dma 5
run --tile-dma 0
