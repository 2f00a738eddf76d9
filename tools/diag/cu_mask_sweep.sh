#!/bin/bash
# Sweep CU partitions between the vision stream and the LM stream (bench.py --vit-cus / --lm-cus).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
out=$O/cu_mask_sweep.txt; : > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 240 python bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>>$O/cu_mask.err \
        | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])" | tee -a $out; }
run
run --vit-cus 32
run --vit-cus 48
run --vit-cus 64
run --vit-cus 96
run --vit-cus 48 --lm-cus 0
run --vit-cus 64 --lm-cus 0
run --vit-cus 96 --lm-cus 0
