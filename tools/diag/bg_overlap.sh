#!/bin/bash
# headline step with the encode of batch k+1 underneath the LM steps of batch k: serial / overlap with the persistent tower /
# overlap with the background tower (with and without a high-priority LM stream)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
run() { echo "== $*"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 "$@" 2>>$O/bg.err \
        | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms/step')"; }
run
run --overlap --tower-bg 0
run --overlap --tower-bg 0 --lm-priority
run --overlap --tower-bg 1
run --overlap --tower-bg 1 --lm-priority
