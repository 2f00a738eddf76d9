#!/bin/bash
# kernel trace of the overlapped headline step: do tower kernels and LM kernels actually run at the same time?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ovl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --overlap "$@" > $O/out.json 2> $O/err.txt; echo "rc=$?"
cd $R
python3 tools/diag/overlap_trace.py $(ls $O/*/*kernel_trace.csv)
rm -rf $O/*/
