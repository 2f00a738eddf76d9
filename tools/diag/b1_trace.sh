#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $O/b1prof
rocprofv3 --kernel-trace --output-format csv -d $O/b1prof -- python3 $R/tools/diag/sink_steps.py 1 0 none 40 > $O/b1_run.log 2>&1
f=$(find $O/b1prof -name "*kernel_trace.csv" | head -1)
python3 $R/tools/diag/trace_lm_seq.py $f "gemm_ws_kernel<3, 2, 4, 2, 5>" 112
rm -rf $O/b1prof
