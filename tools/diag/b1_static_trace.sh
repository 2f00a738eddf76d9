#!/bin/bash
# per-position durations AND gaps of the headline LM step's layer (graph replay); extra args = tunings (engine=2 ...); ANCHOR = kernel-name substring
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $O/b1s
rocprofv3 --kernel-trace --output-format csv -d $O/b1s -- python3 $R/tools/diag/static_trace.py 12 "$@" > $O/b1s_run.log 2>&1
f=$(find $O/b1s -name "*kernel_trace.csv" | head -1)
python3 $R/tools/diag/trace_lm_seq.py $f "${ANCHOR:-gemm_ws_kernel<3, 2, 4, 2, 5>}" 140
rm -rf $O/b1s
