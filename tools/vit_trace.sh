#!/bin/bash
# kernel trace of ten 32-frame vision encodes: tools/vit_trace.sh [preset]  -> gpurun_out/vit_trace_<preset>.csv
R=$GRAFT_REPO_ROOT; P=${1:-bench}; O=$R/gpurun_out/vit_trace_$P; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/diag/vit_trace.py 32 $P > $O/out.txt 2>&1; echo "rc=$?"
cd $R
for f in $O/*/*kernel_stats.csv; do cp $f $R/gpurun_out/vit_trace_$P.csv; done
rm -rf $O
cut -c1-160 $R/gpurun_out/vit_trace_$P.csv
