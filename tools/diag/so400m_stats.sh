#!/bin/bash
# per-kernel stats of 32-frame vision encodes on the so400m/14@384 geometry -> gpurun_out/so400m_stats.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $O/so4prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/so4prof -- python3 $R/tools/diag/vit_trace.py 32 ref > $O/so4_run.log 2>&1 || { tail -5 $O/so4_run.log; exit 1; }
python3 $R/tools/diag/kstats.py $O/so4prof 14 | tee $O/so400m_stats.txt
rm -rf $O/so4prof
