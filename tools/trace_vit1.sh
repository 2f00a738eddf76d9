#!/bin/bash
# per-position kernel durations of the one-frame vision encode, with and without the weight-prefetch riders (arg: vit_prefetch rows)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pf in 0 2400; do
  rm -rf $O/trv1_$pf
  AHA_VIT_PREFETCH=$pf rocprofv3 --kernel-trace --output-format csv -d $O/trv1_$pf -- python3 $R/tools/diag/vit_trace.py 1 bench > /dev/null 2> $O/trv1_$pf.err; echo "trv1_$pf rc=$?"
  python3 $R/tools/diag/trace_layer_seq.py $(find $O/trv1_$pf -name "*kernel_trace.csv" | head -1) 4 > $O/vit1_seq_pf$pf.txt; cat $O/vit1_seq_pf$pf.txt
  find $O/trv1_$pf -name "*kernel_trace.csv" -delete
done
