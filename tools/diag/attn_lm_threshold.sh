#!/bin/bash
# one stream, T = 36: cache attention by key count with the automatic kernel choice (attn_lm=1) and with attn_lm_kernel forced (attn_lm=2)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
export B=1
for Lk in 2084 3000 4000 5000 6500 8000; do
  for v in 1 2; do
    rm -rf $O/thrprof
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/thrprof -- python3 $R/tools/diag/long_attn.py $Lk attn_lm=$v > /dev/null 2>&1
    echo "Lk=$Lk attn_lm=$v:"; python3 $R/tools/diag/kstats.py $O/thrprof 12 | grep -E 'attn_' | cut -c1-100
  done
done
rm -rf $O/thrprof
