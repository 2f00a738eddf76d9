#!/bin/bash
# Where attn_lm_kernel's time goes: diagnostic builds of attention.hip with parts of the kernel removed (AHA_ATTN_ABLATE bit mask:
# 1 no compute, 2 no softmax, 4 no result store, 8 no DMA), each timed by rocprofv3 --kernel-trace --stats on the 8-stream W = 2,048
# shape and on one 21.7k-key stream.
#   tools/diag/attn_lm_ablate.sh build        (here: cross-compiles aha-_amd/libaha_abl<N>.so)
#   tools/diag/attn_lm_ablate.sh run          (on the GPU box) -> gpurun_out/attn_lm_ablate.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/aha-_amd/csrc
VARIANTS="${VARIANTS:-0 1 2 4 8 6 9}"
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null || exit 1
  for n in $VARIANTS; do
    [ $n = 0 ] && continue
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -fno-honor-nans -DAHA_ATTN_ABLATE=$n -c $C/attention.hip -o /tmp/attention_abl$n.o || exit 1
    objs=$(ls $C/*.o | grep -v attention.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/aha-_amd/libaha_abl$n.so $objs /tmp/attention_abl$n.o -L/opt/rocm/lib -lrccl || exit 1
  done
  exit 0
fi
O=$R/gpurun_out; mkdir -p $O; : > $O/attn_lm_ablate.txt
cd /tmp && export TMPDIR=/tmp
for n in $VARIANTS; do
  if [ $n = 0 ]; then unset AHA_AMD_LIB; else export AHA_AMD_LIB=$R/aha-_amd/libaha_abl$n.so; fi
  for shape in ${SHAPES:-8:2048 1:21763}; do
    set -- ${shape/:/ }; export B=$1
    rm -rf $O/abl_prof
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_prof -- python3 $R/tools/diag/long_attn.py $2 > $O/abl_run.log 2>&1 || { echo "variant $n shape $shape failed"; tail -3 $O/abl_run.log; exit 1; }
    echo "ablate=$n B=$1 Lk=$2" | tee -a $O/attn_lm_ablate.txt
    python3 $R/tools/diag/kstats.py $O/abl_prof 12 | grep -E "attn_" | tee -a $O/attn_lm_ablate.txt
  done
done
rm -rf $O/abl_prof
