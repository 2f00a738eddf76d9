#!/bin/bash
# Where the split-K mid-M GEMMs' time goes inside the 8-stream LM step: diagnostic builds of gemm_wl.hip with parts of gemm_wl_kernel removed
# (AHA_WL_ABLATE bit mask: 1 no MFMA, 2 no result store, 4 no W DMA, 8 no X DMA), per-position kernel durations of a layer from a kernel trace.
#   tools/diag/wl_ablate.sh build        (here: cross-compiles aha-_amd/libaha_wlabl<N>.so)
#   tools/diag/wl_ablate.sh run          (on the GPU box) -> gpurun_out/wl_ablate.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/aha-_amd/csrc
VARIANTS="${VARIANTS:-0 1 2 4 8 12 15}"
if [ "$1" = build ]; then
  make -C $C -j8 > /dev/null || exit 1
  for n in $VARIANTS; do
    [ $n = 0 ] && continue
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -DAHA_WL_ABLATE=$n -c $C/gemm_wl.hip -o /tmp/gemm_wl_abl$n.o || exit 1
    objs=$(ls $C/*.o | grep -v gemm_wl.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/aha-_amd/libaha_wlabl$n.so $objs /tmp/gemm_wl_abl$n.o -L/opt/rocm/lib -lrccl || exit 1
  done
  exit 0
fi
O=$R/gpurun_out; mkdir -p $O; : > $O/wl_ablate.txt
cd /tmp && export TMPDIR=/tmp
for n in $VARIANTS; do
  if [ $n = 0 ]; then unset AHA_AMD_LIB; else export AHA_AMD_LIB=$R/aha-_amd/libaha_wlabl$n.so; fi
  rm -rf $O/wlabl_prof
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/wlabl_prof -- python3 $R/tools/diag/sink_steps.py 8 0 default_sink 72 > $O/wlabl_run.log 2>&1 || { echo "variant $n failed"; tail -3 $O/wlabl_run.log; exit 1; }
  echo "ablate=$n" | tee -a $O/wl_ablate.txt
  python3 $R/tools/diag/trace_lm_seq.py $(find $O/wlabl_prof -name "*kernel_trace.csv" | head -1) gemm_wl_bal18 112 | tee -a $O/wl_ablate.txt
done
rm -rf $O/wlabl_prof
