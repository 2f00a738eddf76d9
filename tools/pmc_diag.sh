#!/bin/bash
# Diagnosis of the rocprofv3 --pmc abort on evicting-cache runs: ONE pass per configuration, each in its own
# process under its own timeout, with the process map dumped so the raw frames of the tool's failure handler can be resolved.
#   none 64            growing cache to 2,304 keys: split attention + combine, no eviction, no re-rotation
#   sliding_window 64  evicts (ring wrap), no re-rotation
#   default_sink 64    evicts + re-rotates (3-D grid launch)
#   default_sink 40    SinkCache that never fills (no eviction)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcdiag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "none 64" "sliding_window 64" "default_sink 64" "default_sink 40"; do
  set -- $cfg; i=$((i+1)); tag="${1}_$2"
  export AHA_DUMP_MAPS=$O/maps_$tag.txt
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/out_$tag -- python3 $R/tools/diag/sink_steps.py 1 0 $1 $2 > $O/$tag.out 2> $O/$tag.err
  rc=$?; echo "PMCDIAG $tag rc=$rc"; grep -E "sink_steps|SIGSEGV|Aborted|PC:" $O/$tag.err | tail -4
  n=$(find $O/out_$tag -name "*counter_collection.csv" | wc -l); echo "  counter files: $n"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "PMCDIAG $tag hit its limit: stopping"; break; fi
done
find $O -name "*.csv" -size +8M -delete
exit 0
