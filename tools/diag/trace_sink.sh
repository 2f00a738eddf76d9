cd /tmp && export TMPDIR=/tmp
for B in 1 8; do
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr_sink_$B -o t -- python3 $GRAFT_REPO_ROOT/tools/diag/sink_steps.py $B > $GRAFT_REPO_ROOT/gpurun_out/tr_sink_$B.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/tools/diag/trace_tail.py $GRAFT_REPO_ROOT/gpurun_out/tr_sink_$B 3000 > $GRAFT_REPO_ROOT/gpurun_out/tr_sink_tail_$B.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/tr_sink_$B
done
