R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O; rm -rf $O/trv $O/trv1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trv -- python3 $R/tools/diag/vit_trace.py 32 bench > /dev/null 2> $O/trv.err; echo "trv rc=$?"
rocprofv3 --kernel-trace --output-format csv -d $O/trv1 -- python3 $R/tools/diag/vit_trace.py 1 bench > /dev/null 2> $O/trv1.err; echo "trv1 rc=$?"
cd $R
python3 tools/diag/trace_gaps.py $(find $O/trv -name "*kernel_trace.csv" | head -1) 40 > $O/vit32_gaps.txt; cat $O/vit32_gaps.txt
python3 tools/diag/trace_gaps.py $(find $O/trv1 -name "*kernel_trace.csv" | head -1) 6 > $O/vit1_gaps.txt; cat $O/vit1_gaps.txt
find $O/trv $O/trv1 -name "*kernel_trace.csv" -delete
