R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr8 -- python3 $R/tools/diag/sink_steps.py 8 0 default_sink 120 > /dev/null 2> $O/tr8.err; echo "tr8 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trg -- python3 $R/tools/diag/sink_steps.py 1 0 none 600 > /dev/null 2> $O/trg.err; echo "trg rc=$?"
cd $R
cp $(find $O/tr8 -name "*kernel_stats.csv" | head -1) $O/eight_stream_sink_kernel_stats.csv
cp $(find $O/trg -name "*kernel_stats.csv" | head -1) $O/growing_600_kernel_stats.csv
find $O/tr8 $O/trg -name "*kernel_trace.csv" -delete
head -8 $O/eight_stream_sink_kernel_stats.csv | cut -c1-130; head -8 $O/growing_600_kernel_stats.csv | cut -c1-130
