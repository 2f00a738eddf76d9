#!/bin/bash
# Round-5 kernel-stat profiles: vision encode at 32 frames and at one frame, the 8-stream SinkCache step; a second bench sample.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 32 1; do
  rm -rf $O/vt$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/vt$n -- python3 $R/tools/diag/vit_trace.py $n bench > /dev/null 2> $O/vt$n.err; echo "vit$n rc=$?"
  cp $(find $O/vt$n -name "*kernel_stats.csv" | head -1) $O/vit${n}_kernel_stats.csv
  [ $n = 1 ] && python3 $R/tools/diag/trace_layer_seq.py $(find $O/vt1 -name "*kernel_trace.csv" | head -1) 4 > $O/vit1_layer_seq.txt
  rm -rf $O/vt$n
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr8 -- python3 $R/tools/diag/sink_steps.py 8 0 default_sink 120 > /dev/null 2> $O/tr8.err; echo "tr8 rc=$?"
cp $(find $O/tr8 -name "*kernel_stats.csv" | head -1) $O/eight_stream_sink_kernel_stats.csv; rm -rf $O/tr8
cd $R
timeout -k 10 600 python bench.py > $O/bench_second.json 2> $O/bench_second.err; echo "BENCH2 rc=$?"
head -6 $O/vit32_kernel_stats.csv | cut -c1-140; cat $O/vit1_layer_seq.txt; head -8 $O/eight_stream_sink_kernel_stats.csv | cut -c1-140
