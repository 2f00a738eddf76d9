#!/bin/bash
# Round-6 measurement session: default bench, the same with the tower on its own stream (overlap at 8 streams), a second default
# sample, kernel trace of the bench, the 8-stream step's and the vision encodes' kernel stats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round6; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "BENCH rc=$?"; cut -c1-300 $O/bench_default.json
timeout -k 10 600 python bench.py --overlap --no-cpu-baseline > $O/bench_overlap.json 2> $O/bench_overlap.err; echo "BENCH overlap rc=$?"
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_second.json 2> $O/bench_second.err; echo "BENCH2 rc=$?"
timeout -k 10 600 python bench.py --overlap --no-cpu-baseline > $O/bench_overlap2.json 2> $O/bench_overlap2.err; echo "BENCH overlap2 rc=$?"
python - <<PY
import json
for f in ("bench_default", "bench_overlap", "bench_second", "bench_overlap2"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        e = d.get("eight_stream_sink") or {}
        print(f, "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "p50", d.get("p50_frame_latency_ms"), "| eight_stream_sink", e.get("frames_per_s"), "lm_step_ms", e.get("lm_step_ms"), "ms_per_step", e.get("ms_per_step"))
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err; echo "TRACE rc=$?"
cp $(find $O/prof_trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/prof_trace
for n in 32 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/vt$n -- python3 $R/tools/diag/vit_trace.py $n bench > /dev/null 2> $O/vt$n.err; echo "vit$n rc=$?"
  cp $(find $O/vt$n -name "*kernel_stats.csv" | head -1) $O/vit${n}_kernel_stats.csv
  [ $n = 1 ] && python3 $R/tools/diag/trace_layer_seq.py $(find $O/vt1 -name "*kernel_trace.csv" | head -1) 4 > $O/vit1_layer_seq.txt
  rm -rf $O/vt$n
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr8 -- python3 $R/tools/diag/sink_steps.py 8 0 default_sink 120 > /dev/null 2> $O/tr8.err; echo "tr8 rc=$?"
cp $(find $O/tr8 -name "*kernel_stats.csv" | head -1) $O/eight_stream_sink_kernel_stats.csv; rm -rf $O/tr8
cd $R
head -10 $O/bench_kernel_stats.csv | cut -c1-150; cat $O/vit1_layer_seq.txt
