#!/bin/bash
# One GPU session: parity tests, smoke, default bench (+CPU baseline), kernel trace, ingest timing (PMC passes: tools/pmc_round.sh, a call of its own).
# Summaries land in gpurun_out/round/ ; copy the ones to be judged into profiles/ (tracked).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu --durations=10 > $O/gpu_tests.log 2>&1; echo "TESTS rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "BENCH rc=$?"; cut -c1-400 $O/bench_default.json
timeout -k 10 120 python tools/diag/ingest_time.py > $O/ingest_time.txt 2>&1; echo "INGEST rc=$?"; grep "us/frame" $O/ingest_time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err; echo "TRACE rc=$?"
cd $R
cp $(find $O/prof_trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
find $O/prof_trace -name "*kernel_trace.csv" -size +30M -delete
head -12 $O/bench_kernel_stats.csv | cut -c1-150
