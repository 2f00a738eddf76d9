#!/bin/bash
# One GPU session: parity tests, smoke, default bench (+CPU baseline), kernel trace, PMC passes.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "TESTS rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "BENCH rc=$?"; cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err; echo "TRACE rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --frames 4 --no-cpu-baseline > $O/bench_fetch.json 2> $O/bench_fetch.err; echo "PMC_FETCH rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 1 --warmup 0 --frames 4 --no-cpu-baseline > $O/bench_write.json 2> $O/bench_write.err; echo "PMC_WRITE rc=$?"
cd $R; find $O/prof_fetch $O/prof_write -name "*.csv" | head; ls -la $O/prof_fetch/*/ 2>/dev/null | head
