#!/bin/bash
# HBM-traffic counters (separate FETCH_SIZE / WRITE_SIZE passes, MI355X_MICROARCH.md HBM section) for the kernels the bench
# quotes rooflines for.  Two workloads, static cache: the headline step (1 stream: gemm_ws kernels) and the 8-stream step
# (M = 288: gemm_wl kernels).  Known limit (round 2): rocprofv3 --pmc crashes (SIGSEGV inside the tool's dispatch interception)
# or stalls on every run that uses an EVICTING cache policy, while --kernel-trace on the same commands works; the steady-state
# sink kernels (attn_fwd / attn_lm at 2,048 keys, sink_rerotate) therefore have no PMC traffic figure (bench.py reports null).
# Output: gpurun_out/round/pmc_hbm_traffic.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O; rm -rf $O/pmc_f $O/pmc_w
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc_f; [ $c = WRITE_SIZE ] && d=$O/pmc_w
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $d/head -- python3 $R/bench.py --steps 1 --warmup 0 --frames 4 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_head_$c.err; echo "PMC $c head rc=$?"
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $d/b8 -- python3 $R/bench.py --streams 8 --steps 1 --warmup 0 --frames 4 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_b8_$c.err; echo "PMC $c 8-stream rc=$?"
done
cd $R
python tools/pmc_summary.py $O/pmc_f $O/pmc_w $O/pmc_hbm_traffic.json
rm -rf $O/pmc_f $O/pmc_w
