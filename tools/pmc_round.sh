#!/bin/bash
# Counter passes (rounds 3-4) (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc runs, no tracing flags, the program
# itself behind `--`).  Four workloads for the HBM-traffic summary bench.py reads (each row carries its workload):
#   static_1stream       headline step (gemm_ws kernels, fused static attention)      bench.py --no-secondary
#   static_8stream       M = 288 step (gemm_wl kernels)                                bench.py --streams 8 --no-secondary
#   sink_1stream_steady  1 stream, SinkCache W=2048: evicts + re-rotates + attends over 2,048 keys   tools/diag/sink_steps.py 1
#   sink_8stream_steady  8 streams of the same                                         tools/diag/sink_steps.py 8
# and one SQ/GRBM pass for MFMA utilisation on the vision tower (4 layers, 32 frames) and on the 8-stream LM step.
# Output: gpurun_out/round/pmc_hbm_traffic.json, pmc_mfma_vit32.json, pmc_mfma_lm8.json   (copy into profiles/r05_*)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O; rm -rf $O/pmc_*
cd /tmp && export TMPDIR=/tmp
run() {   # tag counter(s) program args...
  local tag=$1 ctr=$2; shift 2
  timeout -k 10 400 rocprofv3 --pmc $ctr $FILTER --output-format csv -d $O/pmc_$tag -- "$@" > /dev/null 2> $O/pmc_$tag.err
  local rc=$?; echo "PMC $tag rc=$rc"; [ $rc -eq 124 -o $rc -eq 137 ] && { echo "PMC $tag hit its limit: stopping"; exit 1; }
  # (round 4 deleted CSVs above 40 MB here, BEFORE the summaries were computed: the 6-counter pass of the 120-step 8-stream run lost
  # its table and profiles/r04_pmc_mfma_lm8.json came out empty.  The raw tables are removed at the end of this script, after the
  # summaries have read them; the summaries themselves exit non-zero on an empty table.)
}
B1="python3 $R/bench.py --steps 1 --warmup 0 --frames 4 --no-cpu-baseline --no-secondary"
B8="python3 $R/bench.py --streams 8 --steps 1 --warmup 0 --frames 4 --no-cpu-baseline --no-secondary"
S1="python3 $R/tools/diag/sink_steps.py 1 0 default_sink 120"
S8="python3 $R/tools/diag/sink_steps.py 8 0 default_sink 120"
G1="python3 $R/tools/diag/sink_steps.py 1 0 none 600"          # growing cache to 21.6k keys; attention kernels only (120k dispatches otherwise)
for c in FETCH_SIZE WRITE_SIZE; do
  x=f; [ $c = WRITE_SIZE ] && x=w
  run s1_$x $c $B1; run s8_$x $c $B8; run k1_$x $c $S1; run k8_$x $c $S8
  FILTER="--kernel-include-regex attn_" run g1_$x $c $G1
done
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
run mfma_vit "$SQ" python3 $R/tools/diag/vit_only.py 1
FILTER="--kernel-include-regex gemm_wl|attn_|resid_norm|qkv_finish" run mfma_lm8 "$SQ" python3 $R/tools/diag/sink_steps.py 8 0 default_sink 90
cd $R
python3 tools/pmc_summary.py $O/pmc_hbm_traffic.json \
  "static_1stream=$O/pmc_s1_f,$O/pmc_s1_w,bench.py --steps 1 --warmup 0 --frames 4 --no-cpu-baseline --no-secondary" \
  "static_8stream=$O/pmc_s8_f,$O/pmc_s8_w,bench.py --streams 8 --steps 1 --warmup 0 --frames 4 --no-cpu-baseline --no-secondary" \
  "sink_1stream_steady=$O/pmc_k1_f,$O/pmc_k1_w,tools/diag/sink_steps.py 1 0 default_sink 120 (second half of each kernel's dispatches = cache full, evicting)" \
  "sink_8stream_steady=$O/pmc_k8_f,$O/pmc_k8_w,tools/diag/sink_steps.py 8 0 default_sink 120 (second half of each kernel's dispatches)" \
  "growing_1stream_tail=$O/pmc_g1_f,$O/pmc_g1_w,tools/diag/sink_steps.py 1 0 none 600 (attention kernels only; last 5 % of the dispatches = 20.5k-21.6k keys)" > $O/pmc_hbm_traffic.txt
tail -30 $O/pmc_hbm_traffic.txt
python3 tools/pmc_mfma_summary.py $O/pmc_mfma_vit $O/pmc_mfma_vit32.json || echo "PMC SUMMARY mfma_vit FAILED"
python3 tools/pmc_mfma_summary.py $O/pmc_mfma_lm8 $O/pmc_mfma_lm8.json || echo "PMC SUMMARY mfma_lm8 FAILED"
rm -rf $O/pmc_s1_* $O/pmc_s8_* $O/pmc_k1_* $O/pmc_k8_* $O/pmc_g1_* $O/pmc_mfma_vit $O/pmc_mfma_lm8
