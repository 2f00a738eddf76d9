#!/bin/bash
# One SQ/GRBM counter pass on the 8-stream LM step (M = 288 rows: the mid-M kernels) -> gpurun_out/round/pmc_mfma_lm8.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O; rm -rf $O/pmc_mfma_lm8
cd /tmp && export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_mfma_lm8 -- python3 $R/bench.py --streams 8 --steps 1 --warmup 0 --frames 4 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_mfma_lm8.err
echo "PMC rc=$?"
cd $R
python3 tools/pmc_mfma_summary.py $O/pmc_mfma_lm8 $O/pmc_mfma_lm8.json
rm -rf $O/pmc_mfma_lm8
python3 -c "
import json; d=json.load(open('$O/pmc_mfma_lm8.json'))
rows = d if isinstance(d, list) else d.get('kernels', d)
for r in rows[:14]: print(r.get('kernel','')[-50:], r.get('workgroups'), r.get('dispatches'), r.get('mfma_util'), r.get('wave_parked_frac'), r.get('issue_stall_frac'))
"
