#!/bin/bash
# SQ counters of the dense (vision) attention kernels at 32 frames -> gpurun_out/round/pmc_attn.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O; rm -rf $O/pmc_attn
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "attn_head64|attn_dense" --output-format csv -d $O/pmc_attn/$tag -- python3 $R/tools/diag/vit_attn_time.py 32 > /dev/null 2> $O/pmc_attn_$tag.err; echo "PMC attn $tag rc=$?"
done
python3 - <<EOF
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$O/pmc_attn/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open("$O/pmc_attn.txt", "w") as out:
    for k, c in acc.items():
        line = k + ": " + "  ".join(f"{n} {v[1]/v[0]:.3e}" for n, v in sorted(c.items()))
        print(line); out.write(line + "\n")
EOF
rm -rf $O/pmc_attn
