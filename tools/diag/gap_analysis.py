"""Idle time between consecutive kernels of a rocprofv3 kernel trace: python gap_analysis.py <kernel_trace.csv>
Prints, for the densest stretch of LM kernels, the sum of kernel durations, the wall span and the gap distribution."""
import csv, sys, statistics
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(sys.argv[1]))))
lm = [r for r in rows if any(k in r[2] for k in ("gemm_ws", "resid_norm", "qkv_finish", "attn_fwd", "attn_combine", "rmsnorm", "heads_kernel"))]
# take the last 224*8 LM kernels (steady state)
lm = lm[-224 * 8:]
gaps = [(lm[i + 1][0] - lm[i][1]) / 1e3 for i in range(len(lm) - 1)]
small = [g for g in gaps if g < 50]                      # ignore step boundaries (host work between steps)
busy = sum((e - s) for s, e, _ in lm) / 1e3
print(f"{len(lm)} LM kernels: busy {busy:.0f} us, gaps<50us: n={len(small)} sum {sum(small):.0f} us, median {statistics.median(small):.2f} us, "
      f"mean {statistics.mean(small):.2f} us, p90 {sorted(small)[int(0.9 * len(small))]:.2f} us; gaps>=50us: {len(gaps) - len(small)}")
per = {}
for i in range(len(lm) - 1):
    g = gaps[i]
    if g < 50: per.setdefault(lm[i + 1][2][-28:], []).append(g)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"  gap before {k:30s} n={len(v):4d} mean {statistics.mean(v):.2f} us")
