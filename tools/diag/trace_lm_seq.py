#!/usr/bin/env python3
"""Per-position kernel durations of the LM step's per-layer launch sequence from a rocprofv3 --kernel-trace CSV.  A layer is anchored at its
gate/up launch (the one kernel name that occurs once per layer); the launches from five before it to three after it are aligned by position
and the median duration per position is printed, with the median gap to the previous launch.  Only the last `layers` layers are used (steady state).
    python tools/diag/trace_lm_seq.py TRACE.csv [anchor substring] [layers]"""
import csv, statistics, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if n.startswith("void at::") or "rocclr" in n:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "gemm_wl_bal18"
n_layers = int(sys.argv[3]) if len(sys.argv) > 3 else 28 * 8
idx = [i for i, r in enumerate(rows) if anchor in r[2]][-n_layers:]
per = {}
for a, nxt in zip(idx, idx[1:]):
    # a layer's launches: from the first launch after the previous anchor's two followers up to this anchor's two followers
    lo = a
    while lo > 0 and a - lo < 12 and anchor not in rows[lo - 1][2]:
        lo -= 1
    for i in range(max(lo + 2, a - 6), min(a + 3, len(rows))):
        d = (rows[i][1] - rows[i][0]) / 1e3
        gap = (rows[i][0] - rows[i - 1][1]) / 1e3
        per.setdefault((i - a, rows[i][2].split("(")[0][-46:]), []).append((d, gap))
tot = 0.0
for (pos, name), v in sorted(per.items()):
    if len(v) < len(idx) // 4:
        continue
    d = statistics.median(x[0] for x in v); g = statistics.median(x[1] for x in v)
    tot += d + g
    print(f"  pos {pos:3d} {name:48s} n={len(v):4d} median {d:7.2f} us  gap before {g:5.2f} us")
print(f"sum of medians + gaps over the positions of a layer: {tot:.1f} us")
