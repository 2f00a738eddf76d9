#!/usr/bin/env python3
"""Idle time between kernels from a rocprofv3 --kernel-trace CSV: for the LAST `window` seconds of the trace (or all of it) the sum of kernel
durations against the span they cover, and the largest gaps with the kernels on either side.   python tools/diag/trace_gaps.py TRACE.csv [tail_ms]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
if tail_ms > 0:
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - tail_ms * 1e6]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = sorted(((rows[i + 1][0] - rows[i][1], rows[i][2][:50], rows[i + 1][2][:50]) for i in range(len(rows) - 1)), reverse=True)
print(f"{len(rows)} dispatches over {span / 1e6:.3f} ms; kernels busy {busy / 1e6:.3f} ms = {busy / span:.3f}; mean gap {(span - busy) / max(1, len(rows) - 1) / 1e3:.2f} us")
for g, a, b in gaps[:12]:
    print(f"  gap {g / 1e3:8.1f} us  after {a:50s} before {b}")
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    agg[n[:60]][0] += 1; agg[n[:60]][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print(f"  {n:60s} n={c:5d} total {t / 1e6:8.3f} ms avg {t / c / 1e3:7.1f} us")
