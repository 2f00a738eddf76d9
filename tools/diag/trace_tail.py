#!/usr/bin/env python3
"""Steady-state view of a rocprofv3 --kernel-trace run: over the LAST n dispatches, per kernel: calls, mean duration, and the
mean idle gap between the previous kernel's end and this kernel's start.  python tools/diag/trace_tail.py <dir> [n=2000]"""
import csv, glob, re, sys
from collections import defaultdict
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:64]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name] += e - s; cnt[name] += 1
    if prev_end is not None:
        gap[name] += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"last {len(rows)} dispatches span {span/1e3:.1f} us; busy {sum(dur.values())/1e3:.1f} us; gaps {sum(gap.values())/1e3:.1f} us")
for k in sorted(dur, key=lambda k: -dur[k]):
    print(f"{k:64s} {cnt[k]:6d}  dur {dur[k]/cnt[k]/1e3:8.2f} us   gap before {gap[k]/cnt[k]/1e3:6.2f} us   total {100*dur[k]/span:5.1f}%")
