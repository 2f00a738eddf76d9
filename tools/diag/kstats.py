#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats csv: name (short), calls, avg us, total ms, %.  python tools/diag/kstats.py <dir>"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    name = re.sub(r"\(.*", "", r["Name"])[:70]
    print(f"{name:70s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.2f} us {float(r['TotalDurationNs'])/1e6:9.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}%")
