#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel."""
import re, sys, subprocess
txt = sys.stdin.read()
cur = None; rows = []
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
    if "error" in line: print(line)
for r in rows:
    try: name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception: name = r["name"]
    name = re.sub(r"\(.*", "", name)[:70]
    print(f"{name:70s} vgpr={r.get('vgpr')} spill={r.get('spill')} occ={r.get('occ')} sgpr={r.get('sgpr')} lds={r.get('lds')}")
