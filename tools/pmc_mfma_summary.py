#!/usr/bin/env python3
"""Per-kernel means of an SQ/GRBM counter pass (rocprofv3 --pmc ... csv) + MFMA utilisation:
    pmc_mfma_summary.py <pmc_dir> <out.json>
util = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 4 SIMDs * 256 CUs): the gfx94x MfmaUtil formula (gfx950 has
no derived-counter section in ROCm 7.2, MI355X_MICROARCH.md "rocprofv3 PMC slots") with GRBM_GUI_ACTIVE divided by 8 because
rocprofv3 reports it summed over the 8 XCDs (checked: SQ_VALU_MFMA_BUSY_CYCLES is exactly 16 cycles per issued
v_mfma_f32_16x16x32_bf16, and busy / (trace duration * 2.4 GHz * 1024) reproduces flops / time / 2.5 PF).  The kernels run
~15-25 % slower under counter collection, so this utilisation reads lower than the one computed from the trace durations."""
import collections, csv, glob, json, os, sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if name.startswith("void at::") or "rocclr" in name:
            continue
        a = acc[(name.split("(")[0], r["Grid_Size"], r["Workgroup_Size"])][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
rows = []
for (name, grid, wg), c in acc.items():
    m = {k: v[1] / v[0] for k, v in c.items()}
    n = max(v[0] for v in c.values())
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    row = {"kernel": name[-70:], "workgroups": int(grid) // max(1, int(wg)), "dispatches": n, **{k: round(v, 1) for k, v in m.items()}}
    if gui:
        row["mfma_util"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 4 * 256), 4)
        if m.get("SQ_WAVE_CYCLES"):
            row["wave_parked_frac"] = round(m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 3)
            row["issue_stall_frac"] = round(m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 3)
    rows.append(row)
if not rows:
    sys.exit(f"pmc_mfma_summary: no counter rows under {sys.argv[1]} - nothing written to {sys.argv[2]} (a claimed measurement needs its table)")
rows.sort(key=lambda r: -r.get("GRBM_GUI_ACTIVE", 0) * r["dispatches"])
json.dump({"formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); *_frac relative to SQ_WAVE_CYCLES",
           "kernels": rows[:16]}, open(sys.argv[2], "w"), indent=1)
for r in rows[:10]:
    print(f"{r['kernel'][:52]:52s} wgs {r['workgroups']:5d} n={r['dispatches']:4d} mfma_util {r.get('mfma_util')}  parked {r.get('wave_parked_frac')}  issue-stall {r.get('issue_stall_frac')}")
