"""Summarise a rocprofv3 kernel-trace CSV by (kernel, grid, workgroup): calls, mean/min us, share."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    if name.startswith("void at::") or "rocclr" in name: continue
    agg[(name, r["Grid_Size_X"], r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:62s} grid {int(k[1])//max(1,int(k[2])):6d} x {k[2]:>4s}  n={len(v):5d} mean {sum(v)/len(v):8.1f} us  min {min(v):8.1f}  {100*sum(v)/tot:5.1f}%")
