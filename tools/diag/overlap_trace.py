#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: time covered by vision-tower kernels, by LM kernels, and by both at once, per queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
TOWER = ("gemm_tile", "attn_dense", "attn_head64", "layernorm", "im2col", "pool_kernel", "gather_pool")
LM = ("gemm_ws", "gemm_wl", "resid_norm", "qkv_finish", "rmsnorm", "heads_kernel", "attn_fwd", "attn_lm", "attn_combine")
ev = []
qs = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    fam = "T" if any(k in n for k in TOWER) else ("L" if any(k in n for k in LM) else None)
    if not fam: continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    qs[(fam, r.get("Queue_Id", "?"))] += 1
    ev.append((s, 1, fam)); ev.append((e, -1, fam))
ev.sort()
act = {"T": 0, "L": 0}
cov = {"T": 0, "L": 0, "both": 0}
last = ev[0][0]
for t, d, fam in ev:
    dt = t - last
    if act["T"] > 0: cov["T"] += dt
    if act["L"] > 0: cov["L"] += dt
    if act["T"] > 0 and act["L"] > 0: cov["both"] += dt
    act[fam] += d; last = t
print("kernels per (family, queue):", dict(qs))
print({k: round(v / 1e6, 2) for k, v in cov.items()}, "ms covered (tower / LM / both at once)")
# per background encode (tower kernels on a queue of their own): span, kernel time, LM kernel time inside the span
tq = [q for (f, q), c in qs.items() if f == "T"]
lq = max(((c, q) for (f, q), c in qs.items() if f == "L"), default=(0, None))[1]
bgq = [q for q in tq if q != lq and qs[("T", q)] < max(qs[("T", x)] for x in tq)] or tq
T = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows
           if r.get("Queue_Id") in bgq and any(k in r["Kernel_Name"] for k in TOWER))
L = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r.get("Queue_Id") == lq and any(k in r["Kernel_Name"] for k in LM))
groups, cur = [], []
for s, e, n in T:
    if cur and s - cur[-1][1] > 2_000_000: groups.append(cur); cur = []
    cur.append((s, e, n))
if cur: groups.append(cur)
for g in groups:
    s0, e0 = g[0][0], g[-1][1]
    kt = sum(e - s for s, e, _ in g)
    lin = sum(min(e, e0) - max(s, s0) for s, e in L if e > s0 and s < e0)
    nl = sum(1 for s, e in L if e > s0 and s < e0)
    gemm = [e - s for s, e, n in g if "gemm_tile" in n]
    print(f"encode on queue {bgq}: {len(g)} kernels, span {(e0 - s0) / 1e6:.2f} ms, tower kernel time {kt / 1e6:.2f} ms, "
          f"LM kernel time inside {lin / 1e6:.2f} ms over {nl} launches; tile GEMM avg {sum(gemm) / max(len(gemm), 1) / 1e3:.1f} us")
# LM step time: gate/up launches inside vs outside the encodes
def inside(t): return any(g[0][0] <= t <= g[-1][1] for g in groups)
gu = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "gemm_ws_kernel<3, 2" in r["Kernel_Name"] and r.get("Queue_Id") == lq]
a = [e - s for s, e in gu if inside(s)]; b = [e - s for s, e in gu if not inside(s)]
if a and b: print(f"gate/up launch: {sum(a) / len(a) / 1e3:.1f} us under an encode ({len(a)}), {sum(b) / len(b) / 1e3:.1f} us otherwise ({len(b)})")
