#!/usr/bin/env python3
"""Per-position kernel durations of the vision tower's per-layer launch sequence from a rocprofv3 --kernel-trace CSV: the last `encodes`
encodes are cut at their im2col launches, the launches between the first two LayerNorms of every layer are aligned by position, and the median
duration per position is printed (LN1, QKV, attention, out-proj, LN2, fc1, fc2).   python tools/diag/trace_layer_seq.py TRACE.csv [encodes]"""
import csv, statistics, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n_enc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
starts = [i for i, r in enumerate(rows) if r[2].startswith("im2col_norm_kernel")]
starts = starts[-n_enc:]
per_pos = {}
tot = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    enc = [r for r in rows[a:b] if not r[2].startswith("void at::") and "rocclr" not in r[2]]
    ln = [i for i, r in enumerate(enc) if "layernorm_kernel" in r[2]]
    tot.append((enc[-1][1] - enc[0][0]) / 1e3)
    for j in range(0, len(ln) - 1, 2):            # a layer = from its LN1 up to (not including) the next layer's LN1
        lo, hi = ln[j], (ln[j + 2] if j + 2 < len(ln) else ln[j] + 7)
        for pos, r in enumerate(enc[lo:hi]):
            per_pos.setdefault((pos, r[2].split("(")[0][-44:]), []).append((r[1] - r[0]) / 1e3)
    head = enc[:ln[0]] if ln else enc
    tail = enc[ln[-1] + 3:] if ln else []
    for k, r in enumerate(head):
        per_pos.setdefault((-10 + k, "head " + r[2].split("(")[0][-40:]), []).append((r[1] - r[0]) / 1e3)
    for k, r in enumerate(tail):
        per_pos.setdefault((100 + k, "tail " + r[2].split("(")[0][-40:]), []).append((r[1] - r[0]) / 1e3)
print(f"{len(starts)} encodes; span per encode (first kernel start to last kernel end): " + " ".join(f"{t:.0f}" for t in tot) + " us")
layer_sum = 0.0
for (pos, name), v in sorted(per_pos.items()):
    m = statistics.median(v)
    if 0 <= pos < 100:
        layer_sum += m
    print(f"  pos {pos:4d} {name:50s} n={len(v):4d} median {m:7.2f} us  min {min(v):7.2f}  max {max(v):7.2f}")
print(f"sum of per-position medians inside a layer: {layer_sum:.1f} us")
