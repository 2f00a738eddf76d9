#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/gpu_round.sh into the JSON that
bench.py reads for roofline.traffic:  pmc_summary.py <fetch_dir> <write_dir> <out.json>
Correction (MI355X_MICROARCH.md, HBM section): the counters are in KB and on gfx950 FETCH_SIZE reports half the bytes
of wide coalesced reads, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, averaged per launch of each kernel."""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])          # dispatches, sum, max
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                a[2] = max(a[2], float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k, (n, tot, fmax) in fetch.items():
        wn, wtot, wmax = write.get(k, [0, 0.0, 0.0])
        f_avg, w_avg = tot / n, (wtot / wn if wn else 0.0)
        rows.append({"kernel": k if len(k) < 160 else k[:157] + "...", "dispatches": n, "FETCH_SIZE_KB_avg": round(f_avg, 1),
                     "WRITE_SIZE_KB_avg": round(w_avg, 1), "hbm_bytes_per_launch": int((2 * f_avg + w_avg) * 1024),
                     # the largest dispatch: kernels whose work grows with the cache (attention, re-rotation) reach their steady state there
                     "hbm_bytes_per_launch_max": int((2 * fmax + wmax) * 1024),
                     "_total": (2 * f_avg + w_avg) * n})
    rows.sort(key=lambda r: -r["_total"])
    for r in rows:
        del r["_total"]
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (two separate passes) --output-format csv -- python3 bench.py --steps 1 --warmup 0 --frames 4 --no-cpu-baseline (secondary data on: steady-state sink stream and 8-stream step included)",
           "correction": "MI355X_MICROARCH.md HBM section: counters are KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
           "kernels": rows[:40]}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for r in rows[:8]:
        print(f"{r['kernel'][:70]:70s} n={r['dispatches']:5d} {r['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
