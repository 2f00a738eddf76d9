#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs) into the JSON bench.py reads for
roofline.traffic:   pmc_summary.py <out.json> <workload>=<fetch_dir>,<write_dir>[,<command>] ...
Every row carries the workload it was collected on; bench.py takes `traffic` only from a row of the workload it timed.
Correction (MI355X_MICROARCH.md, HBM section): the counters are in KB and on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, averaged per launch of each kernel.  With `steady` in a
workload's name only the second half of each kernel's dispatches is averaged (the first half fills the cache); with `tail`
the last 5 % (a growing cache at its final length)."""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter, steady, tail=False):
    seq = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                seq[r["Kernel_Name"]].append((int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"])))
    out = {}
    for k, v in seq.items():
        v.sort()
        vals = [x for _, x in v]
        if steady and len(vals) >= 4:
            vals = vals[len(vals) // 2:]
        if tail and len(vals) >= 40:                         # a growing cache: the last 5 % of the dispatches (the cache at its final length)
            vals = vals[-(len(vals) // 20):]
        out[k] = (len(vals), sum(vals) / len(vals))
    return out


def main():
    out_path, rows, cmds = sys.argv[1], [], {}
    for spec in sys.argv[2:]:
        wl, rest = spec.split("=", 1)
        parts = rest.split(",", 2)
        cmds[wl] = parts[2] if len(parts) > 2 else ""
        steady, tail = "steady" in wl, "tail" in wl
        fetch, write = per_kernel(parts[0], "FETCH_SIZE", steady, tail), per_kernel(parts[1], "WRITE_SIZE", steady, tail)
        for k, (n, f_avg) in fetch.items():
            wn, w_avg = write.get(k, (0, 0.0))
            rows.append({"workload": wl, "kernel": k if len(k) < 160 else k[:157] + "...", "dispatches": n, "FETCH_SIZE_KB_avg": round(f_avg, 1),
                         "WRITE_SIZE_KB_avg": round(w_avg, 1), "hbm_bytes_per_launch": int((2 * f_avg + w_avg) * 1024),
                         "_total": (2 * f_avg + w_avg) * n})
    rows.sort(key=lambda r: (r["workload"], -r["_total"]))
    kept = []
    for wl in cmds:
        kept += [r for r in rows if r["workload"] == wl][:24]
    for r in kept:
        del r["_total"]
    out = {"commands": cmds, "passes": "FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs of each command (no tracing flags)",
           "correction": "MI355X_MICROARCH.md HBM section: counters are KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
           "kernels": kept}
    json.dump(out, open(out_path, "w"), indent=1)
    for r in kept:
        print(f"{r['workload']:22s} {r['kernel'][:64]:64s} n={r['dispatches']:5d} {r['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch")


if __name__ == "__main__":
    main()
