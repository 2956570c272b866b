"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

usage: pmc_summary.py <FETCH counter_collection.csv> <WRITE counter_collection.csv> <out.json>
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in
KiB; on gfx950 FETCH_SIZE reports half of the coalesced read bytes (x2), WRITE_SIZE is accurate.  The
two calibration kernels of this library confirm it on the box: k_state_planes reads N*8 B of f64 log-odds
and the slab fill writes N*8 B."""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].strip()].append(float(r["Counter_Value"]))
    # skip the first launch of every kernel (cold caches / lazy allocation), average the rest
    return {k: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --no-cpu-baseline`",
       "correction": "hbm_bytes = (2 * FETCH_SIZE_KiB + WRITE_SIZE_KiB) * 1024 (gfx950: FETCH_SIZE counts half of the read bytes)",
       "commit": sys.argv[4] if len(sys.argv) > 4 else None,  # git revision of the binary the passes ran
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out["kernels"][k] = {"FETCH_SIZE_KiB_raw": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-34s %10.2f MB" % (k[:34], v["hbm_bytes_per_launch"] / 1e6))
