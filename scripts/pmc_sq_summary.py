"""Summarise a rocprofv3 --pmc pass of SQ counters into per-kernel issue / wait shares.

usage: pmc_sq_summary.py <counter_collection.csv> <out.json>
Per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAIT_ANY (wave parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY
(issue stall) + SQ_ACTIVE_INST_ANY (issuing) ~= SQ_WAVE_CYCLES, summed over the waves of the dispatch.  Shares are
fractions of SQ_WAVE_CYCLES; valu_share_of_issue = SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_ANY."""
import csv, json, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0].strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --pmc SQ_* (own pass) of `python bench.py --no-cpu-baseline --serial-stages`", "kernels": {}}
for k, c in sorted(acc.items()):
    m = {n: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    e = {"launches": len(next(iter(c.values()))), "waves": round(m.get("SQ_WAVES", 0.0)),
         "wave_cycles": round(wc)}
    for n, key in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_ANY", "parked_waitcnt_or_barrier"),
                   ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_ACTIVE_INST_VALU", "valu"), ("SQ_ACTIVE_INST_LDS", "lds")):
        if n in m:
            e[key + "_share"] = round(m[n] / wc, 4)
    if m.get("SQ_ACTIVE_INST_ANY"):
        e["valu_share_of_issue"] = round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / m["SQ_ACTIVE_INST_ANY"], 4)
    if m.get("SQ_BUSY_CYCLES"):
        e["busy_cycles"] = round(m["SQ_BUSY_CYCLES"])
    out["kernels"][k] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, e in out["kernels"].items():
    print("%-34s" % k[:34], {a: b for a, b in e.items() if a.endswith("share") or a == "valu_share_of_issue"})
