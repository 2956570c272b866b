"""Kernel timeline of two consecutive plan cycles from the middle of a rocprofv3 --kernel-trace CSV:
cycle_timeline.py <kernel_trace.csv> [anchor-kernel-substring]   (anchor: first kernel of the map chain)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_inflate"
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split('(')[0].replace("void ", "")[:34],
             r.get("Queue_Id")) for r in rows)
idx = [i for i, k in enumerate(ks) if anchor in k[2] and "inflate_x" not in k[2]]
mid = len(idx) // 2
s, e = idx[mid], idx[mid + 2]
t0 = ks[s][0]
qs = sorted({k[3] for k in ks[s:e]})
for k in ks[s:e]:
    print("%8.1f +%6.1f us  q%d %s" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, qs.index(k[3]) + 1, k[2]))
print("two cycles: %.1f us; mean period over the trace: %.1f us" %
      ((ks[e][0] - t0) / 1e3, (ks[idx[-1]][0] - ks[idx[10]][0]) / 1e3 / (len(idx) - 11)))
