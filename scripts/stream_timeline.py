"""Kernel + copy timeline of the last complete streaming step (anchor: k_project_depth) from rocprofv3 CSVs."""
import csv, sys
d = sys.argv[1]
ev = []
for r in csv.DictReader(open(d + "/s_kernel_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split('(')[0][-24:]))
try:
    for r in csv.DictReader(open(d + "/s_memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[-14:]))
except FileNotFoundError:
    pass
ev.sort()
idx = [i for i, k in enumerate(ev) if k[2].endswith("k_project_depth")]
s, e = idx[-2], idx[-1]
t0 = ev[s][0]
for k in ev[s:e]:
    print("%8.1f +%7.1f us  %s" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[2]))
print("step len", (ev[e][0] - t0) / 1e3)
