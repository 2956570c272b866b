"""Print the kernel timeline of the last full cycle from a rocprofv3 --kernel-trace CSV (dev aid)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split('(')[0][-22:], r.get("Queue_Id")) for r in rows)
idx = [i for i, k in enumerate(ks) if k[2].endswith('k_box_and')]
s, e = idx[-2], idx[-1]
t0 = ks[s][0]
for k in ks[s:e]:
    print("%8.1f +%7.1f us  q%s %s" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2]))
print("cycle len", (ks[e][0] - t0) / 1e3)
