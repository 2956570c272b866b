#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in two default; do
  E="X=1"; [ $V = two ] && E="FUELMI_RM_TWO_PASS=1"
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rm_$V -o s -- python bench.py --workload G800S --steps 40 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv
rows=list(csv.reader(open("gpurun_out/rm_$V/s_kernel_stats.csv")))
for r in rows[1:30]:
    n=r[0].split("(")[0][-30:]
    if any(k in n for k in ("k_rm","k_pool","k_tile","k_resolve","k_insert")):
        print("$V %-30s n %5s avg %8.1f min %8.1f max %8.1f" % (n, r[1], float(r[3])/1e3, float(r[5])/1e3, float(r[6])/1e3))
PY
done
