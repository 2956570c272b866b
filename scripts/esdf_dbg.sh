#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 3 7; do
  FUELMI_ESDF_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ed_$d -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial-stages > /dev/null 2>&1
  python - <<PY
import csv
rows=list(csv.reader(open('gpurun_out/ed_$d/s_kernel_stats.csv')))
d={r[0].split('(')[0]:(float(r[3])/1e3,float(r[5])/1e3) for r in rows[1:]}
k=[x for x in d if 'k_esdf_zy4' in x][0]
print("dbg=$d zy4 avg %.1f min %.1f" % d[k])
PY
done
