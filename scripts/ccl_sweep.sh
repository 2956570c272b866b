#!/bin/bash
# tuning aid: k_ccl_local / k_union time vs tile shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in ${TILES:-7x16 4x8 2x8 8x32 4x32 1x16}; do
  FUELMI_CCL_TILE=$t rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_$t -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial-stages > /dev/null 2>&1
  python - <<PY
import csv
rows=list(csv.reader(open('gpurun_out/sweep_$t/s_kernel_stats.csv')))
d={r[0].split('(')[0]:float(r[3])/1e3 for r in rows[1:]}
print("$t", "ccl_local %.1f union %.1f flatten %.1f claim %.1f" % (d.get('k_ccl_local',0), d.get('k_union',0), d.get('k_flatten',0), d.get('k_claim',0)))
PY
done
