#!/bin/bash
# dev aid (round 6): k_tile_cross's last workgroup resolving the search (default) against the separate k_resolve
# (FUELMI_FR_FUSE=0), same box, interleaved; full GPU suite first
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['stage_ms'].get('frontier'), d.get('stage_ms_isolated',{}).get('frontier'), d.get('host_issue_us',{}).get('search_end_polling'), d.get('frontier_path'))"; }
[ -z "$NOTEST" ] && timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -4
for i in 1 2 3 4; do for B in 0 1; do
  FUELMI_FR_FUSE=$B timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | line "G400 fuse=$B"
done; done
for W in G800 G800S; do for i in 1 2; do for B in 0 1; do
  FUELMI_FR_FUSE=$B timeout 200 python bench.py --workload $W --no-cpu-baseline 2>/dev/null | line "$W fuse=$B"
done; done; done
for B in 0 1; do
  FUELMI_FR_FUSE=$B FUELMI_FR_TIMING=1 timeout 120 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep "fr-timing" | tail -6 | grep -v "entry avg" | cut -c1-300
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 0 1; do
FUELMI_FR_FUSE=$B rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_serial$B -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial-stages > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.reader(open("gpurun_out/q_serial$B/s_kernel_stats.csv")))
tot=0
for r in rows[1:16]:
    n=r[0].split("(")[0][-28:]
    if any(k in n for k in ("k_tile","k_resolve")):
        tot+=float(r[3])/1e3
        print("fuse=$B %-28s n %5s avg %8.1f min %8.1f" % (n, r[1], float(r[3])/1e3, float(r[5])/1e3))
print("fuse=$B frontier kernels sum %.1f us" % tot)
PY
done
