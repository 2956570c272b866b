#!/bin/bash
# dev aid (round 3): parity + chain phase stamps (stages serialised: no interference from the map stream) + kernel stats
[ -z "$NOTEST" ] && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -${TAILN:-8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for T in ${TILES:-"" 8x16}; do
  echo "== FTILE=$T"
  FUELMI_FTILE=$T FUELMI_FR_TIMING=1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep fr-timing | tail -8 | cut -c1-330
  FUELMI_FTILE=$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_serial -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial-stages > /dev/null 2>&1
  python - <<PY
import csv
rows=list(csv.reader(open("gpurun_out/q_serial/s_kernel_stats.csv")))
for r in rows[1:14]:
    n=r[0].split("(")[0][-28:]
    if any(k in n for k in ("k_tile","k_resolve","k_pred3","k_esdf","k_inflate","k_bspline")):
        print("%-28s n %5s avg %8.1f min %8.1f" % (n, r[1], float(r[3])/1e3, float(r[5])/1e3))
PY
  FUELMI_FTILE=$T python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'], d['stage_ms_isolated'])"
done
