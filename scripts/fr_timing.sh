#!/bin/bash
# dev aid: full GPU parity suite, then the frontier chain's in-kernel phase stamps (FUELMI_FR_TIMING), its kernels'
# serialised rocprofv3 durations and the headline / streaming figures of the same box
[ -z "$NOTEST" ] && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -${TAILN:-6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FUELMI_FR_TIMING=1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep fr-timing | tail -7 | grep -v "entry avg" | cut -c1-330
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_serial -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial-stages > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.reader(open("gpurun_out/q_serial/s_kernel_stats.csv")))
tot=0
for r in rows[1:16]:
    n=r[0].split("(")[0][-28:]
    if any(k in n for k in ("k_tile","k_resolve","k_pred3")):
        tot+=float(r[3])/1e3
        print("%-28s n %5s avg %8.1f min %8.1f" % (n, r[1], float(r[3])/1e3, float(r[5])/1e3))
print("frontier kernels sum %.1f us" % tot)
PY
python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'], d['stage_ms_isolated'])"
echo "== streaming"
FUELMI_FR_TIMING=1 python bench.py --workload G800S --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep fr-timing | tail -7 | grep -v "entry avg" | cut -c1-330
python bench.py --workload G800S --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'], d['stage_ms_isolated'])"
