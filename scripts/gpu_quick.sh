#!/bin/bash
# dev aid: GPU parity tests + isolated (serial-stage) kernel timings + overlapped bench line
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_serial -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial-stages > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.reader(open("gpurun_out/q_serial/s_kernel_stats.csv")))
for r in rows[1:${1:-8}]:
    print("%-28s avg %8.1f min %8.1f" % (r[0].split("(")[0][-28:], float(r[3])/1e3, float(r[5])/1e3))
PY
python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'])"
