#!/bin/bash
# dev aid (round 3): parity tests, serial-stage kernel stats, headline, chain phase stamps, streaming line
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -${TAILN:-25} | tee gpurun_out/r3_tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_serial -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial-stages > /dev/null 2>&1
python - <<PY | tee gpurun_out/r3_kstats.log
import csv
rows=list(csv.reader(open("gpurun_out/q_serial/s_kernel_stats.csv")))
for r in rows[1:16]:
    print("%-28s n %5s avg %8.1f min %8.1f" % (r[0].split("(")[0][-28:], r[1], float(r[3])/1e3, float(r[5])/1e3))
PY
python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'], d.get('frontier_path'))" | tee gpurun_out/r3_bench.log
FUELMI_FR_TIMING=1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep fr-timing | tail -4 | tee gpurun_out/r3_frt.log
python bench.py --workload G800S --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'], d.get('frontier_path'))" | tee gpurun_out/r3_stream.log
python bench.py --workload G800 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'], d.get('frontier_path'))" | tee gpurun_out/r3_g800.log
