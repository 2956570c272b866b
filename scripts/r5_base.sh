#!/bin/bash
# round-5 dev aid: new regression tests + baseline numbers of this box (G400 / G800, overlapped and serialised)
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_gpu_parity_r5.py -x -q 2>&1 | tail -5
for w in G400 G800; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/r5/base_$w.json 2> gpurun_out/r5/base_$w.err
  python bench.py --workload $w --no-cpu-baseline --serial-stages > gpurun_out/r5/base_${w}_serial.json 2>> gpurun_out/r5/base_$w.err
  python - <<PY
import json
for f in ("gpurun_out/r5/base_$w.json","gpurun_out/r5/base_${w}_serial.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['stage_ms'], d.get('stage_ms_isolated'))
PY
done
