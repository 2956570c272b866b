#!/bin/bash
# dev aid (round 3): workgroup sizes of the tile kernels
for NTS in 512,512,512 512,256,256 256,256,256 512,256,512; do
  echo "== FT_THREADS=$NTS"
  FUELMI_FT_THREADS=$NTS FUELMI_FR_TIMING=1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep fr-timing | tail -5 | cut -c1-330
  FUELMI_FT_THREADS=$NTS python bench.py --no-cpu-baseline --steps 50 --warmup 5 --serial-stages | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms_isolated'])"
  FUELMI_FT_THREADS=$NTS python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'], d.get('frontier_path'))"
done
