#!/bin/bash
for T in default 8x32 16x16 8x16; do
  FUELMI_FTILE=$T python bench.py --workload G800 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T', round(d['value'],1), d['stage_ms'], d['stage_ms_isolated']['frontier'])"
done
