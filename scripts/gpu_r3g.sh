#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -${TAILN:-6} | cut -c1-300
for i in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side stream', d['value'], d['stage_ms'])"
FUELMI_BSPLINE_INLINE=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inline     ', d['value'], d['stage_ms'])"
done
python bench.py --no-cpu-baseline --serial-stages 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial     ', d['value'], d['stage_ms_isolated'])"
