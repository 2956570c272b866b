#!/bin/bash
timeout 200 python -m pytest tests -m gpu -x -q --timeout 90 -k "reference_order or golden or facade" 2>&1 | tail -${TAILN:-6} || exit 1
timeout 400 python -m pytest tests -m gpu -x -q --timeout 120 2>&1 | tail -${TAILN:-6}
FUELMI_FR_TIMING=1 timeout 120 python bench.py --workload G800S --no-cpu-baseline --steps 12 --warmup 2 --reference-order 1 2>&1 | grep "reference order" | tail -14
for RO in 0 2; do
timeout 120 python bench.py --workload G800S --no-cpu-baseline --reference-order $RO 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G800S ro=$RO', d['value'], d['ms_per_step'], d['stage_ms'])"
done
timeout 200 python scripts/facade_bench.py 2>&1 | tail -3
