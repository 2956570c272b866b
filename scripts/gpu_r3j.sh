#!/bin/bash
# dev aid (round 3): reference-order sweep in LDS -- parity, then its price per search
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -${TAILN:-6}
for RO in 0 1; do
  python bench.py --workload G800S --no-cpu-baseline --reference-order $RO 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G800S ro=$RO', d['value'], d['ms_per_step'], d['stage_ms'])"
done
python bench.py --no-cpu-baseline --reference-order 1 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G400 ro=1', d['value'], d['ms_per_step'], d['stage_ms'])"
python scripts/facade_bench.py 2>&1 | tail -4
