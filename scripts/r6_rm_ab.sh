#!/bin/bash
# dev aid (round 6): the changed-cluster test as ONE launch with an in-kernel barrier (default) against two launches (FUELMI_RM_TWO_PASS=1)
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['stage_ms'])"; }
[ -z "$NOTEST" ] && timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -4
for i in 1 2 3 4; do
  FUELMI_RM_TWO_PASS=1 timeout 200 python bench.py --workload G800S --no-cpu-baseline 2>/dev/null | line "G800S two launches"
  timeout 200 python bench.py --workload G800S --no-cpu-baseline 2>/dev/null | line "G800S one launch"
done
