#!/bin/bash
# dev aid (round 6): the batched cost / gradient evaluation on the map's side stream (beside the next cycle's inflation
# and z/y pass) against the map's own stream (FUELMI_BATCH_STREAM=0), same box, interleaved
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['stage_ms'], d.get('host_issue_us',{}).get('host_busy_us_per_cycle'))"; }
[ -z "$NOTEST" ] && timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -4
for i in 1 2 3; do for B in 0 1; do
  FUELMI_BATCH_STREAM=$B timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | line "G400 batch_stream=$B"
done; done
for W in G800 G800S "G400 --candidates 256" "G400 --candidates 1"; do for B in 0 1; do
  FUELMI_BATCH_STREAM=$B timeout 200 python bench.py --workload $W --no-cpu-baseline 2>/dev/null | line "$W batch_stream=$B"
done; done
