#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in ${TILES:-7x16 8x16 12x16 16x16 8x32 16x32}; do
  for wl in ${WLS:-G400}; do
    FUELMI_FTILE=$t python bench.py --workload $wl --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$t $wl', d.get('frontier_path'), 'value %7.0f iso fr %.4f cyc fr %.4f' % (d['value'], d['stage_ms_isolated']['frontier'], d['stage_ms']['frontier']))"
  done
done
