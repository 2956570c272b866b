#!/bin/bash
# same-box A/B of the plan cycle under tuning hooks: r4_cycle_ab.sh "ENV1=.. ENV2=.." "ENV=.." ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "X=1" "$@"; do
  echo -n "$cfg | "; env $cfg python bench.py --no-cpu-baseline ${WLARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%.0f'%d['value'], d['stage_ms'], d.get('host_issue_us'))"
done; done
