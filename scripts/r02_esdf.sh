#!/bin/bash
# ESDF kernel variants: isolated stage times on G400 / G800 under tuning env switches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # label, workload, env...
  local label=$1 wl=$2; shift 2
  env "$@" python bench.py --workload $wl --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d.get('frontier_path'), '%-28s %-5s value %7.0f  iso zy %.4f x %.4f infl %.4f fr %.4f | cyc zy %.4f x %.4f' % ('$label', '$wl', d['value'], d['stage_ms_isolated']['esdf_zy'], d['stage_ms_isolated']['esdf_x'], d['stage_ms_isolated']['inflate'], d['stage_ms_isolated']['frontier'], d['stage_ms']['esdf_zy'], d['stage_ms']['esdf_x']))"
}
for wl in ${WLS:-G400 G800}; do
  run nopipe $wl FUELMI_X_NOPIPE=1
  run pipe-default $wl A=1
  run pipe-grid1 $wl FUELMI_X_GRID=1
  run pipe-grid2 $wl FUELMI_X_GRID=2
done
