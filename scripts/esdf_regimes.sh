#!/bin/bash
# ESDF kernels across distance regimes: plain / FAR pinned, and the adaptive default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$SKIPTEST" ] || python -m pytest tests -m gpu -x -q -k "esdf or sparse or optimistic or golden or map" 2>&1 | tail -3
for far in ${FARS:-0 1 auto}; do
for wl in ${WLS:-G400 G400K G400E G800}; do
  if [ $far = auto ]; then unset FUELMI_ESDF_FAR; else export FUELMI_ESDF_FAR=$far; fi
  python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['stage_ms_isolated']
print('far %-4s %-6s value %7.0f zy %.4f x %.4f' % ('$far', '$wl', d['value'], i['esdf_zy'], i['esdf_x']))"
done; done
