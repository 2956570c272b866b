#!/bin/bash
# tuning aid: ESDF tile shapes on a workload ($1, default G800): isolated stage times
W=${1:-G800}
IFS=";" read -ra LIST <<< "${CFGS:-32 8;48 8;64 8;96 8;128 8}"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  FUELMI_ZY_TILE_KB=$1 FUELMI_X_SEGS=$2 python bench.py --workload $W --no-cpu-baseline --steps 8 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zy_kb=$1 x_segs=$2', round(d['value'],1), d['stage_ms_isolated'])"
done
