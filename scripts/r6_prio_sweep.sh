#!/bin/bash
# dev aid (round 6): frontier stream priority and k_tile_out's workgroup size on top of the fused cross + resolve, same box
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['stage_ms'])"; }
CFGS=("X=1" "FUELMI_FT_THREADS=512,512,256" "FUELMI_FR_PRIO=normal" "FUELMI_FR_PRIO=low" "FUELMI_FR_PRIO=normal FUELMI_FT_THREADS=512,512,256")
for W in G400 G800S G800; do
for rep in 1 2 3 4; do for cfg in "${CFGS[@]}"; do
  env $cfg timeout 200 python bench.py --workload $W --no-cpu-baseline 2>/dev/null | line "$W $cfg |"
done; done; done
