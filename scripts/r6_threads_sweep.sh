#!/bin/bash
# dev aid (round 6): the map chain sets the period -- do lighter frontier workgroups (256 lanes) / lower stream priority
# leave it more room?  Three-kernel chain (FUELMI_FR_CHAIN=0), same box, two repetitions interleaved.
cd $GRAFT_REPO_ROOT
WL=${1:-G400}
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['stage_ms'])"; }
CFGS=("X=1" "FUELMI_FT_THREADS=256,256,256" "FUELMI_FT_THREADS=256,512,512" "FUELMI_FT_THREADS=512,256,256" "FUELMI_FT_THREADS=512,512,256" \
 "FUELMI_FR_PRIO=normal" "FUELMI_FR_PRIO=low" "FUELMI_FR_PRIO=normal FUELMI_FT_THREADS=256,256,256" "FUELMI_FR_CHAIN=1")
for rep in 1 2; do for cfg in "${CFGS[@]}"; do
  env FUELMI_FR_CHAIN=0 $cfg timeout 200 python bench.py --workload $WL --no-cpu-baseline 2>/dev/null | line "$WL $cfg |"
done; done
