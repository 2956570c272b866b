#!/bin/bash
# dev aid (round 6): is the plan cycle bound by LDS-occupancy x time?  libfuelmi_prev.so (FR_TCELL 4096: three tile
# workgroups per CU) against libfuelmi.so (2048: four), each with the one-launch chain on and off, same box.
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['stage_ms'], d.get('frontier_path'))"; }
for L in prev new; do
  P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi.so; [ $L = prev ] && P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi_prev.so
  FUELMI_LIB_PATH=$P timeout 600 python -m pytest tests -m gpu -x -q --timeout 300 -k "frontier or cycle or config4 or golden or pillar" 2>&1 | tail -3
done
for i in 1 2 3; do
  for L in prev new; do
    P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi.so; [ $L = prev ] && P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi_prev.so
    for C in 0 1; do
      FUELMI_LIB_PATH=$P FUELMI_FR_CHAIN=$C timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | line "G400 $L chain=$C"
    done
  done
done
for L in prev new; do
  P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi.so; [ $L = prev ] && P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi_prev.so
  for C in 0 1; do
    FUELMI_LIB_PATH=$P FUELMI_FR_CHAIN=$C timeout 120 python bench.py --workload G800S --no-cpu-baseline 2>/dev/null | line "G800S $L chain=$C"
    FUELMI_LIB_PATH=$P FUELMI_FR_CHAIN=$C timeout 200 python bench.py --workload G800 --no-cpu-baseline 2>/dev/null | line "G800 $L chain=$C"
  done
done
