#!/bin/bash
# dev aid (round 6): the one-launch frontier chain (k_tile_chain) against the three kernels it replaces
# (FUELMI_FR_CHAIN=0), same box: parity subset, interleaved headline / streaming / G800 runs, phase stamps.
cd $GRAFT_REPO_ROOT
[ -z "$NOTEST" ] && timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 ${KEXPR:+-k "$KEXPR"} 2>&1 | tail -${TAILN:-6}
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['stage_ms'].get('frontier'), d.get('stage_ms_isolated',{}).get('frontier'), d.get('frontier_path'))"; }
for i in 1 2 3; do
  for C in 0 1; do
    FUELMI_FR_CHAIN=$C timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | line "G400 chain=$C"
  done
done
for C in 0 1; do
  FUELMI_FR_CHAIN=$C timeout 120 python bench.py --workload G800S --no-cpu-baseline 2>/dev/null | line "G800S chain=$C"
  FUELMI_FR_CHAIN=$C timeout 200 python bench.py --workload G800 --no-cpu-baseline 2>/dev/null | line "G800 chain=$C"
done
for C in 0 1; do
  FUELMI_FR_CHAIN=$C FUELMI_FR_TIMING=1 timeout 120 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep "fr-timing" | tail -6 | grep -v "entry avg" | cut -c1-300
done
