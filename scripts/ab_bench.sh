#!/bin/bash
# dev aid: A/B of two builds of libfuelmi on ONE box (boxes of the pool differ by ~10 %): copy the previous build to
# fuel_amd/libfuelmi_prev.so, rebuild, then `gpurun -- 'bash scripts/ab_bench.sh'` -- a parity subset (KEXPR), three
# interleaved headline runs per build, the tile-CCL phase stamps and one streaming run per build.  ~1 GPU-minute.
timeout 300 python -m pytest tests -m gpu -x -q --timeout 120 -k "${KEXPR:-frontier or cycle or config4}" 2>&1 | tail -4
for i in 1 2 3; do
  for L in prev new; do
    P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi.so; [ $L = prev ] && P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi_prev.so
    FUELMI_LIB_PATH=$P timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']), d['stage_ms']['frontier'], d['stage_ms_isolated']['frontier'])"
  done
done
for L in prev new; do
  P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi.so; [ $L = prev ] && P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi_prev.so
  FUELMI_LIB_PATH=$P FUELMI_FR_TIMING=1 timeout 120 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep "fr-timing\] tiles" | tail -1 | cut -c1-200
  FUELMI_LIB_PATH=$P timeout 120 python bench.py --workload G800S --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L G800S', round(d['value']), d['stage_ms']['frontier'])"
done
for L in prev new; do
  P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi.so; [ $L = prev ] && P=$GRAFT_REPO_ROOT/fuel_amd/libfuelmi_prev.so
  FUELMI_LIB_PATH=$P FUELMI_FR_TIMING=1 timeout 120 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep "fr-timing\] resolve" | tail -1 | cut -c1-200
done
