#!/bin/bash
# the two timing files of scripts/collect_profiles.sh on their own (same commands)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FUELMI_COMMIT=${COMMIT:-unknown}
O=gpurun_out/prof_r04x; mkdir -p $O
# where a streaming frame goes (host wall clock per call group, device timeline from events) and what the reference's cell
# order costs on the full box (the level sweeps' own clock; workgroup sizes of the large-cluster sweep; x pass phase stamps)
{ echo "# FUELMI_STREAM_TIMING=2 python bench.py --workload G800S --no-cpu-baseline (one line pair per bench_stream call: warm-up, timed, frame sources), commit $FUELMI_COMMIT"
  FUELMI_STREAM_TIMING=2 python bench.py --workload G800S --no-cpu-baseline 2>&1 >/dev/null | grep stream-timing
  echo "# default order (search bookkeeping, then the map chain) against FUELMI_STREAM_MAP_FIRST=1, frames/s, alternating"
  for V in 0 1 0 1; do if [ $V = 1 ]; then export FUELMI_STREAM_MAP_FIRST=1; else unset FUELMI_STREAM_MAP_FIRST; fi
    python bench.py --workload G800S --no-cpu-baseline | V=$V python -c "import sys,json,os; print('map_first=%s: %d frames/s' % (os.environ['V'], round(json.loads(sys.stdin.readline())['value'])))"; done
  unset FUELMI_STREAM_MAP_FIRST; } > $O/stream_frame_timing.txt 2>&1
{ echo "# FUELMI_FR_TIMING=1 python bench.py --no-cpu-baseline --reference-order 1 --steps 10 --warmup 3 (400x400x100, full box: one cluster of 139 k cells), commit $FUELMI_COMMIT"
  for T in 512 256 1024; do echo "# k_bfs_sweep_g with $T threads"; FUELMI_BFSG_T=$T FUELMI_FR_TIMING=1 python bench.py --no-cpu-baseline --reference-order 1 --steps 10 --warmup 3 2>&1 >/dev/null | grep "reference order" | tail -2; done
  echo "# x pass phase stamps (FUELMI_ZY_TIMING=1, scripts/esdf_only.py)"
  for WL in G400 G800; do FUELMI_ZY_TIMING=1 python scripts/esdf_only.py $WL 0 3 2>&1 | grep x-timing | tail -1 | sed "s/^/$WL /"; done; } > $O/reference_order_timing.txt 2>&1
cat $O/stream_frame_timing.txt $O/reference_order_timing.txt
