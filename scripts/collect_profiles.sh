#!/bin/bash
# Round-3 evidence, one gpurun call (COMMIT=<git hash of the code> in the environment stamps every summary): bench lines (headline, big map, sparse regimes, streaming), rocprofv3
# kernel stats of the same commands (overlapped cycle and serial stages), PMC passes (HBM bytes: FETCH_SIZE and
# WRITE_SIZE in separate passes; SQ issue / wait cycles in a third), next-row timings, facade bench, fleet test.
# Everything lands under gpurun_out/prof_r02; scripts/publish_profiles.sh copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${R:-r03}
export FUELMI_COMMIT=${COMMIT:-unknown}
O=gpurun_out/prof_$R
mkdir -p $O
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
for WL in G400 G800; do
  ST=6; [ $WL = G800 ] && ST=4
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --steps $ST --warmup 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --steps $ST --warmup 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --serial-stages --steps $ST --warmup 2 > /dev/null 2> $O/pmc_sq_$WL.err
  python scripts/pmc_summary.py $O/pmc_fetch_$WL/s_counter_collection.csv $O/pmc_write_$WL/s_counter_collection.csv $O/pmc_hbm_traffic_$WL.json $FUELMI_COMMIT > /dev/null
  cp $O/pmc_hbm_traffic_$WL.json profiles/${R}_pmc_hbm_traffic_$WL.json   # bench.py reads roofline.traffic from here
  python scripts/pmc_sq_summary.py $O/pmc_sq_$WL/s_counter_collection.csv $O/pmc_sq_$WL.json > /dev/null 2>> $O/pmc_sq_$WL.err
done
timeout 300 python bench.py > $O/bench_G400.json 2> $O/bench_G400.err
timeout 300 python bench.py --workload G800 > $O/bench_G800.json 2> $O/bench_G800.err
for WL in G400K G400E; do timeout 300 python bench.py --workload $WL --no-cpu-baseline > $O/bench_$WL.json 2>/dev/null; done
timeout 300 python bench.py --workload G800S > $O/bench_G800S.json 2> $O/bench_G800S.err
timeout 300 python bench.py --no-cpu-baseline --candidates 1 > $O/bench_G400_C1.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --candidates 256 > $O/bench_G400_C256.json 2>/dev/null
for WL in G400 G800; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cycle_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline > $O/bench_${WL}_under_rocprof.json 2>/dev/null
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --serial-stages > /dev/null 2>&1
done
for WL in G400K G400E; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --serial-stages --steps 20 --warmup 3 > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stream -o s -- python bench.py --workload G800S --no-cpu-baseline > /dev/null 2>&1
timeout 300 python scripts/bench_next.py > $O/next_rows.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/next -o s -- python scripts/bench_next.py > /dev/null 2>&1
timeout 300 python scripts/facade_bench.py --map G800S --frames 30 > $O/facade_bench_G800S.json 2> $O/facade_bench.err
timeout 300 python scripts/facade_bench.py --fullbox G400 > $O/facade_bench_G400_fullbox.json 2>> $O/facade_bench.err
for RO in 1 2; do timeout 300 python bench.py --workload G800S --no-cpu-baseline --reference-order $RO > $O/bench_G800S_reforder$RO.json 2>/dev/null; done
timeout 300 python bench.py --no-cpu-baseline --reference-order 1 --steps 20 --warmup 3 > $O/bench_G400_reforder1.json 2>/dev/null
timeout 300 python -m pytest tests/test_fleet_gpu.py -q -s -m gpu 2>&1 | grep -E "fleet on one device|passed|failed" > $O/fleet_one_device.txt
for f in bench_G400 bench_G800 bench_G400K bench_G400E bench_G800S; do tail -1 $O/$f.json | cut -c1-160; done
cat $O/fleet_one_device.txt; tail -1 $O/facade_bench_G800S.json | cut -c1-300
find $O -name "*.csv" | wc -l
