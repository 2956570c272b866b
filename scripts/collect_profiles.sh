#!/bin/bash
# Round-6 evidence (the round-5 script, brought forward), one gpurun call (COMMIT=<git hash of the code> in the environment stamps every summary): bench lines
# (headline, big map, sparse regimes, streaming, fleet of two on one device), rocprofv3 kernel stats of the same commands
# (overlapped cycle and serial stages), PMC passes (HBM bytes: FETCH_SIZE and WRITE_SIZE in separate passes; SQ issue /
# wait cycles in a third; SQ instruction counts of the ESDF families in a fourth), cycle timelines, in-kernel phase
# stamps of the z/y pass, same-box A/B runs of this round's switches, next-row timings, facade bench, fleet / perf tests.
# Everything lands under gpurun_out/prof_$R; scripts/publish_profiles.sh $R copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${R:-r06}
export FUELMI_COMMIT=${COMMIT:-unknown}
O=gpurun_out/prof_$R
rm -rf $O; mkdir -p $O
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQI="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
for WL in G400 G800; do
  ST=6; [ $WL = G800 ] && ST=4
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --steps $ST --warmup 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --steps $ST --warmup 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --serial-stages --steps $ST --warmup 2 > /dev/null 2> $O/pmc_sq_$WL.err
  python scripts/pmc_summary.py $O/pmc_fetch_$WL/s_counter_collection.csv $O/pmc_write_$WL/s_counter_collection.csv $O/pmc_hbm_traffic_$WL.json $FUELMI_COMMIT > /dev/null
  cp $O/pmc_hbm_traffic_$WL.json profiles/${R}_pmc_hbm_traffic_$WL.json   # bench.py reads roofline.traffic from here
  python scripts/pmc_sq_summary.py $O/pmc_sq_$WL/s_counter_collection.csv $O/pmc_sq_$WL.json > /dev/null 2>> $O/pmc_sq_$WL.err
done
# instruction counts of the z/y and x passes, packed family (0) and 32-bit family (2), ESDF updates only
for FAM in 0 2; do
  timeout 300 rocprofv3 --pmc $SQI --output-format csv -d $O/pmc_insts_fam$FAM -o s -- python scripts/esdf_only.py G400 $FAM 4 > /dev/null 2>&1
done
O=$O python - > $O/esdf_instruction_counts.txt <<'PY'
import csv, glob, collections, os
print("# SQ_INSTS_* per launch (wave instructions, whole dispatch) of the ESDF kernels on the 400x400x100 map, rocprofv3 --pmc, own pass;")
print("# family 0 = packed 16-bit kernels with the 16-bit hand-over (k_esdf_zy_pk2, k_esdf_x_pk2), family 2 = 32-bit kernels (k_esdf_zy4, k_esdf_x4h)")
for fam in (0, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.environ["O"] + "/pmc_insts_fam%d/**/*counter_collection.csv" % fam, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(acc.items()):
        if "esdf" in k:
            print("family", fam, k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
timeout 300 python bench.py > $O/bench_G400.json 2> $O/bench_G400.err
timeout 300 python bench.py --workload G800 > $O/bench_G800.json 2> $O/bench_G800.err
for WL in G400K G400E; do timeout 300 python bench.py --workload $WL --no-cpu-baseline > $O/bench_$WL.json 2>/dev/null; done
timeout 300 python bench.py --workload G800S > $O/bench_G800S.json 2> $O/bench_G800S.err
timeout 300 python bench.py --workload G800S --no-cpu-baseline --steps 100 --warmup 10 > $O/bench_G800S_100steps.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --candidates 1 > $O/bench_G400_C1.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --candidates 256 > $O/bench_G400_C256.json 2>/dev/null
FUELMI_FLEET_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --cpu-budget 6 > $O/bench_G400_two_ranks_one_device.json 2> $O/bench_two.err
for WL in G400 G800; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cycle_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline > $O/bench_${WL}_under_rocprof.json 2>/dev/null
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --serial-stages > /dev/null 2>&1
done
{ echo "# two consecutive plan cycles from the middle of the timed region (rocprofv3 --kernel-trace of python bench.py --no-cpu-baseline, commit $FUELMI_COMMIT;"
  echo "# the profiler slows the host that issues both streams: the period under it is longer than the unprofiled cycle): start [us], duration, queue, kernel"
  python scripts/cycle_timeline.py $O/cycle_G400/s_kernel_trace.csv; } > $O/cycle_timeline_G400.txt 2>&1
for WL in G400K G400E; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$WL -o s -- python bench.py --workload $WL --no-cpu-baseline --serial-stages --steps 20 --warmup 3 > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stream -o s -- python bench.py --workload G800S --no-cpu-baseline --steps 40 > /dev/null 2>&1
{ echo "# two consecutive streaming frames (rocprofv3 --kernel-trace of python bench.py --workload G800S --no-cpu-baseline --steps 40, commit $FUELMI_COMMIT)"
  python scripts/cycle_timeline.py $O/stream/s_kernel_trace.csv k_insert_classify; } > $O/stream_timeline_G800S.txt 2>&1
# in-kernel phase stamps of the packed z/y pass, and the ESDF kernels per family (same box)
{ for WL in G400 G800 G400K; do FUELMI_ZY_TIMING=1 python scripts/esdf_only.py $WL 0 3 2>&1 | grep -E "zy2-timing|x2-timing|slowest" | tail -4 | sed "s/^/$WL /"; done
  for WL in G400 G800 G400K G400E; do for FAM in 0 2 1; do echo -n "$WL family $FAM: "; python scripts/esdf_only.py $WL $FAM 8; done; done; } > $O/esdf_family_ab.txt 2>&1
# this round's one remaining switch, same box, interleaved: the last workgroup of k_tile_cross resolving the search (default)
# against k_resolve as a launch of its own (FUELMI_FR_FUSE=0)
NOTEST=1 bash scripts/r6_fuse_ab.sh > $O/cross_resolve_fusion_ab_final.txt 2>&1
# the round-5 build of the library on the same box (build/r5 is a worktree of 5605e15 built by hand before the call)
if [ -d build/r5/fuel_amd ]; then
  { echo "# same box, same call: round-5 final (5605e15) against this tree, twice, interleaved; bench.py --no-cpu-baseline of each (value, stage_ms in the cycle, isolated)"
    for REP in 1 2; do for WL in G400 G800 G400K G400E G800S; do
      for T in build/r5 .; do (cd $T && python bench.py --workload $WL --no-cpu-baseline 2>/dev/null | T=$T WL=$WL python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ['WL'], 'r5' if 'r5' in os.environ['T'] else 'r6', round(d['value'],1), d['stage_ms'], d.get('stage_ms_isolated'))"); done; done; done; } > $O/r5_vs_r6_same_box.txt 2>&1
fi
# the driver's own command line (short timed regions, repeated) and the host's time per call / inside the search calls
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_G400_driver_cmdline.json 2>/dev/null
{ echo "# FUELMI_HOST_TIMING=1 python bench.py --no-cpu-baseline (host microseconds inside fuelmi_frontier_search_begin / _end per search), commit $FUELMI_COMMIT"
  FUELMI_HOST_TIMING=1 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep host-timing | tail -2
  echo "# hardware queues: scripts/r5_hwq.py at the library's fuelmi_init() default, then with the runtime's (GPU_MAX_HW_QUEUES=4)"
  python scripts/r5_hwq.py 2>&1 | tail -5; GPU_MAX_HW_QUEUES=4 python scripts/r5_hwq.py 2>&1 | tail -5; } > $O/host_timing.txt 2>&1
{ echo "# FUELMI_FR_TIMING=1 python bench.py --workload W --no-cpu-baseline --steps 5 --warmup 2 --serial-stages: in-kernel phase stamps of the frontier chain (last search), commit $FUELMI_COMMIT"
  for WL in G400 G800 G800S; do FUELMI_FR_TIMING=1 python bench.py --workload $WL --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep "fr-timing" | tail -6 | grep -v "entry avg" | cut -c1-330 | sed "s/^/$WL /"; done; } > $O/frontier_phase_stamps.txt 2>&1
timeout 300 python scripts/bench_next.py > $O/next_rows.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/next -o s -- python scripts/bench_next.py > /dev/null 2>&1
timeout 300 python scripts/facade_bench.py --map G800S --frames 30 > $O/facade_bench_G800S.json 2> $O/facade_bench.err
timeout 300 python scripts/facade_bench.py --fullbox G400 > $O/facade_bench_G400_fullbox.json 2>> $O/facade_bench.err
for RO in 1 2; do timeout 300 python bench.py --workload G800S --no-cpu-baseline --reference-order $RO > $O/bench_G800S_reforder$RO.json 2>/dev/null; done
timeout 300 python bench.py --no-cpu-baseline --reference-order 1 --steps 20 --warmup 3 > $O/bench_G400_reforder1.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/reforder1_G400 -o s -- python bench.py --no-cpu-baseline --reference-order 1 --steps 20 --warmup 3 > /dev/null 2>&1
# where a streaming frame goes (host wall clock per call group, device timeline from events) and what the reference's cell
# order costs on the full box (the level sweeps' own clock; workgroup sizes of the large-cluster sweep; x pass phase stamps)
{ echo "# FUELMI_STREAM_TIMING=2 python bench.py --workload G800S --no-cpu-baseline (one line pair per bench_stream call: warm-up, timed, frame sources), commit $FUELMI_COMMIT"
  FUELMI_STREAM_TIMING=2 python bench.py --workload G800S --no-cpu-baseline 2>&1 >/dev/null | grep stream-timing
  } > $O/stream_frame_timing.txt 2>&1
{ echo "# FUELMI_FR_TIMING=1 python bench.py --no-cpu-baseline --reference-order 1 --steps 10 --warmup 3 (400x400x100, full box: one cluster of 139 k cells), commit $FUELMI_COMMIT"
  FUELMI_FR_TIMING=1 python bench.py --no-cpu-baseline --reference-order 1 --steps 10 --warmup 3 2>&1 >/dev/null | grep "reference order" | tail -2
  echo "# x pass phase stamps (FUELMI_ZY_TIMING=1, scripts/esdf_only.py)"
  for WL in G400 G800; do FUELMI_ZY_TIMING=1 python scripts/esdf_only.py $WL 0 3 2>&1 | grep x2-timing | tail -1 | sed "s/^/$WL /"; done; } > $O/reference_order_timing.txt 2>&1
timeout 300 python -m pytest tests/test_fleet_gpu.py -q -s -m gpu 2>&1 | grep -E "fleet on one device|bench --gpus|passed|failed" > $O/fleet_one_device.txt
timeout 300 python -m pytest tests/test_perf_gpu.py -q -s -m perf 2>&1 | grep -E "ESDF ms|z/y pass ms|passed|failed" > $O/perf_statements.txt
for f in bench_G400 bench_G800 bench_G400K bench_G400E bench_G800S bench_G400_two_ranks_one_device; do tail -1 $O/$f.json | cut -c1-160; done
cat $O/fleet_one_device.txt $O/perf_statements.txt; tail -1 $O/facade_bench_G800S.json | cut -c1-300
find $O -name "*.csv" | wc -l
