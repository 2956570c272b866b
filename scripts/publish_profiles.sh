#!/bin/bash
# copy the summaries scripts/collect_profiles.sh left under gpurun_out/prof_<round> into profiles/ (tracked)
R=${1:-r06}; O=gpurun_out/prof_$R; P=profiles
for f in bench_G400 bench_G800 bench_G400K bench_G400E bench_G400_C1 bench_G400_C256 bench_G400_under_rocprof bench_G800_under_rocprof bench_G400_two_ranks_one_device bench_G800S_100steps; do
  [ -s $O/$f.json ] && tail -1 $O/$f.json > $P/${R}_$f.json
done
[ -s $O/bench_G800S.json ] && tail -1 $O/bench_G800S.json > $P/${R}_bench_G800S_streaming.json
for WL in G400 G800; do
  cp $O/cycle_$WL/s_kernel_stats.csv $P/${R}_bench_${WL}_kernel_stats.csv
  cp $O/serial_$WL/s_kernel_stats.csv $P/${R}_bench_${WL}_serial_stages_kernel_stats.csv
  cp $O/pmc_hbm_traffic_$WL.json $P/${R}_pmc_hbm_traffic_$WL.json
  [ -s $O/pmc_sq_$WL.json ] && cp $O/pmc_sq_$WL.json $P/${R}_pmc_sq_$WL.json
done
for WL in G400K G400E; do cp $O/serial_$WL/s_kernel_stats.csv $P/${R}_bench_${WL}_serial_stages_kernel_stats.csv; done
cp $O/stream/s_kernel_stats.csv $P/${R}_bench_G800S_streaming_kernel_stats.csv
cp $O/next/s_kernel_stats.csv $P/${R}_next_rows_kernel_stats.csv
cp $O/next_rows.json $P/${R}_next_rows.json
[ -s $O/reforder1_G400/s_kernel_stats.csv ] && cp $O/reforder1_G400/s_kernel_stats.csv $P/${R}_bench_G400_reforder1_kernel_stats.csv
[ -s $O/facade_bench_G800S.json ] && tail -1 $O/facade_bench_G800S.json > $P/${R}_facade_bench_G800S.json
[ -s $O/facade_bench_G400_fullbox.json ] && tail -1 $O/facade_bench_G400_fullbox.json > $P/${R}_facade_bench_G400_fullbox.json
for f in bench_G800S_reforder1 bench_G800S_reforder2 bench_G400_reforder1; do [ -s $O/$f.json ] && tail -1 $O/$f.json > $P/${R}_$f.json; done
[ -s $O/bench_G400_driver_cmdline.json ] && tail -1 $O/bench_G400_driver_cmdline.json > $P/${R}_bench_G400_driver_cmdline.json
for f in fleet_one_device perf_statements esdf_family_ab esdf_instruction_counts cross_resolve_fusion_ab_final frontier_phase_stamps cycle_timeline_G400 stream_timeline_G800S stream_frame_timing reference_order_timing r5_vs_r6_same_box host_timing; do
  [ -s $O/$f.txt ] && cp $O/$f.txt $P/${R}_$f.txt
done
ls $P | grep ${R}_ | wc -l
