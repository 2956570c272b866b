#!/bin/bash
# copy the summaries scripts/collect_profiles.sh left under gpurun_out/prof_r01 into profiles/ (tracked)
O=gpurun_out/prof_r01; P=profiles; R=${1:-r01}
tail -1 $O/bench.json > $P/${R}_bench_G400.json
tail -1 $O/bench_C1.json > $P/${R}_bench_G400_C1.json
tail -1 $O/bench_C256.json > $P/${R}_bench_G400_C256.json
tail -1 $O/bench_under_rocprof.json > $P/${R}_bench_G400_under_rocprof.json
tail -1 $O/bench_G800S.json > $P/${R}_bench_G800S_streaming.json
cp $O/cycle/s_kernel_stats.csv $P/${R}_bench_G400_kernel_stats.csv
cp $O/serial/s_kernel_stats.csv $P/${R}_bench_G400_serial_stages_kernel_stats.csv
cp $O/stream/s_kernel_stats.csv $P/${R}_bench_G800S_streaming_kernel_stats.csv
cp $O/next/s_kernel_stats.csv $P/${R}_next_rows_kernel_stats.csv
cp $O/next_rows.json $P/${R}_next_rows.json
cp $O/pmc_fetch/s_counter_collection.csv $P/${R}_pmc_FETCH_SIZE_counter_collection.csv
cp $O/pmc_write/s_counter_collection.csv $P/${R}_pmc_WRITE_SIZE_counter_collection.csv
cp $O/pmc_hbm_traffic.json $P/${R}_pmc_hbm_traffic.json
