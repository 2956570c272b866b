#!/bin/bash
# round 2: HBM-roofline evidence on the grid that leaves the Infinity Cache (800x800x200, working set >1 GB)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_g800
mkdir -p $O
python bench.py --workload G800 --no-cpu-baseline --steps 40 --warmup 5 > $O/bench_G800.json 2> $O/bench_G800.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o s -- python bench.py --workload G800 --no-cpu-baseline --steps 20 --warmup 3 --serial-stages > $O/bench_G800_serial_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o s -- python bench.py --workload G800 --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o s -- python bench.py --workload G800 --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
python scripts/pmc_summary.py $O/pmc_fetch/s_counter_collection.csv $O/pmc_write/s_counter_collection.csv $O/pmc_hbm_traffic_G800.json
tail -1 $O/bench_G800.json | cut -c1-600
find $O -name "*stats*.csv" | head
