#!/bin/bash
# round-end evidence: bench line, rocprofv3 kernel stats of the same command, isolated (serial-stage)
# kernel stats, PMC HBM traffic (separate passes), next-row measurements.  Everything under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r01
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cycle -o s -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o s -- python bench.py --no-cpu-baseline --serial-stages > $O/bench_serial_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o s -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o s -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
python scripts/pmc_summary.py $O/pmc_fetch/s_counter_collection.csv $O/pmc_write/s_counter_collection.csv $O/pmc_hbm_traffic.json
python bench.py --workload G800S > $O/bench_G800S.json 2> $O/bench_G800S.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stream -o s -- python bench.py --workload G800S --no-cpu-baseline > /dev/null 2>&1
python bench.py --no-cpu-baseline --candidates 1 > $O/bench_C1.json 2>/dev/null
python bench.py --no-cpu-baseline --candidates 256 > $O/bench_C256.json 2>/dev/null
python scripts/bench_next.py > $O/next_rows.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/next -o s -- python scripts/bench_next.py > /dev/null 2>&1
tail -1 $O/bench.json; tail -1 $O/bench_G800S.json; tail -1 $O/bench_C1.json | cut -c1-120; tail -1 $O/bench_C256.json | cut -c1-120; tail -1 $O/next_rows.json
find $O -name "*.csv" | head -20
