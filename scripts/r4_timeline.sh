#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_timeline; rm -rf $O; mkdir -p $O
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o s -- python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
python scripts/cycle_timeline.py $(find $O/t -name "*kernel_trace.csv" | head -1)
head -12 $(find $O/t -name "*kernel_stats.csv" | head -1) | cut -c1-150
