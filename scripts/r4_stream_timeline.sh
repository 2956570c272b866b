#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_stl; rm -rf $O; mkdir -p $O
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python bench.py --workload G800S --no-cpu-baseline --steps 40 > $O/bench.json 2>/dev/null
python scripts/cycle_timeline.py $(find $O/t -name "*kernel_trace.csv" | head -1) k_insert_classify
