#!/bin/bash
# dev aid: WRITE_SIZE / FETCH_SIZE per launch of the ESDF kernels for a given environment (tile-shape experiments)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-probe}
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pw_$tag -o s -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pf_$tag -o s -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pf_$tag/s_counter_collection.csv gpurun_out/pw_$tag/s_counter_collection.csv gpurun_out/pmc_$tag.json > /dev/null
python - <<PY
import json
t=json.load(open("gpurun_out/pmc_$tag.json"))["kernels"]
for k in t:
    if "esdf" in k: print("$tag", k[:30], "fetch MB %.1f write MB %.1f" % (2*t[k]["FETCH_SIZE_KiB_raw"]/1024*1.048576, t[k]["WRITE_SIZE_KiB"]/1024*1.048576))
PY
