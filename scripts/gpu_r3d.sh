#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FUELMI_FR_TIMING=1 python bench.py --workload G800S --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep fr-timing | tail -14 | cut -c1-330
