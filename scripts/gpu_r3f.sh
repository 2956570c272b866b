#!/bin/bash
for V in 3 4; do
echo "== variant $V"
FUELMI_CCL_VARIANT=$V FUELMI_FR_TIMING=1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep "fr-timing. tiles\|fr-timing. ccl" | tail -2 | cut -c1-300
FUELMI_CCL_VARIANT=$V FUELMI_FR_TIMING=1 python bench.py --workload G800S --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep "fr-timing. tiles\|fr-timing. ccl" | tail -2 | cut -c1-300
done
