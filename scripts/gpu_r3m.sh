#!/bin/bash
FUELMI_FR_TIMING=1 timeout 120 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --serial-stages 2>&1 | grep "fr-timing" | tail -6 | cut -c1-260
