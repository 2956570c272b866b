#!/bin/bash
FUELMI_ESDF_DEBUG=1 python bench.py --no-cpu-baseline --steps 3 --warmup 2 2>&1 | grep "esdf regime" | head -12
FUELMI_ESDF_DEBUG=1 python -m pytest tests/test_gpu_parity_r2.py -m gpu -x -q -s -k switches 2>&1 | grep "esdf regime\|explored hall" | head
