#!/bin/bash
# ESDF parity subset (every family, ragged boxes, signed / optimistic, kernel choice) -- quick gate for kernel work
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity.py tests/test_gpu_parity_r3.py -m gpu -x -q \
  -k "esdf or g400 or sparse or ragged or kernel_choice or switches or signed or smoke or local" 2>&1 | tail -5
