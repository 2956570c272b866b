#!/bin/bash
timeout 200 python -m pytest tests/test_gpu_parity_r3.py -m gpu -x -q --timeout 90 -k "many_clusters" 2>&1 | tail -${TAILN:-12}
