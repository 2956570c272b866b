#!/bin/bash
python scripts/facade_bench.py --fullbox G400 2>&1 | tail -3
python scripts/facade_bench.py --map G800S --frames 30 2>&1 | tail -2
