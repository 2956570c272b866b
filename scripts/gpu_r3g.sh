#!/bin/bash
timeout 1500 python -m pytest tests/test_golden_pillar.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -6
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_golden_pillar.py 2>&1 | grep -v "^$" | tail -${TAILN:-22}
