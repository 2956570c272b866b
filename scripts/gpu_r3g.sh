#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -${TAILN:-6} | cut -c1-300
