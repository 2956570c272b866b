#!/bin/bash
# round-5 dev aid: parity suite, then ESDF stage times for the env settings given as arguments ("A=1 B=2" per argument)
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for w in G400 G800; do
  for envs in "" "$@"; do
    for ser in "" "--serial-stages"; do
      env $envs python bench.py --workload $w --no-cpu-baseline $ser 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w [$envs] $ser', round(d['value'],1), d['stage_ms'])"
    done
  done
done
