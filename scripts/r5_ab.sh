#!/bin/bash
# round-5 dev aid: A/B of an environment switch on one box.  usage: r5_ab.sh "ENV=VAL" [workloads...]
sw="$1"; shift
mkdir -p gpurun_out/r5
for w in "${@:-G400}"; do
  for mode in new old; do
    for ser in "" "--serial-stages"; do
      if [ $mode = old ]; then export $sw; else unset ${sw%%=*}; fi
      python bench.py --workload $w --no-cpu-baseline $ser 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w $mode $ser', round(d['value'],1), d['stage_ms'], d['roofline'].get('frac'))"
    done
  done
done
