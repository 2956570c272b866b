#!/bin/bash
# dev aid: the frontier chain's tuning hooks on one box (headline cycles/s, frontier stage isolated)
run() {
  env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s' % '$*', round(d['value']), d['stage_ms_isolated']['frontier'], d['frontier_path'])"
}
run A=default
run FUELMI_NO_GRAPH=1
run FUELMI_FT_THREADS=256,512,512
run FUELMI_FT_THREADS=512,256,512
run FUELMI_FT_THREADS=512,512,256
run FUELMI_FT_THREADS=256,256,256
run FUELMI_FTILE=8x32
run FUELMI_FTILE=16x16
run A=default
