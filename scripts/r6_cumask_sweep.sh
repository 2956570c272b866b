#!/bin/bash
# give the two streams their own CUs (hipExtStreamCreateWithCUMask through the FUELMI_CUMASK_FR /
# FUELMI_CUMASK_MAP hooks of fuelmi_stream_create).  Same box, every configuration twice, interleaved.
# usage: r6_cumask_sweep.sh WORKLOAD [bench args]
cd $GRAFT_REPO_ROOT
WL=${1:-G400}; shift
CFGS=("X=1" \
 "FUELMI_CUMASK_FR=0xff:32" \
 "FUELMI_CUMASK_FR=0xff:32 FUELMI_CUMASK_MAP=0xff:32" \
 "FUELMI_CUMASK_FR=0x03:32 FUELMI_CUMASK_MAP=0xfc:32" \
 "FUELMI_CUMASK_FR=0x0f:32 FUELMI_CUMASK_MAP=0xf0:32" \
 "FUELMI_CUMASK_FR=0x01:32 FUELMI_CUMASK_MAP=0xfe:32" \
 "FUELMI_CUMASK_FR=0xff:32 FUELMI_CUMASK_MAP=0xfc:32" \
 "FUELMI_CUMASK_FR=0xff:32 FUELMI_CUMASK_MAP=0xf0:32" \
 "FUELMI_CUMASK_FR=0x03:32 FUELMI_CUMASK_MAP=0xff:32" \
 "FUELMI_CUMASK_FR=0x0f:32 FUELMI_CUMASK_MAP=0xff:32" \
 "FUELMI_CUMASK_FR=0xff:8 FUELMI_CUMASK_MAP=0xff:8-31" \
 "FUELMI_CUMASK_FR=0xff:16 FUELMI_CUMASK_MAP=0xff:16-31" \
 "FUELMI_CUMASK_FR=0xff:16 FUELMI_CUMASK_MAP=0xff:32" \
 "FUELMI_CUMASK_FR=0xff:32 FUELMI_CUMASK_MAP=0xff:8-31")
echo "# workload $WL $@ : cycles/s (or frames/s) | stage_ms (in-cycle event brackets) | host issue us"
for rep in $(seq 1 ${REPS:-2}); do
for cfg in "${CFGS[@]}"; do
  echo -n "$cfg | "; env $cfg timeout 300 python bench.py --workload $WL --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
L=[l for l in sys.stdin if l.startswith('{')]
if not L: print('FAILED'); sys.exit(0)
d=json.loads(L[0])
print('%.0f'%d['value'], '|', d.get('stage_ms'), '|', d.get('host_issue_us'))"
done; done
