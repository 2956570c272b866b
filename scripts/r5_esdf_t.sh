#!/bin/bash
# round-5 dev aid: isolated ESDF pass medians (esdf_only.py) per env setting; arguments: "A=1 B=2" strings
for w in G400 G800; do
  for envs in "" "$@"; do
    echo "$w [$envs] $(env $envs python scripts/esdf_only.py $w -1 8 2>&1 | tail -1)"
    env $envs FUELMI_ZY_TIMING=1 python scripts/esdf_only.py $w -1 3 2>&1 | grep timing | tail -2
  done
done
