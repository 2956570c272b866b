#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/r4_esdf_parity.sh
for wl in G400 G800 G400K; do
  for cfg in "FUELMI_X_PK=0" "X=1" "FUELMI_X_PK_THREADS=512" "FUELMI_X_PK_THREADS=128"; do echo -n "$wl $cfg: "; env $cfg python scripts/esdf_only.py $wl 0 8; done
done
bash scripts/r4_cycle_ab.sh FUELMI_X_PK=0 FUELMI_X_PK_THREADS=512 | cut -c1-150
