#!/bin/bash
# quick gate for z/y kernel work: ESDF parity subset, timings per family (G400, G800), VALU/SALU instruction counts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/r4_esdf_parity.sh
for wl in G400 G800; do
  for fam in 2 0; do echo -n "$wl fam $fam: "; python scripts/esdf_only.py $wl $fam 8; done
  echo -n "$wl fam 0 timing: "; FUELMI_ZY_TIMING=1 python scripts/esdf_only.py $wl 0 3 2>&1 | grep zy-timing | tail -1
done
O=gpurun_out/r4_zy_pmc; rm -rf $O; mkdir -p $O
C1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
timeout 300 rocprofv3 --pmc $C1 --output-format csv -d $O/pk_c1 -o s -- python scripts/esdf_only.py G400 0 4 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r4_zy_pmc/pk_c*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "esdf_zy" in k: print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
