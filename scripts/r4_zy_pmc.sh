#!/bin/bash
# instruction counts and cycles of the z/y pass variants (SQ counters, own pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_zy_pmc; mkdir -p $O
C1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
C2="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
C3="GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() { # tag fam env...
  tag=$1; fam=$2; shift 2
  for i in 1 2 3; do
    eval cs=\$C$i
    env "$@" timeout 300 rocprofv3 --pmc $cs --output-format csv -d $O/${tag}_c$i -o s -- python scripts/esdf_only.py ${WL:-G400} $fam 4 > /dev/null 2> $O/${tag}_c$i.err
  done
}
run plain32 2 X=1
run pk 0 FUELMI_ZY_PP=0
run pp 0 X=1
python - <<'PY'
import csv, glob, collections
for tag in ("plain32", "pk", "pp"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r4_zy_pmc/%s_c*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "esdf" not in k: continue
        print(tag, k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
