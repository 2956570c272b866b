cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in 4 5; do
FUELMI_INS_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/insdbg$dbg -o s -- python bench.py --workload G800S --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/insdbg$dbg/s_kernel_stats.csv")):
    if "k_insert" in r["Name"]: print("dbg $dbg %-22s avg %7.1f us" % (r["Name"].split("(")[0], float(r["AverageNs"])/1e3))
PY
done
