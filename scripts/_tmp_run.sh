cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ser -o s -- python bench.py --no-cpu-baseline --serial-stages > /dev/null 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ser/s_kernel_stats.csv")):
    if float(r["Percentage"])>0.8: print("%-30s avg %7.1f us calls %s" % (r["Name"].split("(")[0][:30], float(r["AverageNs"])/1e3, r["Calls"]))
PY
