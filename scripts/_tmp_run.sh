cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "insert or fusion or depth or stream or golden or ceiling or camera" 2>&1 | tail -3
python bench.py --workload G800S --no-cpu-baseline > gpurun_out/ins1.json 2>gpurun_out/ins1.err
python - <<PY
import json
d=json.loads(open("gpurun_out/ins1.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("frame_source"), d["stage_ms"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ins1 -o s -- python bench.py --workload G800S --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ins1/s_kernel_stats.csv")):
    if float(r["Percentage"])>1.5: print("%-30s avg %7.1f us calls %s" % (r["Name"].split("(")[0][:30], float(r["AverageNs"])/1e3, r["Calls"]))
PY
