#!/bin/bash
# quick measurement loop: bench line + serialised kernel stats of the G400 cycle
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/q_${1:-x}
mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ser -o s -- python bench.py --no-cpu-baseline --serial-stages --steps 100 > $O/bench_serial_under_rocprof.json 2>/dev/null
python - <<PY
import json,csv
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],4), d["stage_ms"], d["stage_ms_isolated"])
rows=list(csv.DictReader(open("$O/ser/s_kernel_stats.csv")))
for r in rows[:26]:
    print("%-60s %6s %10.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
tail -3 $O/bench.err
