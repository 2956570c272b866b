#!/bin/bash
# tuning aid: frontier-chain kernel times vs CCL tile shape / threads ("TXxTY:threads" list in CFGS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-7x16:256 7x16:512}; do
  t=${cfg%%:*}; th=${cfg##*:}
  FUELMI_CCL_TILE=$t FUELMI_CCL_THREADS=$th rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_$cfg -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --serial-stages ${BENCH_ARGS} > gpurun_out/sweep_$cfg.log 2>&1
  python - <<PY
import csv, json
rows=list(csv.reader(open('gpurun_out/sweep_$cfg/s_kernel_stats.csv')))
d={r[0].split('(')[0].split('<')[0].replace('void ',''):float(r[3])/1e3 for r in rows[1:]}
keys=['k_load_var','k_pred','k_scan_sums','k_compact','k_ccl_local','k_union','k_claim','k_sizes','k_finalize','k_rank_kept','k_ms_hist','k_ms_scan','k_ms_scatter','k_ms_info','k_pack']
print("$cfg", " ".join("%s %.1f" % (k[2:], d.get(k,0)) for k in keys), "| sum %.1f" % sum(d.get(k,0) for k in keys))
try:
    print("   ", json.loads(open('gpurun_out/sweep_$cfg.log').read().strip().splitlines()[-1])['value'])
except Exception as e:
    print("   bench line unreadable", e)
PY
done
