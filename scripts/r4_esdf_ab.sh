#!/bin/bash
# round 4: packed 16-bit z/y pass vs the 32-bit one -- parity of every family, then same-box A/B of the serialised
# stage times and of the overlapped cycle (G400, G800)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r4_esdf_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity.py tests/test_gpu_parity_r3.py -m gpu -x -q \
  -k "esdf or g400 or sparse or ragged or kernel_choice or switches or signed or smoke or local" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for wl in G400 G800; do
  for fam in 0 2; do
    timeout 300 python bench.py --workload $wl --esdf-family $fam --serial-stages --no-cpu-baseline > $O/${wl}_serial_fam$fam.json 2>$O/${wl}_serial_fam$fam.err
    timeout 300 python bench.py --workload $wl --esdf-family $fam --no-cpu-baseline > $O/${wl}_cycle_fam$fam.json 2>$O/${wl}_cycle_fam$fam.err
  done
done
timeout 300 python -m pytest tests/test_perf_gpu.py -m perf -q -s > $O/perf.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_esdf_ab/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(f.split('/')[-1], 'value %.0f'%d['value'], 'iso', d['stage_ms_isolated'], 'in-cycle', d['stage_ms'])
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -15 $O/perf.log
