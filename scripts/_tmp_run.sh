cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for k in 1 2 3; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('G400', d['value'], d['ms_per_step'], d['stage_ms'], d['stage_ms_isolated']['frontier'])"
done
FUELMI_HCELLS_DIRECT=100000000 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('G400 direct', d['value'], d['ms_per_step'], d['stage_ms'], d['stage_ms_isolated']['frontier'])"
