set -u
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6)
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
$B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'])"
python tools/soak_step_many.py 2>&1 | tail -3
