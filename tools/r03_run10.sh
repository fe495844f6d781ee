set -u
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4)
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
for CFG in "merged_lists=0" "merged_lists=1"; do
$B --opt $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$CFG ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'])"
done
