set -u
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
for CFG in "part_cell_x8=0" "part_cell_x8=4" "part_cell_x8=6" "part_cell_x8=8" "part_cell_x8=12"; do
$B --opt $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$CFG ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'])"
done
