set -u
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_pair_brick.py -x -q -m gpu 2>&1 | tail -8)
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
for CFG in "pair_brick=1" "pair_brick=2"; do
  echo "== $CFG"
  $B --opt $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/tick', round(d['ms_per_step'],4), 'phases', {k:round(v,4) for k,v in d['phase_ms_per_step_rank0'].items()}, 'settled ms', round(d['settled']['ms_per_step'],4), {k:round(v,4) for k,v in d['settled']['phase_ms_per_step'].items()})"
done
