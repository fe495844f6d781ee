set -u
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
for CFG in "flow6_urgent=0" "flow6_urgent=1" "flow6_urgent=0 --opt flow6_poll_prio=3" "flow6_urgent=1 --opt flow6_poll_prio=3" "flow6_urgent=1 --opt flow6_poll_prio=1"; do
  echo "== $CFG"
  $B --opt $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/tick', round(d['ms_per_step'],4), 'solver frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), 'settled frac', d['settled']['roofline']['frac'])"
done
