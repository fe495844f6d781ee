set -u
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
for CFG in "flow6_poll_pipe=0" "flow6_poll_pipe=1"; do
$B --opt $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$CFG ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'])"
done
MGF_F6_OPTS=resort_every=0,flow6_poll_pipe=1 python tools/flow_trace.py 64 40 6 2>&1 | tail -8
