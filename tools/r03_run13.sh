set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cell_store.py tests/test_gpu_caller_lists.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
for CFG in "flow6_poll_k=1" "flow6_poll_k=2" "flow6_poll_k=1" "flow6_poll_k=2"; do
$B --opt $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$CFG ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'], 'launches', d.get('launches_per_tick'))"
done
python tools/flow_trace.py 64 40 6 2>&1 | tail -25
