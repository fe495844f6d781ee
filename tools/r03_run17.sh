set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest17.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_pytest17.log | tail -4
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
$B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config2 ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'])"
for S in config3 config5; do
$B --scene $S --no-settled 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$S ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'])"
done
bash tools/r03_trace.sh r03h_c3 --scene config3
bash tools/r03_trace.sh r03h_c5 --scene config5
