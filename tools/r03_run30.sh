set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest30.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_pytest30.log | tail -3
for K in 0 1 0 1; do python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5 --opt readback_kernel=$K 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('readback_kernel=$K ms/tick', round(d['ms_per_step'],4), 'G/s', round(d['value']/1e9,3), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4))"; done
