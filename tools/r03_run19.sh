set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_compound_bodies.py tests/test_gpu_fullsize.py tests/test_gpu_obstacles.py -m gpu -x -q > gpurun_out/r03_pytest19.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_pytest19.log | tail -4
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
for S in config5 config5; do
$B --scene $S --no-settled 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$S ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], d.get('phase_ms'))"
done
