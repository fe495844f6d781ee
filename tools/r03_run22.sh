set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cell_store.py tests/test_gpu_parity.py tests/test_gpu_tiles_native.py -m gpu -x -q > gpurun_out/r03_pytest22.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_pytest22.log | tail -3
B="python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5"
for CFG in "flow6_compact_records=0" "flow6_compact_records=1" "flow6_compact_records=0" "flow6_compact_records=1"; do
$B --opt $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$CFG ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'])"
done
for S in config3 config5; do for CFG in "flow6_compact_records=0" "flow6_compact_records=1"; do
$B --scene $S --no-settled --opt $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$S $CFG ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'])"
done; done
