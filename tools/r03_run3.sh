set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_cell_store.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r03c_pytest.log
cat gpurun_out/r03c_pytest.log
B="python bench.py --no-cpu-baseline --no-order-check --min-seconds 0.5"
for CFG in "resort_every=32 --opt resort_partition=0 --opt flow6_slot_blocks=0" "resort_every=32 --opt resort_partition=0 --opt flow6_slot_blocks=1" "resort_every=32 --opt resort_partition=1 --opt flow6_slot_blocks=1" "resort_every=128 --opt resort_partition=1 --opt flow6_slot_blocks=1" "resort_every=32 --opt resort_partition=1 --opt flow6_slot_blocks=0"; do
  echo "== $CFG"
  $B --opt $CFG 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/tick', round(d['ms_per_step'],4), 'solver frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), 'settled frac', d['settled']['roofline']['frac'])"
done
python tools/store_probe.py 64 200 2>&1 | tail -12
