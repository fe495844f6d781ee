set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_pytest.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_final_pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err ) 2>&1 | grep real
python -c "
import json
d=json.loads(open('gpurun_out/r03_final_bench.json').read().strip().splitlines()[-1])
print('ms', round(d['ms_per_step'],4), 'G/s', round(d['value']/1e9,3), 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'cpu', round(d['cpu_baseline']['value']))"
