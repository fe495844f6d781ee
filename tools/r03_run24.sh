set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cell_store.py tests/test_gpu_parity.py tests/test_gpu_tiles_native.py tests/test_gpu_caller_lists.py -m gpu -x -q > gpurun_out/r03_pytest24.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_pytest24.log | tail -3
python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config2 ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r03m_settled_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 --warmup 400 > $R/gpurun_out/r03m_settled_trace.log 2>&1
( cd $R; python tools/rocprof_summary.py gpurun_out/r03m_settled_trace/bench_results.db 60 --timed k_solve_flow6 60 > gpurun_out/r03m_settled_kernel_stats.txt; rm -rf gpurun_out/r03m_settled_trace )
