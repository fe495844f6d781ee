set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest21.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_pytest21.log | tail -3
cd /tmp && export TMPDIR=/tmp
for S in config2; do
rocprofv3 --kernel-trace -d $R/gpurun_out/r03k_${S}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 --scene $S > $R/gpurun_out/r03k_${S}_trace.log 2>&1
( cd $R; python tools/rocprof_summary.py gpurun_out/r03k_${S}_trace/bench_results.db 60 --timed k_solve_flow6 60 > gpurun_out/r03k_${S}_kernel_stats.txt; rm -rf gpurun_out/r03k_${S}_trace )
done
cd $R
python bench.py --no-cpu-baseline --no-order-check --no-other-configs --min-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config2 ms/tick', round(d['ms_per_step'],4), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'])"
