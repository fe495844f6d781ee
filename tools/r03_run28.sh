set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest28.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_pytest28.log | tail -3
python bench.py --no-cpu-baseline --no-order-check --min-seconds 0.5 > gpurun_out/r03r_bench.json 2> gpurun_out/r03r_bench.err; tail -3 gpurun_out/r03r_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r03r_bench.json').read().strip().splitlines()[-1])
print('config2 ms/tick', round(d['ms_per_step'],4), 'G/s', round(d['value']/1e9,3), 'frac', d['roofline']['frac'], 'settled ms', round(d['settled']['ms_per_step'],4), d['settled']['roofline']['frac'])
print('phase', d['phase_ms_per_step_rank0'], d['gpu_event_ms'])
for k in ('config3','config5'): print(k, round(d[k]['ms_per_step'],4), d[k]['roofline']['frac'])
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r03r_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 > $R/gpurun_out/r03r_trace.log 2>&1
cd $R; python tools/trace_gaps.py gpurun_out/r03r_trace/bench_results.db 240 | head -8; rm -rf gpurun_out/r03r_trace
