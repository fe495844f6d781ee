set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r03n_c4_trace -o bench -- python $R/bench.py --gpus 1 --scene config4 --no-cpu-baseline --steps 20 --warmup 10 > $R/gpurun_out/r03n_c4_trace.log 2>&1
cd $R; python tools/rocprof_summary.py gpurun_out/r03n_c4_trace/bench_results.db > gpurun_out/r03n_c4_kernel_stats.txt; rm -rf gpurun_out/r03n_c4_trace
tail -1 gpurun_out/r03n_c4_trace.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/tick', d['ms_per_step'], 'tile tick', d['tile_tick_ms_rank0'], d['exchange'])"
