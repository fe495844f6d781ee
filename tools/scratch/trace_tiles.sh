#!/bin/bash
# kernel-trace summary of the tile line (8 tiles on one GPU): the replayed 60 ticks = the last 480 tile-ticks
SC=${1:-config4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace -d $O/tt_${SC}_trace -o bench -- python $R/bench.py --gpus 1 --scene $SC --no-cpu-baseline > $O/tt_${SC}.log 2>&1
( cd $R; python tools/rocprof_summary.py gpurun_out/tt_${SC}_trace/bench_results.db 480 --tick-start k_integrate --timed k_solve_flow6 2400 > gpurun_out/tt_${SC}_kernel_stats.txt; rm -rf gpurun_out/tt_${SC}_trace )
echo "== $SC"; cut -c1-60,75-140 $O/tt_${SC}_kernel_stats.txt | head -${HEAD:-40}
grep -a '"metric"' $O/tt_${SC}.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'solver ms/tile-tick', d['solver_kernel_ms_per_tile_tick_rank0'], 'tile-tick', d['tile_tick_ms_rank0'])"
