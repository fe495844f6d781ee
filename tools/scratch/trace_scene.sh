#!/bin/bash
# kernel-trace summary of one bench window:  bash tools/scratch/trace_scene.sh <name> <ticks> "<bench args>"
N=$1; K=$2; ARGS=$3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/ts_${N}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 $ARGS > $O/ts_${N}.log 2>&1
( cd $R; python tools/rocprof_summary.py gpurun_out/ts_${N}_trace/bench_results.db $K --timed k_solve_flow6 $K > gpurun_out/ts_${N}_kernel_stats.txt; rm -rf gpurun_out/ts_${N}_trace )
echo "== $N"; cut -c1-60,75-140 $O/ts_${N}_kernel_stats.txt | head -${HEAD:-26}
