#!/bin/bash
# kernel trace of one bench window: bash tools/r06/trace_cfg.sh <tag> <timed ticks> <bench args...>  -> gpurun_out/<tag>_kernel_stats.txt
TAG=$1; K=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/${TAG}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 "$@" > $O/${TAG}_trace.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/${TAG}_trace/bench_results.db $K --timed k_solve_flow6 $K > gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/${TAG}_trace
cat gpurun_out/${TAG}_kernel_stats.txt
