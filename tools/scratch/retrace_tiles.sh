#!/bin/bash
# the two tile lines' kernel-trace summaries once more (tools/collect_profiles.sh's run_tiles, the trace pass only)
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pair in "config4_tiles config4" "config5_tiles config5_tiles"; do
  set -- $pair; N=$1; SC=$2
  timeout 500 rocprofv3 --kernel-trace -d $O/${TAG}_${N}_trace -o bench -- python $R/bench.py --gpus 1 --scene $SC --no-cpu-baseline > $O/${TAG}_${N}_trace.log 2>&1
  ( cd $R; python tools/rocprof_summary.py gpurun_out/${TAG}_${N}_trace/bench_results.db 480 --tick-start k_integrate --timed k_solve_flow6 2400 > gpurun_out/${TAG}_${N}_kernel_stats.txt; rm -rf gpurun_out/${TAG}_${N}_trace )
  head -n 3 $O/${TAG}_${N}_kernel_stats.txt | cut -c1-140
done
