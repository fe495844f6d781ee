# usage: bash tools/r03_trace.sh <tag> <bench args...>   -> gpurun_out/<tag>_kernel_stats.txt (timed window only)
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --min-seconds 0 "$@" > $R/gpurun_out/${TAG}_trace.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/${TAG}_trace/bench_results.db 60 > gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/${TAG}_trace
